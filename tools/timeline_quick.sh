#!/bin/bash
# kernel timeline of one MD step (concurrent streams visible): gpurun -- 'bash tools/timeline_quick.sh [R] [tag]'
set -u
export TMPDIR=/tmp
R=${1:-4096}; TAG=${2:-tl}
OUT=$PWD/gpurun_out/tl_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace -d "$OUT/tr" -o tr -- python3 bench.py --replicas $R --steps 12 --warmup 6 --no-cpu-baseline --no-single-system --no-parity-check > "$OUT/tr.log" 2>&1
db=$(find "$OUT/tr" -name "*.db" | head -1)
python3 tools/rocpd_timeline.py "$db" 10 "$OUT/timeline.txt"
find "$OUT" -name "*.db" -delete
cat "$OUT/timeline.txt"
