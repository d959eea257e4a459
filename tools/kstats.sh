#!/bin/bash
# GPU box: per-kernel times of the default bench command (4096 systems) -> gpurun_out/kstats/<tag>.txt   (usage: kstats.sh [tag] [replicas])
set -u
export TMPDIR=/tmp
TAG=${1:-k}; R=${2:-4096}
OUT=$PWD/gpurun_out/kstats
mkdir -p "$OUT"; rm -rf "$OUT/$TAG"
rocprofv3 --kernel-trace --stats -d "$OUT/$TAG" -o trace -- python3 bench.py --replicas $R --steps 60 --warmup 15 --no-cpu-baseline --no-single-system --no-parity-check > "$OUT/$TAG.log" 2>&1
db=$(find "$OUT/$TAG" -name "*.db" | head -1)
[ -n "$db" ] && python3 tools/rocpd_summary.py "$db" "$OUT/$TAG.txt"
grep '^{' "$OUT/$TAG.log" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value']), d['ms_per_step'])" >> "$OUT/$TAG.txt"
rm -rf "$OUT/$TAG"
head -45 "$OUT/$TAG.txt"; tail -1 "$OUT/$TAG.txt"
