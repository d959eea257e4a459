#!/bin/bash
# GPU box: one-step kernel timeline -> gpurun_out/diag/timeline_<workload>_R<replicas>_<tag>.txt   (usage: timeline.sh workload replicas [tag])
set -u
export TMPDIR=/tmp
W=$1; R=$2; TAG=${3:-t}
OUT=$PWD/gpurun_out/diag; mkdir -p "$OUT"; rm -rf "$OUT/tr_tmp"
rocprofv3 --kernel-trace -d "$OUT/tr_tmp" -o trace -- python3 bench.py --workload $W --replicas $R --steps 60 --warmup 20 --no-cpu-baseline --no-single-system --no-parity-check > "$OUT/tr_tmp.log" 2>&1
db=$(find "$OUT/tr_tmp" -name "*.db" | head -1)
[ -n "$db" ] && python3 tools/rocpd_timeline.py "$db" 40 "$OUT/timeline_${W}_R${R}_${TAG}.txt"
rm -rf "$OUT/tr_tmp"
