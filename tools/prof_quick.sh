#!/bin/bash
# Quick serialized per-kernel profile on the GPU box (gpurun -- 'bash tools/prof_quick.sh [R] [tag]'): one PMC pass (dispatches
# run one at a time under counter collection, so the durations are per-kernel wall times) of a short bench run.
set -u
export TMPDIR=/tmp
R=${1:-4096}; TAG=${2:-quick}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python3 bench.py --replicas $R --steps 12 --warmup 6 --no-cpu-baseline --no-single-system --no-parity-check"
PMC=${PMC:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE}
rocprofv3 --kernel-trace --pmc $PMC -d "$OUT/sq" -o sq -- $CMD > "$OUT/sq.log" 2>&1
db=$(find "$OUT/sq" -name "*.db" | head -1)
[ -n "$db" ] && python3 tools/rocpd_summary.py "$db" "$OUT/sq_summary.txt"
find "$OUT" -name "*.db" -delete
head -40 "$OUT/sq_summary.txt"
