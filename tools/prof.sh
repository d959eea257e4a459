#!/bin/bash
# GPU box: timeline of one MD step + per-kernel totals + (small batches) the per-op trace of the fused lists, for any workload.
#   gpurun -- 'bash tools/prof.sh WORKLOAD R TAG [step_index]'   -> gpurun_out/prof_TAG/{timeline.txt,kstats.txt,fuse.txt,bench.json}
set -u
export TMPDIR=/tmp
W=${1:-syn300_10A}; R=${2:-1}; TAG=${3:-p}; IDX=${4:-20}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
STEPS=${STEPS:-60}
ARGS="--workload $W --replicas $R --steps $STEPS --warmup 30 --no-cpu-baseline --no-single-system --no-parity-check"
python3 bench.py $ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace -d "$OUT/tr" -o tr -- python3 bench.py $ARGS > "$OUT/tr.log" 2>&1
db=$(find "$OUT/tr" -name "*.db" | head -1)
python3 tools/rocpd_timeline.py "$db" $IDX "$OUT/timeline.txt"
python3 tools/rocpd_summary.py "$db" "$OUT/kstats.txt"
find "$OUT" -name "*.db" -delete; rm -rf "$OUT/tr"
if [ "$R" -le 64 ]; then UPSIDE_HIP_FUSE_TRACE=1 python3 bench.py $ARGS 2>&1 >/dev/null | grep "fused op" > "$OUT/fuse.txt"; fi
python3 -c "
import json
d=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][0]); print('$W R=$R', round(d['value']), 'steps/s', round(d['ms_per_step']*1e3,1), 'us/step')"
