#!/bin/bash
# GPU box: small-batch diagnostics (throughput table + one-step timelines) -> gpurun_out/diag/
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/diag
mkdir -p "$OUT"
tools/ubench/launch_chain > "$OUT/launch_chain.txt" 2>&1
line() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,1), 'us/step')"; }
for cfg in "proteinG56_7A 1" "proteinG56_7A 8" "syn150_10A 64" "syn300_10A 1" "syn300_10A 64"; do
  set -- $cfg
  python3 bench.py --workload $1 --replicas $2 --steps 300 --warmup 40 --no-cpu-baseline --no-single-system --no-parity-check 2>/dev/null | line "$1 R=$2 default"
  UPSIDE_HIP_GRAPH=1 python3 bench.py --workload $1 --replicas $2 --steps 300 --warmup 40 --no-cpu-baseline --no-single-system --no-parity-check 2>/dev/null | line "$1 R=$2 graph"
  UPSIDE_HIP_ASYNC_PREPARE=0 python3 bench.py --workload $1 --replicas $2 --steps 300 --warmup 40 --no-cpu-baseline --no-single-system --no-parity-check 2>/dev/null | line "$1 R=$2 one-stream"
  UPSIDE_HIP_ASYNC_PREPARE=0 UPSIDE_HIP_GRAPH=1 python3 bench.py --workload $1 --replicas $2 --steps 300 --warmup 40 --no-cpu-baseline --no-single-system --no-parity-check 2>/dev/null | line "$1 R=$2 one-stream+graph"
done > "$OUT/table.txt" 2>&1
for cfg in "proteinG56_7A 1" "syn300_10A 1" "syn300_10A 64"; do
  set -- $cfg
  rm -rf "$OUT/tr_$1_$2"
  rocprofv3 --kernel-trace -d "$OUT/tr_$1_$2" -o trace -- python3 bench.py --workload $1 --replicas $2 --steps 60 --warmup 20 --no-cpu-baseline --no-single-system --no-parity-check > "$OUT/tr_$1_$2.log" 2>&1
  db=$(find "$OUT/tr_$1_$2" -name "*.db" | head -1)
  [ -n "$db" ] && python3 tools/rocpd_timeline.py "$db" 40 "$OUT/timeline_$1_R$2.txt"
  find "$OUT/tr_$1_$2" -name "*.db" -delete
done
cat "$OUT/launch_chain.txt" "$OUT/table.txt"
