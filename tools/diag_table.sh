#!/bin/bash
# GPU box: small-batch throughput table -> gpurun_out/diag/table.txt   (usage: diag_table.sh [tag])
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/diag
mkdir -p "$OUT"
TAG=${1:-table}
line() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,1), 'us/step')"; }
for cfg in "proteinG56_7A 1" "proteinG56_7A 8" "syn150_10A 64" "syn300_10A 1" "syn300_10A 64" "syn300_10A 4096"; do
  set -- $cfg
  st=300; [ $2 -ge 1024 ] && st=60
  python3 bench.py --workload $1 --replicas $2 --steps $st --warmup 40 --no-cpu-baseline --no-single-system 2>/dev/null | line "$1 R=$2 default"
  [ $2 -le 64 ] && UPSIDE_HIP_ASYNC_PREPARE=0 python3 bench.py --workload $1 --replicas $2 --steps $st --warmup 40 --no-cpu-baseline --no-single-system 2>/dev/null | line "$1 R=$2 one-stream"
  [ $2 -le 64 ] && UPSIDE_HIP_ASYNC_PREPARE=0 UPSIDE_HIP_GRAPH=1 python3 bench.py --workload $1 --replicas $2 --steps $st --warmup 40 --no-cpu-baseline --no-single-system 2>/dev/null | line "$1 R=$2 one-stream+graph"
done > "$OUT/$TAG.txt" 2>&1
cat "$OUT/$TAG.txt"
