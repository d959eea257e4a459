#!/bin/bash
# GPU box: small-batch throughput table -> gpurun_out/diag/<tag>.txt   (usage: diag_table.sh [tag] ["ENV=1 ENV2=0" variants...])
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/diag
mkdir -p "$OUT"
TAG=${1:-table}; shift
[ $# -eq 0 ] && set -- ""
line() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,1), 'us/step')"; }
for cfg in "proteinG56_7A 1" "proteinG56_7A 8" "syn150_10A 64" "syn300_10A 1" "syn300_10A 64" "syn300_10A 4096"; do
  for v in "$@"; do
    set -- $cfg "$@"; w=$1; r=$2; shift 2
    st=300; [ $r -ge 1024 ] && st=60
    env $v python3 bench.py --workload $w --replicas $r --steps $st --warmup 40 --no-cpu-baseline --no-single-system --no-parity-check 2>/dev/null | line "$w R=$r [$v]"
  done
done > "$OUT/$TAG.txt" 2>&1
cat "$OUT/$TAG.txt"
