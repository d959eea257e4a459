#!/bin/bash
# GPU box: A/B sweep of environment variants over (workload, replicas) pairs -> stdout
#   usage: sweep.sh "w1 R1;w2 R2;..." "VAR=a" "VAR=b VAR2=c" ...   (the empty variant "" = defaults)
export TMPDIR=/tmp
line() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('$1', round(d['value']), round(d['ms_per_step']*1e3,1), 'us/step')"; }
IFS=';' read -ra CFGS <<< "$1"; shift
for cfg in "${CFGS[@]}"; do
  set -- $cfg "$@"; w=$1; r=$2; shift 2
  st=200; [ $r -ge 1024 ] && st=60
  for v in "$@"; do
    env $v python3 bench.py --workload $w --replicas $r --steps $st --warmup 40 --no-cpu-baseline --no-single-system --no-parity-check 2>/dev/null | line "$w R=$r [$v]"
  done
done
