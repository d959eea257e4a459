#!/bin/bash
# GPU box: A/B of environment variants at large batches (usage: var4096.sh "VAR=.." "VAR2=.." ...; the empty variant is the default)
export TMPDIR=/tmp
line() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), 'ms/step')"; }
[ $# -eq 0 ] && set -- ""
for R in ${REPLICAS:-4096}; do for rep in 1 2; do for v in "$@"; do
  env $v python3 bench.py --replicas $R --steps 80 --warmup 20 --no-cpu-baseline --no-single-system --no-parity-check 2>/dev/null | line "R=$R [$v]"
done; done; done
