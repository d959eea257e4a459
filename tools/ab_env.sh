#!/bin/bash
# usage: ab_env.sh VAR "v1 v2" [rounds] [bench args...]
VAR=$1; VALS=$2; ROUNDS=${3:-2}; shift 3
for r in $(seq $ROUNDS); do for v in $VALS; do
  env $VAR=$v python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-single-system "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
k=d['roofline']['kernels']; pc=d.get('parity_check') or {}
print('$VAR=$v: %d  parity %.1e %s  ' % (round(d['value']), pc.get('max_rel_rms',-1), 'ok' if pc.get('ok',True) else 'FAIL') + '  '.join('%s=%.3f' % (n.replace('igraph_',''), x['avg_ms']) for n,x in sorted(k.items()) if n.startswith('bp') or 'rotamer' in n))"
done; done
