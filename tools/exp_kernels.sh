#!/bin/bash
# Per-kernel times (bench.py's HIP-event leg) of the library variants under upside-md_amd/csrc/exp/*.so, inside ONE gpurun call:
#   tools/exp_kernels.sh [workload] [replicas] [rounds]
L=upside-md_amd/csrc
W=${1:-syn300_10A}; R=${2:-4096}; ROUNDS=${3:-1}
cp $L/libupside_hip.so $L/exp/_keep.so
for round in $(seq $ROUNDS); do
for f in $L/exp/*.so; do
  t=$(basename $f .so); [ "$t" = "_keep" ] && continue
  cp $f $L/libupside_hip.so
  python bench.py --workload $W --replicas $R --steps 30 --warmup 10 --no-cpu-baseline --no-single-system --no-parity-check 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
    k=d['roofline']['kernels']
    print('$t: %d  ' % round(d['value']) + '  '.join('%s=%.3f' % (n.replace('igraph_','').replace('hbond_coverage','cov').replace('environment_coverage','env').replace('_hydrophobe','H'), v['avg_ms']) for n, v in sorted(k.items())) + '  pairs: ' + ' '.join('%.1fM' % ((v.get('pair_evaluations') or 0) / 1e6) for n, v in sorted(k.items()) if v.get('pair_evaluations')))
except Exception as e:
    print('$t: FAILED', e)"
done
done
cp $L/exp/_keep.so $L/libupside_hip.so
