#!/bin/bash
# Compare several builds of libupside_hip.so inside ONE gpurun call (boxes differ by up to 10 %).  Variants are built
# beforehand into upside-md_amd/csrc/exp/<tag>.so (tools/exp_build.sh); usage on the box: bash tools/exp_libs.sh [bench args]
L=upside-md_amd/csrc
cp $L/libupside_hip.so $L/exp/_keep.so
for f in $L/exp/*.so; do
  t=$(basename $f .so); [ "$t" = "_keep" ] && continue
  cp $f $L/libupside_hip.so
  echo "== $t"
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ', round(d['value']), 'system-steps/s', round(d['ms_per_step'],2), 'ms/step')
print('  ', ' '.join('%s=%.2f' % (k.replace('igraph_','').replace('hbond_coverage','cov').replace('_hydrophobe','H').replace('environment_coverage','env').replace('protein_hbond','hb'), v['avg_ms']) for k,v in sorted(d['roofline']['kernels'].items())))"
done
cp $L/exp/_keep.so $L/libupside_hip.so
