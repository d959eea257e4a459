#!/bin/bash
# A/B of environment switches inside ONE gpurun call: bash tools/ab_env.sh "<bench args>" "VAR=a" "VAR=b OTHER=c" ...
# Prints throughput and the per-kernel-family times of every setting ("-" = no extra variables).
ARGS="$1"; shift
for setting in "$@"; do
  echo "== $setting"
  s="$setting"; [ "$s" = "-" ] && s=""
  env $s python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-single-system --no-parity-check $ARGS 2>gpurun_out/ab_env.err | python3 -c "
import sys,json
L=[l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')]
if not L: print('   FAILED'); sys.exit()
d=json.loads(L[-1])
print('  ', round(d['value']), 'system-steps/s', round(d['ms_per_step'],2), 'ms/step')
print('  ', ' '.join('%s=%.2f' % (k.replace('igraph_','').replace('hbond_coverage','cov').replace('_hydrophobe','H').replace('environment_coverage','env').replace('protein_hbond','hb'), v['avg_ms']) for k,v in sorted(d['roofline']['kernels'].items())))"
  tail -2 gpurun_out/ab_env.err | cut -c1-300
done
