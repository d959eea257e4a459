#!/bin/bash
# Compare the library variants of upside-md_amd/csrc/exp/*.so (tools/exp_kg.sh) inside ONE gpurun call, over several
# workloads: tools/exp_cmp.sh "proteinG56_7A:1 syn300_10A:1 syn300_10A:4096" [rounds]   (value = system-steps/s; the bench's own parity
# check of the timed engine rides along)
L=upside-md_amd/csrc
CASES=${1:-"proteinG56_7A:1 syn300_10A:1 syn300_10A:4096"}; ROUNDS=${2:-2}
cp $L/libupside_hip.so $L/exp/_keep.so
for round in $(seq $ROUNDS); do
for f in $L/exp/*.so; do
  t=$(basename $f .so); [ "$t" = "_keep" ] && continue
  cp $f $L/libupside_hip.so
  line="$t:"
  for c in $CASES; do
    w=${c%%:*}; r=${c##*:}
    if [ "$r" -ge 1024 ]; then st="--steps 30 --warmup 10"; else st="--steps 600 --warmup 60"; fi
    v=$(python bench.py --workload $w --replicas $r $st --no-cpu-baseline --no-single-system 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
    pc=d.get('parity_check') or {}
    rf=d.get('roofline') or {}
    print('%d(%.0e%s %s=%.3fms)' % (round(d['value']), pc.get('max_rel_rms', -1), '' if pc.get('ok', True) else ' PARITY-FAIL', rf.get('kernel','?'), rf.get('avg_launch_ms', 0)))
except Exception as e:
    print('FAILED')")
    line="$line  $w/R$r=$v"
  done
  echo "$line"
done
done
cp $L/exp/_keep.so $L/libupside_hip.so
