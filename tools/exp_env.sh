L=upside-md_amd/csrc
for lib in c2b c4b; do
cp $L/exp/$lib.so $L/libupside_hip.so
for st in 0 1; do
  echo "== $lib UPKEEP_STREAMS=$st"
  UPSIDE_HIP_UPKEEP_STREAMS=$st python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ', round(d['value']), 'system-steps/s', round(d['ms_per_step'],2), 'ms/step')"
done; done
