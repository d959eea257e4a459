for sk in ${SKINS:-0.35 0.5 0.65 0.8}; do
  echo -n "skin $sk: "
  UPSIDE_HIP_SKIN_SCALE=$sk python bench.py --steps 60 --warmup 30 --no-cpu-baseline --no-single-system --no-parity-check 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],2))"
done
