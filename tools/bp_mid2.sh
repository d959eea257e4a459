for R in 40 48 56; do for cfg in "X=0" "UPSIDE_HIP_BP_CLUSTER=1"; do
  echo -n "R=$R $cfg: "
  env $cfg python bench.py --replicas $R --steps 200 --warmup 50 --no-cpu-baseline --no-single-system --no-parity-check 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print(round(d['value']), round(d['ms_per_step'],3), 'bp', round(d['roofline']['kernels']['bp:rotamer']['avg_ms'],3))"
done; done
