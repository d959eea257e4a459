for cfg in "UPSIDE_HIP_BP_RESIDENT=1" "UPSIDE_HIP_BP_RESIDENT=2" "UPSIDE_HIP_BP_RESIDENT=0" "UPSIDE_HIP_BP_RESIDENT=1 UPSIDE_HIP_BP_LDS_MSG_KB=120" "UPSIDE_HIP_BP_RESIDENT=1 UPSIDE_HIP_BP_LDS_MSG_KB=90"; do
  echo -n "$cfg: "
  env $cfg python bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-single-system --no-parity-check 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print(round(d['value']), round(d['ms_per_step'],2), 'bp', round(d['roofline']['kernels']['bp:rotamer']['avg_ms'],3))"
done
