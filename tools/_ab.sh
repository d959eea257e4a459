for v in 0 1 2 0 1 2; do
echo "== RESIDENT=$v"
UPSIDE_HIP_BP_RESIDENT=$v python bench.py --steps 60 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'])"
done
for v in 0 1; do UPSIDE_HIP_BP_RESIDENT=$v python tools/bp_trace.py syn300_10A 1024 | grep -v "^sweeps"; done
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
