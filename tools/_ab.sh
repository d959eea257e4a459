python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alternate or forces or md or rotamer or bp" > gpurun_out/pytest_bp.txt 2>&1; grep -E "passed|failed" gpurun_out/pytest_bp.txt
for v in 0 1 0 1; do
echo "== RESIDENT=$v"
UPSIDE_HIP_BP_RESIDENT=$v python bench.py --steps 60 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'])"
done
for v in 0 1; do UPSIDE_HIP_BP_RESIDENT=$v python tools/bp_trace.py syn300_10A 1024 | grep -v "^sweeps"; done
