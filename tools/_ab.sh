L=upside-md_amd/csrc
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alternate or forces or md or rotamer or bp" > gpurun_out/pytest_bp.txt 2>&1; grep -E "passed|failed" gpurun_out/pytest_bp.txt
cp $L/libupside_hip.so $L/libupside_hip.new
for i in 1 2; do
for v in old new; do
cp $L/libupside_hip.$v $L/libupside_hip.so
echo "== $v"
python bench.py --steps 45 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['kernels']['igraph_fwd:rotamer']['avg_ms'])"
done; done
for v in new; do cp $L/libupside_hip.$v $L/libupside_hip.so; python tools/bp_trace.py syn300_10A 1024 | grep -v "^sweeps"; done
