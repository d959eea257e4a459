#!/bin/bash
# GPU box: rocprofv3 kernel-trace averages of the belief-propagation kernels of the default bench command (usage: bp_kernels_time.sh [tag])
set -u
export TMPDIR=/tmp
TAG=${1:-bpk}
OUT=$PWD/gpurun_out/kstats_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/tr" -o trace -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-single-system --no-parity-check > "$OUT/log.txt" 2>&1
db=$(find "$OUT/tr" -name "*.db" | head -1)
python3 tools/rocpd_summary.py "$db" "$OUT/summary.txt"
rm -rf "$OUT/tr"
grep -E "k_rotamer_bp|k_rotamer_pair_energy|k_rotamer_grad2" "$OUT/summary.txt"
grep '^{' "$OUT/log.txt" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value']), d['ms_per_step'])"
