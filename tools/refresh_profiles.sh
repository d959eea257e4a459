#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh [all|pmc|bench|configs]'): regenerates the raw material of
# profiles/ under gpurun_out/prof/.  Each rocprofv3 pass profiles the program itself (no shell hop after --), PMC passes
# are separate from each other and carry no trace domains beyond the kernel trace.
#   pmc      kernel trace + the FETCH_SIZE / WRITE_SIZE / SQ passes of the default command, hbm_traffic.json (first: the bench lines use it)
#   bench    bench.py lines at 1 .. 4096 replicas per GPU
#   configs  the other BASELINE configurations (throughput table, REMD and ensemble lines on one GPU)
set -u
export TMPDIR=/tmp
MODE=${1:-all}
OUT=$PWD/gpurun_out/prof
mkdir -p "$OUT"
want() { [ "$MODE" = all ] || [ "$MODE" = "$1" ]; }

if want pmc; then
[ -s "$OUT/bench_R4096.json" ] || python3 bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>"$OUT/bench_R4096.err" | grep '^{' | tail -1 > "$OUT/bench_R4096.json"
# (the one-replica latency leg is left out of the profiled command: its launches of the same kernels would dilute the
#  per-launch averages of the counters)
CMD="python3 bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-single-system --no-parity-check"
for t in trace fetch write sq; do rm -rf "$OUT/$t"; done
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write -- $CMD > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$OUT/sq" -o sq -- $CMD > "$OUT/sq.log" 2>&1
for t in trace fetch write sq; do
  db=$(find "$OUT/$t" -name "*.db" | head -1)
  [ -n "$db" ] && python3 tools/rocpd_summary.py "$db" "$OUT/${t}_summary.txt"
done
fdb=$(find "$OUT/fetch" -name "*.db" | head -1); wdb=$(find "$OUT/write" -name "*.db" | head -1)
sdb=$(find "$OUT/sq" -name "*.db" | head -1)
PAIRS=$(python3 -c "import json; d=json.load(open('$OUT/bench_R4096.json')); p=d['roofline']['igraph']['pair_evaluations_per_launch']; print('igraph_bwd:rotamer=%r,igraph_fwd:rotamer=%r' % (p, p))")
rm -f "$OUT/hbm_traffic.json"
python3 tools/hbm_traffic.py "$fdb" "$wdb" syn300_10A 4096 "$OUT/hbm_traffic.json" "$sdb" "$PAIRS" > "$OUT/hbm_traffic.txt" 2>&1
# keep the merge small: drop the databases
find "$OUT" -name "*.db" -delete
# the bench lines below (mode all) then read THIS table: bench.py takes profiles/hbm_traffic.json, stamped with the sources it was measured on
[ -s "$OUT/hbm_traffic.json" ] && cp "$OUT/hbm_traffic.json" profiles/hbm_traffic.json
fi

if want bench; then
for R in 1 8 64 256 1024 4096; do
  st=100; [ $R -le 64 ] && st=300
  python3 bench.py --replicas $R --steps $st --warmup 30 2>"$OUT/bench_R$R.err" | grep '^{' | tail -1 > "$OUT/bench_R$R.json"
done
fi

if want configs; then
# other BASELINE configurations
for w in trpcage20_7A proteinG56_7A syn150_10A syn300_7A syn300_10A; do
for R in 1 8 64 512; do
  st=300; [ $R -ge 64 ] && st=150
  python3 bench.py --workload $w --replicas $R --steps $st --warmup 40 --no-cpu-baseline --no-single-system --no-parity-check 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('$w', $R, round(d['value']))"
done; done > "$OUT/other_configs.txt"
# BASELINE configs[3] and [4] on one GPU (the multi-GPU lines need a node the builder cannot launch on)
# BASELINE configs[1]: ONE protein G, with the reference timed on one host core in the same line; and the batch of 8
for R in 1 8; do
  python3 bench.py --workload proteinG56_7A --replicas $R --steps 400 --warmup 50 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_proteinG56_7A_R$R.json"
done
for w in remd64_proteinG56 ens512_syn150; do
  python3 bench.py --workload $w --steps 1665 --warmup 111 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_$w.json"
done
fi
ls -la "$OUT"
