#!/bin/bash
# build container: copy what tools/refresh_profiles.sh left under gpurun_out/prof/ into profiles/ under this round's names
# usage: tools/copy_profiles.sh [tag, default r04]
set -u
T=${1:-r04}; P=gpurun_out/prof
for R in 1 8 64 256 1024 4096; do cp $P/bench_R$R.json profiles/${T}_bench_R$R.json; done
for w in remd64_proteinG56 ens512_syn150 proteinG56_7A_R1 proteinG56_7A_R8; do cp $P/bench_$w.json profiles/${T}_bench_$w.json; done
cp $P/other_configs.txt profiles/${T}_bench_other_configs.txt
cp $P/trace_summary.txt profiles/${T}_rocprof_kernel_stats_R4096.txt
cp $P/fetch_summary.txt profiles/${T}_rocprof_fetch_pmc_R4096.txt
cp $P/write_summary.txt profiles/${T}_rocprof_write_pmc_R4096.txt
cp $P/sq_summary.txt profiles/${T}_rocprof_sq_pmc_R4096.txt
cp $P/hbm_traffic.txt profiles/${T}_rocprof_hbm_traffic_R4096.txt
cp $P/hbm_traffic.json profiles/hbm_traffic.json
