#!/bin/bash
# copy the summaries tools/refresh_profiles.sh left under gpurun_out/prof/ into profiles/ (tracked), named per round
set -u
R=${1:-r05}
S=gpurun_out/prof; D=profiles
for n in 1 8 64 256 1024 4096; do [ -s $S/bench_R$n.json ] && cp $S/bench_R$n.json $D/${R}_bench_R$n.json; done
for w in remd64_proteinG56 ens512_syn150; do [ -s $S/bench_$w.json ] && cp $S/bench_$w.json $D/${R}_bench_$w.json; done
for n in 1 8; do [ -s $S/bench_proteinG56_7A_R$n.json ] && cp $S/bench_proteinG56_7A_R$n.json $D/${R}_bench_proteinG56_7A_R$n.json; done
[ -s $S/other_configs.txt ] && cp $S/other_configs.txt $D/${R}_bench_other_configs.txt
[ -s $S/trace_summary.txt ] && cp $S/trace_summary.txt $D/${R}_rocprof_kernel_stats_R4096.txt
[ -s $S/sq_summary.txt ] && cp $S/sq_summary.txt $D/${R}_rocprof_sq_pmc_R4096.txt
[ -s $S/fetch_summary.txt ] && cp $S/fetch_summary.txt $D/${R}_rocprof_fetch_pmc_R4096.txt
[ -s $S/write_summary.txt ] && cp $S/write_summary.txt $D/${R}_rocprof_write_pmc_R4096.txt
[ -s $S/hbm_traffic.txt ] && cp $S/hbm_traffic.txt $D/${R}_rocprof_hbm_traffic_R4096.txt
[ -s $S/hbm_traffic.json ] && cp $S/hbm_traffic.json $D/hbm_traffic.json
for t in gpurun_out/tl_*/timeline.txt; do [ -s "$t" ] && cp "$t" $D/${R}_timeline_one_step_$(basename $(dirname $t) | sed 's/^tl_//').txt; done
ls -la $D
