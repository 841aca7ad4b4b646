#!/bin/bash
# usage (build container, after tools/profile_forward.sh <tag> and tools/profile_round.sh <tag> ran on the GPU box and gpurun merged
# gpurun_out/): tools/collect_profiles.sh <tag>  -- copies the summaries the judge reads into profiles/<tag>_*
tag=${1:-r03}
o=gpurun_out/$tag
python tools/trim_stats.py $o/stats/s_kernel_stats.csv profiles/${tag}_train_step_kernel_stats.csv
cp $o/hbm_traffic.json profiles/${tag}_hbm_traffic.json
cp $o/bench.json profiles/${tag}_train_step_bench_under_rocprof.json
for B in 65536 1048576; do
  python tools/trim_stats.py $o/fwd_$B/stats/s_kernel_stats.csv profiles/${tag}_forward_bs${B}_kernel_stats.csv
  cp $o/fwd_$B/hbm_traffic.json profiles/${tag}_forward_bs${B}_hbm_traffic.json
done
[ -f gpurun_out/pmc_${tag}_fwd1m.txt ] && cp gpurun_out/pmc_${tag}_fwd1m.txt profiles/${tag}_forward_bs1048576_sq_counters.txt
ls -la profiles/${tag}_*
