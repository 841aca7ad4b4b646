#!/bin/bash
# usage (GPU box, repo root): tools/profile_forward.sh <tag>   -- the fused eval forward kernels alone, batch 65536 and 1M:
# rocprofv3 kernel stats, then FETCH_SIZE / WRITE_SIZE in separate passes (HBM traffic per launch), then the SQ counter passes
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 65536 1048576; do
  d=gpurun_out/$tag/fwd_$B
  mkdir -p $d
  rocprofv3 --kernel-trace --stats -d $d/stats -o s --output-format csv -- python tools/run_forward_once.py $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d/fetch -o f --output-format csv -- python tools/run_forward_once.py $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $d/write -o w --output-format csv -- python tools/run_forward_once.py $B > /dev/null 2>&1
  python tools/hbm_traffic_report.py $d $d/hbm_traffic.json $B
done
bash tools/pmc_kernel.sh ${tag}_fwd1m tools/run_forward_once.py 1048576 > /dev/null 2>&1
