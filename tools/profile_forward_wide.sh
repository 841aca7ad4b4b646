#!/bin/bash
# usage (GPU box, repo root): tools/profile_forward_wide.sh <tag>  -- the eval forward at the PHM2012 wiring (40 patches x 64 points,
# stgcn_forward_mxw_kernel + its scanning launch), batch 131072: rocprofv3 kernel stats, then FETCH_SIZE / WRITE_SIZE in separate passes
tag=${1:-r03}
export NP=40 PS=64
B=131072
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=gpurun_out/$tag/fwdw_$B
mkdir -p $d
rocprofv3 --kernel-trace --stats -d $d/stats -o s --output-format csv -- python tools/run_forward_once.py $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d/fetch -o f --output-format csv -- python tools/run_forward_once.py $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $d/write -o w --output-format csv -- python tools/run_forward_once.py $B > /dev/null 2>&1
python tools/hbm_traffic_report.py $d $d/hbm_traffic.json $B
python tools/trim_stats.py $(find $d/stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_forward_phm2012_40x64_bs${B}_kernel_stats.csv
cp $d/hbm_traffic.json gpurun_out/${tag}_forward_phm2012_40x64_bs${B}_hbm_traffic.json
head -5 gpurun_out/${tag}_forward_phm2012_40x64_bs${B}_kernel_stats.csv
