#!/bin/bash
# usage (GPU box, repo root): NP=40 PS=64 tools/prof_train_pmc.sh <tag> [batch] [steps]
# kernel stats of a plain loop of ST_GCN.update steps, then FETCH_SIZE / WRITE_SIZE in their own passes (tools/hbm_traffic_report.py reads them)
tag=${1:-r04w}; B=${2:-16384}; S=${3:-60}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/stats -o s --output-format csv -- python tools/run_train_steps.py $B $S > gpurun_out/$tag/run.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/$tag/fetch -o f --output-format csv -- python tools/run_train_steps.py $B 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/$tag/write -o w --output-format csv -- python tools/run_train_steps.py $B 6 > /dev/null 2>&1
python tools/hbm_traffic_report.py gpurun_out/$tag gpurun_out/$tag/hbm_traffic.json $B
find gpurun_out/$tag -name "s_kernel_stats.csv" -exec cp {} gpurun_out/$tag/kernel_stats.csv \;
find gpurun_out/$tag -name "*_kernel_trace.csv" -delete; find gpurun_out/$tag -name "*counter_collection.csv" -delete
