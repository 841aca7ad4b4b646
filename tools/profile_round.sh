#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# kernel-trace stats of the default bench, then separate PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE)
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/stats -o s --output-format csv -- python bench.py --steps 20 --warmup 5 --no-families --no-rmse > gpurun_out/$tag/bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/$tag/fetch -o f --output-format csv -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-families --no-rmse --no-roofline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/$tag/write -o w --output-format csv -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-families --no-rmse --no-roofline > /dev/null 2>&1
python tools/hbm_traffic_report.py gpurun_out/$tag gpurun_out/$tag/hbm_traffic.json 65536
find gpurun_out/$tag -name "*kernel_trace.csv" -delete; find gpurun_out/$tag -name "*counter_collection.csv" -delete
grep '^{"metric"' gpurun_out/$tag/bench.log | tail -1 > gpurun_out/$tag/bench.json
ls gpurun_out/$tag/*
