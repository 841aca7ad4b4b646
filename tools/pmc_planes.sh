#!/bin/bash
# SQ + memory-side counters of the pre-split product kernel at the theta shape (GPU box, repo root): tools/pmc_planes.sh <tag>
tag=${1:-planes}
bash tools/pmc_kernel.sh $tag tools/run_planes_once.py | grep -A30 "sgemm_planes_kernel"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d gpurun_out/pmc_${tag}_D -o p --output-format csv -- python tools/run_planes_once.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum -d gpurun_out/pmc_${tag}_E -o p --output-format csv -- python tools/run_planes_once.py > /dev/null 2>&1
python tools/pmc_kernel_report.py gpurun_out/pmc_${tag}_D gpurun_out/pmc_${tag}_E | grep -A12 "sgemm_planes_kernel" | tee -a gpurun_out/pmc_${tag}.txt
