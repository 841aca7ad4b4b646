#!/bin/bash
# SQ + cache counters of the pre-split product kernel at the theta shape (GPU box, repo root): tools/pmc_planes.sh <tag>
# (every pass under its own timeout: a pass with TA_BUSY_avr / TCP_PENDING_STALL_CYCLES_sum did not return within 20 minutes on this pool)
tag=${1:-planes}
timeout 600 bash tools/pmc_kernel.sh $tag tools/run_planes_once.py | grep -A30 "sgemm_planes_kernel"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d gpurun_out/pmc_${tag}_D -o p --output-format csv -- python tools/run_planes_once.py > /dev/null 2>&1
python tools/pmc_kernel_report.py gpurun_out/pmc_${tag}_D | grep -A8 "sgemm_planes_kernel" | tee -a gpurun_out/pmc_${tag}.txt
