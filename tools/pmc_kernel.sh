#!/bin/bash
# usage (GPU box, repo root): tools/pmc_kernel.sh <tag> <python script> [args]
# SQ counters per kernel (two passes: rocprofv3 takes 8 SQ counters per pass); summary by tools/pmc_kernel_report.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmc_${tag}_A -o p --output-format csv -- python "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA -d gpurun_out/pmc_${tag}_B -o p --output-format csv -- python "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 -d gpurun_out/pmc_${tag}_C -o p --output-format csv -- python "$@" > /dev/null 2>&1
python tools/pmc_kernel_report.py gpurun_out/pmc_${tag}_A gpurun_out/pmc_${tag}_B gpurun_out/pmc_${tag}_C | tee gpurun_out/pmc_${tag}.txt
