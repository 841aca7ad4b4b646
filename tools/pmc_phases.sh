#!/bin/bash
# collect SQ counters for the training phase kernels (two passes); run via gpurun from the repo root
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmcA -o p --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d gpurun_out/pmcB -o p --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
