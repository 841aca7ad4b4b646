#!/bin/bash
# usage (GPU box, repo root): tools/profile_r06.sh [parts]   parts: any of  round fwd mfma fam tiled  (default: all)
# Every rocprofv3 pass runs under its own timeout; counters are collected in passes of their own (--kernel-trace + --pmc only).
parts=${1:-"round fwd mfma fam tiled"}
tag=r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T="timeout 420"
if [[ $parts == *round* ]]; then
  mkdir -p gpurun_out/$tag
  $T rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/stats -o s --output-format csv -- python bench.py --steps 20 --warmup 5 --no-families --no-rmse --no-other-shapes > gpurun_out/$tag/bench.log 2>&1
  $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/$tag/fetch -o f --output-format csv -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-families --no-rmse --no-roofline > /dev/null 2>&1
  $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/$tag/write -o w --output-format csv -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-families --no-rmse --no-roofline > /dev/null 2>&1
  python tools/hbm_traffic_report.py gpurun_out/$tag gpurun_out/$tag/hbm_traffic.json 65536
  find gpurun_out/$tag -name "*kernel_trace.csv" -delete; find gpurun_out/$tag -name "*counter_collection.csv" -delete
  grep '^{"metric"' gpurun_out/$tag/bench.log | tail -1 > gpurun_out/$tag/bench.json
fi
if [[ $parts == *fwd* ]]; then
  for B in 65536 1048576; do
    d=gpurun_out/$tag/fwd_$B
    mkdir -p $d
    $T rocprofv3 --kernel-trace --stats -d $d/stats -o s --output-format csv -- python tools/run_forward_once.py $B > /dev/null 2>&1
    $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d/fetch -o f --output-format csv -- python tools/run_forward_once.py $B > /dev/null 2>&1
    $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $d/write -o w --output-format csv -- python tools/run_forward_once.py $B > /dev/null 2>&1
    python tools/hbm_traffic_report.py $d $d/hbm_traffic.json $B
    find $d -name "*kernel_trace.csv" -delete; find $d -name "*counter_collection.csv" -delete
  done
fi
if [[ $parts == *mfma* ]]; then
  # MFMA utilisation: SQ_VALU_MFMA_BUSY_CYCLES against the kernel's shader cycles (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs
  pm="--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES"
  $T rocprofv3 --kernel-trace $pm -d gpurun_out/pmc_${tag}_fwd -o p --output-format csv -- python tools/run_forward_once.py 1048576 > /dev/null 2>&1
  python tools/pmc_kernel_report.py gpurun_out/pmc_${tag}_fwd > gpurun_out/${tag}_forward_bs1048576_sq_counters.txt
  $T rocprofv3 --kernel-trace $pm -d gpurun_out/pmc_${tag}_train -o p --output-format csv -- python tools/run_train_steps.py 65536 12 > /dev/null 2>&1
  python tools/pmc_kernel_report.py gpurun_out/pmc_${tag}_train > gpurun_out/${tag}_train_step_sq_counters.txt
  $T rocprofv3 --kernel-trace $pm -d gpurun_out/pmc_${tag}_planes -o p --output-format csv -- python tools/run_planes_once.py > /dev/null 2>&1
  python tools/pmc_kernel_report.py gpurun_out/pmc_${tag}_planes > gpurun_out/${tag}_sgemm_planes_mfma_counters.txt
  $T rocprofv3 --kernel-trace $pm -d gpurun_out/pmc_${tag}_fc -o p --output-format csv -- python bench.py --family FC_STGNN --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python tools/pmc_kernel_report.py gpurun_out/pmc_${tag}_fc > gpurun_out/${tag}_fcstgnn_sq_counters.txt
  rm -rf gpurun_out/pmc_${tag}_fwd gpurun_out/pmc_${tag}_train gpurun_out/pmc_${tag}_planes gpurun_out/pmc_${tag}_fc
fi
if [[ $parts == *fam* ]]; then
  for fam in FC_STGNN ASTGCNN HAGCN STMSGCN; do
    d=gpurun_out/$tag/$fam
    mkdir -p $d
    $T python bench.py --family $fam --steps 50 --warmup 10 > $d/bench.log 2>&1
    grep '^{"metric"' $d/bench.log | tail -1 > $d/bench.json
    $T rocprofv3 --kernel-trace --stats -d $d/stats -o s --output-format csv -- python bench.py --family $fam --steps 50 --no-cpu-baseline --no-roofline > /dev/null 2>&1
    $T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d/fetch -o f --output-format csv -- python bench.py --family $fam --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
    $T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $d/write -o w --output-format csv -- python bench.py --family $fam --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  done
  $T python bench.py --family FC_STGNN --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/$tag/FC_STGNN/bench_bf16.log 2>&1
  grep '^{"metric"' gpurun_out/$tag/FC_STGNN/bench_bf16.log | tail -1 > gpurun_out/$tag/FC_STGNN/bench_bf16.json
  python tools/family_traffic_report.py gpurun_out/$tag gpurun_out/$tag/family_hbm_traffic.json FC_STGNN ASTGCNN HAGCN STMSGCN
  find gpurun_out/$tag -name "*kernel_trace.csv" -delete; find gpurun_out/$tag -name "*counter_collection.csv" -delete
fi
if [[ $parts == *tiled* ]]; then
  d=gpurun_out/$tag/tiled
  mkdir -p $d
  $T rocprofv3 --kernel-trace --stats -d $d/train -o s --output-format csv -- python tools/time_tiled_one.py > /dev/null 2>&1
  $T rocprofv3 --kernel-trace --stats -d $d/eval -o s --output-format csv -- python tools/run_tiled_eval.py > /dev/null 2>&1
  $T rocprofv3 --kernel-trace --stats -d $d/planes -o s --output-format csv -- python tools/time_sgemm_planes.py > $d/planes_timing.txt 2>&1
  find $d -name "*kernel_trace.csv" -delete
  $T python tools/time_tiled_step.py 1024 512 100 > $d/step_times.txt 2>&1
fi
ls -R gpurun_out/$tag | head -80
