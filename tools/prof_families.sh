#!/bin/bash
# usage (GPU box, repo root): tools/prof_families.sh <tag> FAMILY...   per family: the bench line (with its live dominant-kernel roofline),
# rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes (HBM traffic per kernel)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for fam in "$@"; do
  d=gpurun_out/$tag/$fam
  mkdir -p $d
  python bench.py --family $fam --steps 50 --warmup 10 > $d/bench.log 2>&1
  grep '^{"metric"' $d/bench.log | tail -1 > $d/bench.json
  rocprofv3 --kernel-trace --stats -d $d/stats -o s --output-format csv -- python bench.py --family $fam --steps 50 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $d/fetch -o f --output-format csv -- python bench.py --family $fam --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $d/write -o w --output-format csv -- python bench.py --family $fam --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
python tools/family_traffic_report.py gpurun_out/$tag gpurun_out/$tag/family_hbm_traffic.json "$@"
