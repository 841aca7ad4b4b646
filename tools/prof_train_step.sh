#!/bin/bash
# usage (GPU box, repo root): tools/prof_train_step.sh <tag> [batch] [steps]  -- rocprofv3 kernel stats of a plain loop of ST_GCN.update steps
tag=${1:-r04a}; B=${2:-65536}; S=${3:-100}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/stats -o s --output-format csv -- python tools/run_train_steps.py $B $S > gpurun_out/$tag/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/$tag/stats/**/s_kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in list(csv.DictReader(open(f)))[:18]:
    print(r["Name"][:86].ljust(86), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(8), "us", r["Percentage"].rjust(6))
PY
tail -2 gpurun_out/$tag/run.log
