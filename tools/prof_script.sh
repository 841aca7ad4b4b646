#!/bin/bash
# usage (GPU box, repo root): tools/prof_script.sh <tag> <python script> [args]  -> top kernels of that script
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o s --output-format csv -- python "$@" > gpurun_out/prof_$tag/run.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_$tag/s_kernel_stats.csv")))
for r in rows[:24]:
    print(r["Name"][:110].ljust(110), r["Calls"].rjust(6), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(9), r["Percentage"].rjust(6))
PY
