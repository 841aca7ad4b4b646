#!/bin/bash
# usage (GPU box, repo root): tools/prof_family.sh FAMILY [extra bench args]  -> kernel stats of `bench.py --family FAMILY`
fam=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$fam
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$fam -o s --output-format csv -- python bench.py --family $fam --no-cpu-baseline --steps 50 "$@" > gpurun_out/prof_$fam/bench.log 2>&1
grep '^{"metric"' gpurun_out/prof_$fam/bench.log | tail -1 | cut -c1-220
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_$fam/s_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step (us): %.1f" % (tot/55/1e3))
for r in rows[:26]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(6), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(8), r["Percentage"].rjust(6))
PY
