#!/bin/bash
# usage (GPU box, repo root): tools/trace_family_step.sh <FAMILY> <tag>  -- kernel trace (start / end / queue) of bench.py --family steps; prints the last step's timeline
fam=${1:-FC_STGNN}; tag=${2:-trace}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace -d gpurun_out/$tag/kt -o k --output-format csv -- python bench.py --family $fam --steps 6 --warmup 3 --no-roofline --no-cpu-baseline > gpurun_out/$tag/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/$tag/kt/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of the step's last kernel (adam) marks a step end; take the rows between the two last adam kernels
idx = [i for i, r in enumerate(rows) if "adam_step" in r["Kernel_Name"] or "multi_tensor" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
out = open("gpurun_out/$tag/timeline.txt", "w")
prev_end = {}
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r.get("Queue_Id", "?")
    name = r["Kernel_Name"].replace("rulgnn::", "").replace("(anonymous namespace)::", "")[:60]
    out.write(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f} q{q} {name}\n")
out.close()
print(open("gpurun_out/$tag/timeline.txt").read())
PY
find gpurun_out/$tag -name "*.csv" -delete
