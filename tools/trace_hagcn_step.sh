#!/bin/bash
# usage (GPU box, repo root): tools/trace_hagcn_step.sh <tag> -- queue / start / duration of the kernels between two graph forwards of HAGCN steps
tag=${1:-hagcn_trace}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace -d gpurun_out/$tag/kt -o k --output-format csv -- python bench.py --family HAGCN --steps 6 --warmup 3 --no-roofline --no-cpu-baseline > gpurun_out/$tag/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/$tag/kt/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "hg_forward_kernel" in r["Kernel_Name"]]
lo, hi = idx[-2], idx[-1]
t0 = int(rows[lo]["Start_Timestamp"])
out = open("gpurun_out/$tag/timeline.txt", "w")
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("rulgnn::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:50]
    out.write(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} q{r.get('Queue_Id','?')} {name}\n")
out.close()
PY
find gpurun_out/$tag -name "*.csv" -delete
