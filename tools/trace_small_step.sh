#!/bin/bash
# development aid (GPU box, repo root): kernel timeline of the last steps of ST_GCN.update at a small batch: tools/trace_small_step.sh [B] [NP] [PS]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cps
rocprofv3 --kernel-trace -d gpurun_out/cps -o k --output-format csv -- python tools/host_vs_gpu_small.py ${1:-100} ${2:-14} ${3:-30} > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/cps/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "finalize" in r["Kernel_Name"]]
lo, hi = idx[-3] + 1, idx[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %7.1f q%s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id"), r["Kernel_Name"].replace("rulgnn::", "")[:90]))
PY
find gpurun_out/cps -name "*.csv" -delete
