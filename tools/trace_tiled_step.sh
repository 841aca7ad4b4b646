#!/bin/bash
# development aid (GPU box, repo root): kernel timeline of the last step of tools/time_tiled_one.py (TB = batch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cp
rocprofv3 --kernel-trace -d gpurun_out/cp -o k --output-format csv -- python tools/time_tiled_one.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/cp/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %7.1f q%s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id"), r["Kernel_Name"].replace("rulgnn::", "")[:70]))
PY
find gpurun_out/cp -name "*.csv" -delete
