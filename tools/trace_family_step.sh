#!/bin/bash
# development aid (GPU box, repo root): kernel timeline of one step of a family's update(): tools/trace_family_step.sh FAMILY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cpf
rocprofv3 --kernel-trace -d gpurun_out/cpf -o k --output-format csv -- python tools/host_vs_gpu.py ${1:-ASTGCNN} 60 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/cpf/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_" in r["Kernel_Name"]]
if len(idx) < 12:        # (HAGCN steps torch.optim.Adam: the fused multi-tensor kernel, two launches per step -- take every second one)
    idx = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]][1::2]
# a step from the steady-state loop (the second timed loop of host_vs_gpu.py): ten steps before the end
lo, hi = idx[-12] + 1, idx[-11] + 1
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f %7.1f q%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id"), r["Kernel_Name"].replace("rulgnn::", "").replace("(anonymous namespace)::", "")[:70]))
PY
find gpurun_out/cpf -name "*.csv" -delete
