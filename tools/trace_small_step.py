"""Kernel timeline of one ST_GCN.update at a small batch out of a rocprofv3 --kernel-trace run (development aid):
    rocprofv3 --kernel-trace -d OUT -o k --output-format csv -- python tools/host_vs_gpu_small.py 100 ;  python tools/trace_small_step.py OUT"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/k_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "finalize" in r["Kernel_Name"]]
lo, hi = idx[-12] + 1, idx[-11] + 1
t0 = int(rows[lo]["Start_Timestamp"])
prev = None
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = "" if prev is None else f"gap {(s - prev) / 1e3:5.1f}"
    prev = e
    print("%8.1f %8.1f %7.1f  %-10s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"].replace("rulgnn::", "")[:60]))
