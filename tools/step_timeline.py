"""Timeline of ONE training step from a rocprofv3 kernel trace (development aid): per kernel its queue, start and duration,
so that the critical path of a two-stream step can be read off.  usage: step_timeline.py trace.csv [first_kernel_substring]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "fc_conv1_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
a, b = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
t0 = int(rows[a]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|rulgnn::|void ", "", n)
    return n.split("(")[0][:44]
prev_end = {}
for r in rows[a:b]:
    s, e, q = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Queue_Id"]
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    print(f"q{q} {'    ' * (int(q) % 4)}{s / 1e3:8.1f} +{(e - s) / 1e3:6.1f} us  gap {gap / 1e3:5.1f}  {short(r['Kernel_Name'])}")
print(f"step: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {b - a} kernels")
