"""Mean SQ counters per kernel from rocprofv3 --pmc passes (csv): python tools/pmc_kernel_report.py dirA dirB ..."""
import collections
import csv
import glob
import sys

res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            res[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in res.items():
    if not any(t in k for t in ("stgcn", "rulgnn")):
        continue
    print(k[:150])
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} {sum(v) / len(v):16.1f}   (n={len(v)})")
