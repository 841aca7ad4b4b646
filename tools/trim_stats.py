"""Copy a rocprofv3 --stats kernel summary into profiles/ with over-long (torch template) kernel names cut."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.reader(open(src)))
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0] if len(r[0]) <= 160 else r[0][:157] + "..."
        w.writerow(r)
print("wrote", dst, len(rows) - 1, "kernels")
