"""Per-kernel launch count / average / minimum duration out of a rocprofv3 (rocpd sqlite) output directory: python tools/rocpd_kernel_stats.py DIR"""
import glob
import os
import sqlite3
import sys

for path in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 from {kd} d "
         f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 5 desc")
    print("kernel,calls,avg_us,min_us,total_us")
    for name, n, avg, mn, tot in cur.execute(q):
        print(f"\"{name}\",{n},{avg:.2f},{mn:.2f},{tot:.1f}")
