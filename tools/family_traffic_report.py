"""HBM bytes per launch of every kernel of the family benches from the FETCH_SIZE / WRITE_SIZE passes of tools/prof_families.sh.
    python tools/family_traffic_report.py gpurun_out/<tag> profiles/<name>.json FAMILY [FAMILY ...]
Counters are in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE is taken as is."""
import collections
import csv
import glob
import json
import sys

tag, out, fams = sys.argv[1], sys.argv[2], sys.argv[3:]


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("rulgnn::", "").strip()


def collect(d, counter):
    res = collections.defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "rulgnn" in r["Kernel_Name"]:
                res[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in res.items()}


doc = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --family F --no-roofline`; counters in KB; "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); averages over all launches of a kernel "
               "name (a generic GEMM kernel serves several shapes)", "families": {}}
for fam in fams:
    fetch, write = collect(f"{tag}/{fam}/fetch", "FETCH_SIZE"), collect(f"{tag}/{fam}/write", "WRITE_SIZE")
    doc["families"][fam] = {"kernels": {k: {"fetch_kb_raw": fetch[k], "write_kb_raw": write.get(k, 0.0),
                                            "hbm_bytes_per_launch": (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0} for k in sorted(fetch)}}
json.dump(doc, open(out, "w"), indent=1)
print("wrote", out)
