"""HBM bytes per launch of the ST_GCN phase kernels from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh.
    python tools/hbm_traffic_report.py gpurun_out/<tag> profiles/<name>.json [batch]
Counters are in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B),
WRITE_SIZE is taken as is."""
import csv, collections, glob, json, os, re, sys
tag, out = sys.argv[1], sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
def collect(sub, counter):
    res = collections.defaultdict(list)
    for f in glob.glob(f"{tag}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"stgcn_train_phase_kernel<(\d+), (\d), (\d), (\d)(?:, \d+)*>", r["Kernel_Name"])
            mx = re.search(r"stgcn_train_mx_kernel<(\d+), (\d), (\d), (\d+)>", r["Kernel_Name"])      # matrix-core chain (round 4): <L, KIND, IDX, NFIX>
            mxw = re.search(r"stgcn_train_mxw_kernel<(\d+), (\d), (\d), (\d+)(?:, \d+)?>", r["Kernel_Name"])    # its wide form: <L, KIND, IDX, NT>
            wide = int(os.environ.get("NP", 14)) >= 16                  # which chain this report is about (the default bench also runs 40 x 64 lines)
            if (mxw or "stgcn_train_f0_mxw_kernel" in r["Kernel_Name"]) and not wide:
                continue
            if (mx or "stgcn_train_f0_mx_kernel" in r["Kernel_Name"]) and wide:
                continue
            if m and (m.group(1) == "64") != wide:                      # the fp32 chain's row width: 16 at num_patch <= 16, else 64
                continue
            if mxw:
                CHAIN.add("mx")
                name = {"0": "F", "1": "TOP", "2": "G"}[mxw.group(2)] + (mxw.group(3) if mxw.group(2) != "1" else "")
            elif "stgcn_train_f0_mxw_kernel" in r["Kernel_Name"]:
                CHAIN.add("mx")
                name = "F0"
            elif mx:
                CHAIN.add("mx")
                name = {"0": "F", "1": "TOP", "2": "G"}[mx.group(2)] + (mx.group(3) if mx.group(2) != "1" else "")
            elif m:
                name = {"0": "F", "1": "TOP", "2": "G"}[m.group(3)] + (m.group(4) if m.group(3) != "1" else "")
            elif "stgcn_train_f0_mx_kernel" in r["Kernel_Name"]:      # phase F_0 on the matrix cores (round 3)
                name = "F0"
            elif "stgcn_forward_mx_kernel<2, 14, 30" in r["Kernel_Name"] or "stgcn_forward_eval" in r["Kernel_Name"]:
                name = "EVAL"
            elif "stgcn_forward_mxw_kernel" in r["Kernel_Name"]:       # wide matrix-core eval forward (16 <= num_patch <= 47)
                name = "EVAL_WIDE"
            elif "stgcn_forward_fixup_kernel" in r["Kernel_Name"]:
                name = "EVAL_WIDE_SCAN"
            else:
                continue
            res[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in res.items()}
CHAIN = set()
fetch, write = collect("fetch", "FETCH_SIZE"), collect("write", "WRITE_SIZE")
kern = {}
for k in fetch:
    b = (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0
    kern[k] = {"fetch_kb_raw": fetch[k], "write_kb_raw": write.get(k, 0.0), "hbm_bytes_per_launch": b, "hbm_bytes_per_sample": b / batch}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh); counters are in KB; "
                   "FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE taken as is",
           "chain": "mx" if "mx" in CHAIN else "fp32",
           "workload": {"num_patch": int(os.environ.get("NP", 14)), "patch_size": int(os.environ.get("PS", 30)), "batch": batch}, "kernels": kern}, open(out, "w"), indent=1)
for k, v in kern.items():
    print(f"{k:5s} {v['hbm_bytes_per_sample']:8.0f} B/sample")
