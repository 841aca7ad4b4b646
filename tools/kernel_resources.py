"""Per-kernel register / spill report of one HIP source (development aid).
    python tools/kernel_resources.py gnn_rul_benchmarking_amd/csrc/stgcn_train.hip [name-filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.basename(src), "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, cwd=os.path.dirname(src) or ".", capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^ ]+ +(Function Name|Name): (\S+)", line) or re.search(r"(Function Name|Name): (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("rulgnn::", "").replace("void ", ""))}
        rows.append(cur)
        continue
    m = re.search(r"(SGPRs|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" [")[0]] = int(m.group(2))
print(f"{'kernel':70s} {'SGPR':>5s} {'sSpill':>6s} {'VGPR':>5s} {'AGPR':>5s} {'vSpill':>6s} {'scratch':>7s} {'occ':>4s} {'LDS':>6s}")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:70]:70s} {r.get('SGPRs',0):5d} {r.get('SGPRs Spill',0):6d} {r.get('VGPRs',0):5d} {r.get('AGPRs',0):5d} "
              f"{r.get('VGPRs Spill',0):6d} {r.get('ScratchSize',0):7d} {r.get('Occupancy',0):4d} {r.get('LDS Size',0):6d}")
