"""Instruction census of a kernel's loop from hipcc -S output, weighted by the issue costs measured on gfx950
(profiles/r01_ubench_gfx950.md, gpurun_out/probe_mx.txt).  usage: isa_census.py file.s first_line last_line [inner:first:last:trips ...]"""
import collections
import re
import sys

COST = [  # (regex, cycles)
    (r"v_mfma_f32_16x16x32", 16.6), (r"v_mfma_f32_16x16x1_4b|v_mfma_f32_16x16x4", 33.0), (r"v_mfma", 32.0),
    (r"v_cvt_pk_f16_f32|v_cvt_pk_bf16", 4.2), (r"v_cvt_f32_f16", 4.2), (r"v_pk_", 4.1), (r"_dpp", 4.2), (r"_sdwa", 4.2),
    (r"v_max|v_min|v_med3", 4.2), (r"v_permlane", 7.9), (r"v_rcp|v_sqrt|v_rsq|v_div_|v_exp|v_log", 7.9),
    (r"v_mul_lo|v_mul_hi|v_mad_u64", 4.1), (r"v_cmp", 3.0), (r"v_cndmask", 3.0), (r"v_readlane|v_writelane|v_readfirstlane", 4.0),
    (r"ds_write_b128|ds_write2_b64", 13.0), (r"ds_write_b64", 6.0), (r"ds_write", 4.0), (r"ds_read_b128|ds_read2_b64", 8.0), (r"ds_read", 4.0),
    (r"global_|flat_|buffer_|scratch_", 4.0), (r"v_", 2.2), (r"s_nop", None), (r"s_waitcnt", 0.0), (r"s_", 1.0),
]


def cost(line):
    ins = line.split()[0]
    if ins.startswith("s_nop"):
        return ins, int(line.split()[1]) + 1.0
    for rx, c in COST:
        if re.search(rx, ins):
            return ins, c
    return ins, 1.0


def census(lines, mult=1.0, acc=None, cyc=None):
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":") or t.startswith("//"):
            continue
        ins, c = cost(t)
        key = re.sub(r"_e32|_e64", "", ins)
        acc[key] += mult
        cyc[key] += mult * c


def main():
    path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    inner = [tuple(int(v) for v in s.split(":")[1:]) for s in sys.argv[4:]]
    src = open(path).read().split("\n")
    acc, cyc = collections.Counter(), collections.Counter()
    i = a
    segs, pos = [], a
    for (f, l, trips) in sorted(inner):
        segs.append((pos, f - 1, 1.0)); segs.append((f, l, float(trips))); pos = l + 1
    segs.append((pos, b, 1.0))
    for f, l, m in segs:
        census(src[f - 1:l], m, acc, cyc)
    tot_i, tot_c = sum(acc.values()), sum(cyc.values())
    print(f"{tot_i:.0f} instructions, ~{tot_c:.0f} issue cycles")
    for k, v in sorted(cyc.items(), key=lambda kv: -kv[1])[:40]:
        print(f"  {k:34s} {acc[k]:7.0f} x  -> {v:8.0f} cyc ({100 * v / tot_c:4.1f} %)")


main()
