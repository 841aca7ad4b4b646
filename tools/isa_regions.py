"""Instruction census of the main tile loop of a matrix-core kernel, split at its s_setprio markers (stage boundaries).
    python tools/isa_regions.py file.s <mangled-name-regex>"""
import collections
import re
import sys


def cls(ins):
    if ins.startswith('v_mfma_f32_16x16x32'): return 'mfma_f16'
    if ins.startswith('v_mfma'): return 'mfma_f32'
    if ins.startswith('v_cvt_pk'): return 'cvt_pk'
    if '_dpp' in ins or 'permlane' in ins: return 'dpp'
    if ins.startswith(('v_max', 'v_min')): return 'maxmin'
    if ins.startswith(('v_rcp', 'v_rsq', 'v_sqrt')): return 'trans'
    if ins.startswith('v_cndmask') or ins.startswith('v_cmp'): return 'cmp/sel'
    if ins.startswith(('v_and', 'v_or', 'v_xor', 'v_lshr', 'v_lshl', 'v_perm', 'v_bfi', 'v_bfe')): return 'bitops'
    if ins.startswith(('v_mov', 'v_accvgpr', 'v_readlane', 'v_readfirstlane')): return 'mov'
    if ins.startswith('v_'): return 'valu_fp'
    if ins.startswith('ds_'): return 'lds'
    if ins.startswith(('global_', 'scratch_', 'buffer_')): return 'vmem'
    if ins.startswith('s_nop'): return 's_nop'
    if ins.startswith('s_waitcnt'): return 'waitcnt'
    return 'salu'


VALU = ('cvt_pk', 'dpp', 'maxmin', 'trans', 'cmp/sel', 'bitops', 'mov', 'valu_fp')
s = open(sys.argv[1]).read().split('\n')
a = b = None
for i, ln in enumerate(s):
    if a is None and re.match(r'^' + sys.argv[2] + r'\S*:', ln): a = i
    if a is not None and b is None and ln.strip().startswith('s_endpgm'): b = i
body = s[a:b]
labels = {m.group(1): i for i, ln in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', ln)] if m}
first = next(i for i, ln in enumerate(body) if 'v_mfma_f32_16x16x32' in ln)
best = None
for i, ln in enumerate(body):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)', ln)
    if m:
        lab = m.group(1) or m.group(2)
        if lab in labels and labels[lab] <= first <= i and (best is None or i - labels[lab] < best[1] - best[0]):
            best = (labels[lab], i)
loop = body[best[0]:best[1]]
regions = [[]]
for ln in loop:
    t = ln.strip()
    if t.startswith('s_setprio'): regions.append([])
    regions[-1].append(t)
print('loop lines', len(loop), 'regions', len(regions))
tot = collections.Counter()
for k, reg in enumerate(regions):
    c = collections.Counter()
    for t in reg:
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
        c[cls(t.split()[0])] += 1
    tot.update(c)
    print(k, 'VALU', sum(c[x] for x in VALU), dict(sorted(c.items())))
print('total VALU', sum(tot[x] for x in VALU), dict(sorted(tot.items())))
