import csv, collections, re, sys
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ['gpurun_out/pmcA/p_counter_collection.csv','gpurun_out/pmcB/p_counter_collection.csv']:
    for r in csv.DictReader(open(f)):
        m = re.search(r'stgcn_train_phase_kernel<\d+, (\d), (\d), (\d)(?:, \d+)*>', r['Kernel_Name'])
        if m: name = {'0':'F','1':'TOP','2':'G'}[m.group(2)] + m.group(3)
        elif 'stgcn_forward_mx_kernel' in r['Kernel_Name'] or 'stgcn_forward_eval' in r['Kernel_Name']: name = 'EVAL'
        else: continue
        res[name][r['Counter_Name']].append(float(r['Counter_Value']))
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
print(f"{'k':5s} {'VALU/tile':>9s} {'SALU':>6s} {'LDS':>5s} {'SMEM':>5s} {'VMEM':>5s} {'wavecyc(q)':>10s} {'actVALUq':>9s} {'waitInst':>9s} {'waitAny':>9s} {'actAny':>8s} {'waves':>6s}")
for k in ['F0','F1','F2','F3','TOP0','G3','G2','G1','G0','EVAL']:
    d = {c: sum(v)/len(v) for c,v in res[k].items()}
    if not d: continue
    print(f"{k:5s} {d['SQ_INSTS_VALU']/tiles:9.0f} {d['SQ_INSTS_SALU']/tiles:6.0f} {d['SQ_INSTS_LDS']/tiles:5.0f} {d['SQ_INSTS_SMEM']/tiles:5.0f} {d['SQ_INSTS_VMEM_RD']/tiles:5.0f} {d['SQ_WAVE_CYCLES']/tiles:10.0f} {d['SQ_ACTIVE_INST_VALU']/tiles:9.0f} {d['SQ_WAIT_INST_ANY']/tiles:9.0f} {d['SQ_WAIT_ANY']/tiles:9.0f} {d['SQ_ACTIVE_INST_ANY']/tiles:8.0f} {d['SQ_WAVES']:6.0f}")
