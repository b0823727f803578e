"""Instruction mix per kernel from a rocprofv3 --pmc pass (SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES
SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES ...): instructions per wave and busy shares.
   python tools/pmc_insts.py <dir> [name filter]"""
import collections, csv, glob, os, re, sys
path = max(glob.glob(sys.argv[1] + '/*/*counter_collection.csv'), key=os.path.getmtime)
flt = sys.argv[2] if len(sys.argv) > 2 else 'feast|head|gemm_tn'
acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = set()
def short(name):
    m = re.search(r'(\w+_kernel)(<[^(]*>)?', name)
    return (m.group(1) + (m.group(2) or '')).replace(' ', '') if m else name[:60]
for r in csv.DictReader(open(path)):
    k = short(r['Kernel_Name'])
    if not re.search(flt, k): continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); acc[k]['ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); acc[k]['launches'] += 1
names = sorted({c for v in acc.values() for c in v} - {'ns', 'launches'})
print('counters:', names)
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]['ns']):
    w = max(c.get('SQ_WAVES', 0.0), 1.0)
    cyc = c['ns'] * 2.4 * 1024          # SIMD-cycles of the launches at 2.4 GHz
    line = '%-46s %5d launches %7.1f us  waves %9.0f' % (k[:46], c['launches'], c['ns'] / 1e3 / c['launches'], w / c['launches'])
    for n in names:
        if n.startswith('SQ_INSTS'): line += '  %s/wave %.0f' % (n[9:], c[n] / w)
    for n in ('SQ_ACTIVE_INST_VALU', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INST_CYCLES_VMEM', 'SQ_ACTIVE_INST_LDS'):
        if n in c: line += '  %s %.3f' % (n[3:], c[n] * (4 if n == 'SQ_ACTIVE_INST_VALU' else 1) / cyc)
    print(line)
