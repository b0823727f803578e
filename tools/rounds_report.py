"""rounds of workgroups a FeaSt launch pays against the rounds of work it has, per launch of a training
step, from a rocprofv3 kernel trace of bench.py (grid size / workgroup size per dispatch; resident workgroups per CU from the
kernels' LDS footprint: 16-row fused kernels 4, 32-row and the 128-channel backward 2).
    python tools/rounds_report.py <trace dir> > profiles/r04_rounds_report.txt"""
import csv, glob, sys, collections, math
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0])))
CUS = 256
def short(n): return n.replace('geobi::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
def per_cu(name, lds):
    # the trace does not carry the dynamic LDS size: resident workgroups per CU by kernel family
    if name.startswith('feast_rowpass_fused128'):
        return 2                       # 54 KB of LDS, nine waves: two workgroups per CU
    return 4 if name.rstrip('>').split(',')[-1].strip().startswith('16') or 'rowpass_fused_kernel' in name else 2
agg = collections.OrderedDict()
for r in rows:
    n = short(r['Kernel_Name'])
    if not (n.startswith('feast_fused_kernel') or n.startswith('feast_rowpass_fused')):
        continue
    wg = int(r['Workgroup_Size']) if 'Workgroup_Size' in r else int(r['Workgroup_Size_X'])
    grid = int(r['Grid_Size']) if 'Grid_Size' in r else int(r['Grid_Size_X'])
    tiles = grid // wg
    lds = r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', 0))
    slots = CUS * per_cu(n, lds)
    key = (n, tiles)
    a = agg.setdefault(key, [0, 0.0, slots, wg, lds])
    a[0] += 1
    a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('%-46s %7s %6s %6s %7s %6s %8s' % ('kernel', 'tiles', 'slots', 'work', 'paid', 'fill', 'avg us'))
for (n, tiles), (calls, us, slots, wg, lds) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    work = tiles / slots
    paid = math.ceil(work)
    print('%-46s %7d %6d %6.2f %7d %6.2f %8.1f   (%d launches, %d threads)' % (n[:46], tiles, slots, work, paid, work / paid, us / calls, calls, wg))
