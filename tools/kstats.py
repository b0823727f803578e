"""Per-step kernel time table from a rocprofv3 --kernel-trace --stats run (csv): python tools/kstats.py <dir> [steps] [top]"""
import csv, glob, os, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13; top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows = list(csv.DictReader(open(max(glob.glob(d + '/*/*kernel_stats.csv'), key=os.path.getmtime))))   # newest run of the directory
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per step %.3f ms, launches per step %.0f' % (tot / 1e6 / steps, sum(int(r['Calls']) for r in rows) / steps))
for r in rows[:top]:
    name = r['Name'].replace('geobi::(anonymous namespace)::', '').replace('void ', '')
    name = name.split('(')[0]
    print('%-52s calls/step %5.1f  avg %7.1f us  per step %7.1f us  %4.1f%%' % (name[:52], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3 / steps, float(r['Percentage'])))
