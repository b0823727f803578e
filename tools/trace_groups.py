"""Overlap of the mesh groups in a rocprofv3 kernel trace of `bench.py --groups G` (steps delimited by the bucket-sum
kernel, once per step):  python tools/trace_groups.py <dir> [nsteps]"""
import csv, glob, sys, collections
d = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = list(csv.DictReader(open(glob.glob(d + '/*/*kernel_trace.csv')[0])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Stream_Id'], r['Kernel_Name'], r.get('Queue_Id', '')) for r in rows)
marks = [e for s, e, st, n, q in iv if 'sum_buckets_list_kernel' in n]
t0, t1 = marks[-nsteps - 1], marks[-1]
win = [x for x in iv if t0 <= x[0] < t1]
span = (t1 - t0) / 1e6 / nsteps
print('%d steps, %.3f ms per step (trace), %.0f launches per step' % (nsteps, span, len(win) / nsteps))
per = collections.defaultdict(list)
for s, e, st, n, q in win: per[(st, q)].append((s, e))
for k, l in sorted(per.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    print('  stream %-3s queue %-3s kernel time %.3f ms per step (%.0f launches)' % (k[0], k[1], sum(e - s for s, e in l) / 1e6 / nsteps, len(l) / nsteps))
# concurrency profile: time with k kernels running
ev = []
for s, e, st, n, q in win: ev += [(s, 1), (e, -1)]
ev.sort()
lvl, last, hist = 0, t0, collections.Counter()
for t, dlt in ev:
    hist[lvl] += t - last; last = t; lvl += dlt
hist[lvl] += t1 - last
tot = sum(hist.values())
print('  time with k kernels running (ms per step): ' + '  '.join('%d: %.3f' % (k, v / 1e6 / nsteps) for k, v in sorted(hist.items())))
fam = collections.defaultdict(lambda: [0, 0])
for s, e, st, n, q in win:
    k = n.replace('geobi::(anonymous namespace)::', '').replace('geobi::', '').replace('void ', '').split('(')[0][:60]
    fam[k][0] += e - s; fam[k][1] += 1
print('  kernel time by kernel (ms per step, launches per step, avg us):')
for k, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:28]:
    print('    %-60s %.3f  %5.1f  %7.1f' % (k, t / 1e6 / nsteps, c / nsteps, t / 1e3 / c))
print('  sum of kernel time %.3f ms per step' % (sum(v[0] for v in fam.values()) / 1e6 / nsteps))
