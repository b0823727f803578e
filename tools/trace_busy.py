"""Device busy time over the steady-state steps of a rocprofv3 kernel trace of bench.py (steps are delimited by
the vertex head's forward kernel, which runs once per step):  python tools/trace_busy.py <dir> [nsteps]"""
import csv, glob, sys, collections
d = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = list(csv.DictReader(open(glob.glob(d + '/*/*kernel_trace.csv')[0])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Stream_Id'], r['Kernel_Name']) for r in rows)
marks = [s for s, e, st, n in iv if 'head_fwd_fused_kernel' in n][0::2]       # vertex head of every step
t0, t1 = marks[-nsteps - 1], marks[-1]
win = [x for x in iv if t0 <= x[0] < t1]
def union(ivs):
    tot = 0; cs = ce = None
    for s, e in sorted(ivs):
        if ce is None or s > ce:
            if ce is not None: tot += ce - cs
            cs, ce = s, e
        else: ce = max(ce, e)
    return tot + (ce - cs if ce is not None else 0)
per = collections.defaultdict(list)
for s, e, st, n in win: per[st].append((s, e))
print('%d steps, %.3f ms per step wall (trace), %d launches per step' % (nsteps, (t1 - t0) / 1e6 / nsteps, len(win) / nsteps))
for st, l in sorted(per.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    print('  stream %-3s busy %.3f ms per step (%d launches)' % (st, sum(e - s for s, e in l) / 1e6 / nsteps, len(l) / nsteps))
u = union([(s, e) for s, e, _, _ in win])
print('  device busy (union) %.3f ms per step; idle %.3f ms per step' % (u / 1e6 / nsteps, ((t1 - t0) - u) / 1e6 / nsteps))
main = max(per, key=lambda k: len(per[k]))
gaps, prev = [], None
for s, e, st, n in win:
    if st != main: continue
    if prev is not None and s > prev: gaps.append((s - prev, n))
    prev = max(prev or 0, e)
big = sorted(gaps, reverse=True)[:int(12 * nsteps)]
print('  main stream: %d gaps per step, total %.3f ms per step; the 12 largest per step sum to %.3f ms' %
      (len(gaps) / nsteps, sum(g[0] for g in gaps) / 1e6 / nsteps, sum(g[0] for g in big) / 1e6 / nsteps))
c = collections.Counter(n.split('(')[0].replace('geobi::(anonymous namespace)::', '').replace('void ', '')[:48] for g, n in big)
for k, v in c.most_common(6): print('     kernel after a large gap: %-48s x%.1f/step' % (k, v / nsteps))
