"""Summarise the PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE rocprofv3 --pmc runs as
MI355X_MICROARCH.md prescribes: the two counters do not fit one pass) for the FeaSt kernels
(default: the fused kernel `feast_fused_kernel`; third argument: another kernel-name stem, e.g.
`feast_aggregate_kernel` for the unfused path under GEOBI_FUSED=0).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
  python tools/pmc_summary.py gpurun_out/pmc profiles/r02_pmc_feast_fused.json

Units and corrections (guide, "HBM" section): both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of wide (16 B/lane) coalesced reads, which is what this kernel issues for
feature and logit rows, so the read side is doubled.  WRITE_SIZE is exact for 16 B/lane stores.
"""
import collections
import csv
import glob
import json
import sys


def load(path, stem):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if stem in r['Kernel_Name']:
            inst = r['Kernel_Name'].split(stem)[1].split('(')[0].replace(' ', '')
            d[inst].append(float(r['Counter_Value']))
    return d


def main(src, dst, stem='feast_fused_kernel'):
    f = load(glob.glob(src + '/fetch/*/*counter_collection.csv')[0], stem)
    w = load(glob.glob(src + '/write/*/*counter_collection.csv')[0], stem)
    out = {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 3 --warmup 1',
           'correction': 'bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE); FETCH_SIZE doubled per the gfx950 note',
           'kernels': {}}
    for inst in sorted(f):
        fa, wa = sum(f[inst]) / len(f[inst]), sum(w[inst]) / len(w[inst])
        out['kernels'][stem + inst] = {
            'launches': len(f[inst]), 'fetch_size_kib_avg': fa, 'write_size_kib_avg': wa,
            'hbm_bytes_per_launch': round(1024 * (2 * fa + wa)),
            'fetch_size_kib_max': max(f[inst]), 'write_size_kib_max': max(w[inst])}
    json.dump(out, open(dst, 'w'), indent=1)
    print(json.dumps(out['kernels'], indent=1))


if __name__ == '__main__':
    main(*sys.argv[1:4])
