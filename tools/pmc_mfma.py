"""Summarise the MFMA counters of the dense kernels (node-level GEMMs, fused heads) from one
rocprofv3 --pmc pass over the bench workload:

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d gpurun_out/pmc2/mfma -- \
      python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
  python tools/pmc_mfma.py gpurun_out/pmc2/mfma profiles/r01_pmc_gemm_mfma.json
(round 3: the same pass also covers the fused FeaSt kernels, where the node transforms live since round 2:
 feast_fused_kernel, feast_rowpass_fused_kernel, feast_rowpass_fused128_kernel -> profiles/r03_pmc_mfma.json)

Per kernel (all launches of the run summed):
  flops         = 512 * SQ_INSTS_VALU_MFMA_MOPS_F32   (one MOP = 512 flop; a 32x32x2 f32 MFMA is 8 MOPs, a 16x16x4 one 4)
  mfma_busy     = SQ_VALU_MFMA_BUSY_CYCLES / (kernel ns * 2.4 GHz * 1024 SIMDs): fraction of the chip's MFMA issue
                  slots in use at the nominal clock (64 cycles per 32x32x2 f32 MFMA).  The gfx94x MfmaUtil formula
                  divides by GRBM_GUI_ACTIVE * CUs * 4 instead; on gfx950 GRBM_GUI_ACTIVE comes back summed over the
                  8 XCDs (~22 counts per ns), which makes that figure ~9x too small, so it is not reported.
  tflops        = flops / kernel time (End - Start timestamps of the same dispatches), against the 157.3 TFLOP/s
                  fp32 MFMA peak of MI355X_MICROARCH.md
These GEMMs are K<=1152, N<=128 with 1e3..3e5 rows: HBM / launch bound, so the utilisation is low by
construction -- the number is reported because BASELINE.json's north_star asks for it.
"""
import collections
import csv
import glob
import json
import re
import sys

PEAK_TFLOPS = 157.3
DENSE = ('gemm_nn_kernel', 'gemm_tn_kernel', 'head_fwd_fused_kernel', 'head_bwd_fused_kernel', 'head_kernel',
         'feast_fused_kernel', 'feast_rowpass_fused_kernel', 'feast_rowpass_fused128_kernel')


def short(name):
    m = re.search(r'(\w+_kernel)(<[^(]*>)?', name)
    return (m.group(1) + (m.group(2) or '')).replace(' ', '') if m else name[:60]


def main(src, dst):
    path = glob.glob(src + '/*/*counter_collection.csv')[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        if not k.startswith(DENSE):
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            acc[k]['ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            acc[k]['launches'] += 1
    out = {'source': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, '
                     'bench.py --steps 3 --warmup 1 (4 steps of 4 x icosphere n=32)',
           'formulas': 'flops = 512*MOPS_F32; mfma_busy = MFMA_BUSY_CYCLES / (ns * 2.4 * 1024 SIMDs); peak 157.3 TFLOP/s fp32 MFMA',
           'note': 'kernel time under counter collection is serialised and slower than the un-profiled run', 'kernels': {}}
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    for k, c in sorted(acc.items()):
        for name in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F32', 'GRBM_GUI_ACTIVE', 'ns', 'launches'):
            fam[k.split('<')[0]][name] += c[name]
    for k, c in list(sorted(acc.items())) + [(f + ' (all instantiations)', c) for f, c in sorted(fam.items())]:
        flops = 512.0 * c['SQ_INSTS_VALU_MFMA_MOPS_F32']
        out['kernels'][k] = {
            'launches': int(c['launches']), 'avg_us': round(c['ns'] / 1e3 / c['launches'], 2),
            'gflop_per_launch': round(flops / 1e9 / c['launches'], 4),
            'tflops': round(flops / c['ns'] / 1e3, 2), 'frac_of_peak': round(flops / c['ns'] / 1e3 / PEAK_TFLOPS, 4),
            'mfma_busy': round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['ns'] * 2.4 * 1024), 4)}
    json.dump(out, open(dst, 'w'), indent=1)
    for k, v in out['kernels'].items():
        if 'all inst' in k:
            print(k, v)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
