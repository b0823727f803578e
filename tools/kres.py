"""Per-kernel register / scratch / occupancy table from a `hipcc -Rpass-analysis=kernel-resource-usage` log:
   python tools/kres.py build.log [name filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)
names = [b.split('\n')[0] for b in blocks[1:]]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
for b, nm in zip(blocks[1:], dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    nm = nm.replace('void ', '').replace('geobi::(anonymous namespace)::', '')
    nm = re.sub(r'\((float|int|geobi|HIP|long|unsigned|void|char).*', '', nm)
    if flt and flt not in nm:
        continue
    print('%-52s vgpr %3d agpr %3d scratch %4d occ %d sgpr %3d lds %6d' % (
        nm[:52], g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g('SGPRs'),
        g(r'LDS Size \[bytes/block\]')))
