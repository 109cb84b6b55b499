#!/bin/bash
# usage: tools/kres.sh [name filter regex]: compiles msplat_device.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and
# prints VGPRs / SGPRs / spills / scratch / occupancy / LDS of the kernels whose demangled name matches (no GPU needed)
cd "$(dirname "$0")/.."
F=${1:-composite_kernel|project_kernel}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c -Rpass-analysis=kernel-resource-usage \
    splatapult_amd/csrc/msplat_device.hip -o /tmp/kres_dev.o 2> /tmp/kres.txt || { grep -E "error" -A5 /tmp/kres.txt | head -60; exit 1; }
python3 - "$F" <<'PY'
import re, subprocess, sys
txt = open('/tmp/kres.txt').read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
flt = re.compile(sys.argv[1])
names = [b.split()[0] for b in blocks]
dem = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.splitlines()
for b, d in zip(blocks, dem):
    d = re.sub(r'\(.*', '', d).replace('void msplat::', '')
    if not flt.search(d): continue
    g = lambda k: (re.search(k + r': (\d+)', b) or [None, '?'])[1]
    print('%-52s VGPR %3s SGPR %3s sgpr-spill %2s vgpr-spill %2s scratch %3s occ %s LDS %s' % (
        d[:52], g('VGPRs'), g('SGPRs'), g('SGPRs Spill'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'),
        g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
PY
