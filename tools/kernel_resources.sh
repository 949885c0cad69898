#!/bin/bash
# Register / LDS / spill figures of every kernel instantiation as hipcc reports them (-Rpass-analysis=kernel-resource-usage).
# usage: bash tools/kernel_resources.sh > profiles/rNN/kernel_resources.txt
cd "$(dirname "$0")/../opf-graph-neural-solver_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -ffp-contract=off -Wno-unused-function"
echo "# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, shipped sources (per instantiation)"
for f in gns_forward gns_backward gns_backward_split gns_gridwg gns_gridwg_bwd gns_api; do
  [ -n "$KR_REUSE" ] || hipcc $F -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/kr_$f.o 2> /tmp/kr_$f.txt &
done
wait
for f in gns_forward gns_backward gns_backward_split gns_gridwg gns_gridwg_bwd gns_api; do
  python3 - /tmp/kr_$f.txt <<'PY'
import re, subprocess, sys
cur = None; rows = {}
for l in open(sys.argv[1]):
    m = re.search(r'remark: Function Name: (\S+)', l)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)', l)
    if m and cur: rows[cur][m.group(1)] = m.group(2)
names = subprocess.run(['c++filt'] + list(rows), capture_output=True, text=True).stdout.split('\n')
for n, k in sorted(zip(names, rows)):
    r = rows[k]
    print(n.split('(')[0]); print('    ' + '; '.join(f'{a}: {b}' for a, b in r.items()))
PY
done
