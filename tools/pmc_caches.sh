#!/bin/bash
# Scalar-data-cache and instruction-cache hit rates of the fused kernels (separate counter-only passes).
set -o pipefail
OUT=${1:-gpurun_out/pmc_caches}
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_TC_DATA_READ_REQ SQC_TC_INST_REQ SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $ROOT/$OUT/p$i -o pmc -- python3 $ROOT/tools/gpu_time.py > $ROOT/$OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'gns_forward_kernel' in k or 'gns_backward_kernel' in k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$OUT/summary.txt', 'w') as o:
    for k in sorted(agg):
        o.write(k + '\n')
        for c in sorted(agg[k]):
            v = agg[k][c]
            o.write(f'   {c:30s} mean {sum(v)/len(v):16.1f}  (n={len(v)})\n')
print(open('$OUT/summary.txt').read())
PY
