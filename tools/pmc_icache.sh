#!/bin/bash
# Instruction-cache counters of the training kernels (one rocprofv3 --pmc pass each, counters only).
# usage (on the GPU box): bash tools/pmc_icache.sh <outdir>
set -o pipefail
OUT=${1:-gpurun_out/pmc_icache}
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $ROOT/$OUT/p$i -o pmc -- python3 $ROOT/tools/gpu_train_once.py > $ROOT/$OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'gns_' in k and ('forward' in k or 'backward' in k):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$OUT/summary.txt', 'w') as o:
    for k in sorted(agg):
        o.write(k + '\n')
        for c in sorted(agg[k]):
            v = agg[k][c]
            o.write('   %-28s mean %16.1f  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open('$OUT/summary.txt').read())
PY
