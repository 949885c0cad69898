#!/bin/bash
# Hardware counters of the training-mode forward and the backward for one mapping (separate rocprofv3 passes, counters only).
# usage (on the GPU box): GNS_TRAIN_MAPPING=lds GNS_GW_PACK=1 bash tools/pmc_train.sh <outdir>
set -o pipefail
OUT=${1:-gpurun_out/pmc_train}
mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
         "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $ROOT/$OUT/p$i -o pmc -- python3 $ROOT/tools/gpu_train_once.py > $ROOT/$OUT/p$i.log 2>&1 || echo "pass $i failed"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'gns_' in k and ('forward' in k or 'backward' in k or 'bwds' in k):
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$OUT/summary.txt', 'w') as o:
    for k in sorted(agg):
        o.write(k + '\n')
        for c in sorted(agg[k]):
            v = agg[k][c]
            o.write(f'   {c:28s} mean {sum(v)/len(v):16.1f}  (n={len(v)})\n')
print(open('$OUT/summary.txt').read())
PY
