#!/bin/bash
# rocprofv3 kernel-trace summary of a few training steps (tools/gpu_train_once.py) under the options given in the environment
# (GNS_BWD_VARIANT, GNS_TRAIN_MAPPING, ...).  usage (GPU box): GNS_BWD_VARIANT=4 bash tools/prof_train_once.sh <outdir> [case bt K]
set -o pipefail
OUT=${1:-gpurun_out/prof_train}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT -o train -- python3 $ROOT/tools/gpu_train_once.py "$@" > $ROOT/$OUT/train.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob
for f in glob.glob('$OUT/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} {r['Percentage']}%")
PY
