#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command (the run the round's profiles/ summary comes from).
set -o pipefail
OUT=${1:-gpurun_out/prof_bench}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT -o bench -- python3 $ROOT/bench.py --steps 20 --warmup 3 --sustained-steps 0 --no-cpu-baseline > $ROOT/$OUT/bench.json 2> $ROOT/$OUT/bench.err
cd $ROOT
ls $OUT
