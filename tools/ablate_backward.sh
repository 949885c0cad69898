#!/bin/bash
# Diagnostic builds of the lane-per-grid backward with one cost component removed (results are WRONG by design, only the
# kernel time means something): -DGNS_ABLATE_PASS (no matrix-pipe contraction), -DGNS_ABLATE_REC (no record stores),
# -DGNS_ABLATE_HBM (sweep rows served from 4 cache-resident buses), -DGNS_ABLATE_SLOAD (weight streams re-use their first chunk).
# usage (build container): bash tools/ablate_backward.sh ; then on the GPU box: GNS_LIB=tools/abl_<name>.so python tools/gpu_time.py 118 16384 4 1:0:2
set -e
cd "$(dirname "$0")/../opf-graph-neural-solver_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -ffp-contract=off -Wno-unused-function"
build() { name=$1; shift; hipcc $F "$@" -c gns_backward.hip -o /tmp/abl_$name.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abl_$name.so /tmp/abl_$name.o gns_forward.o gns_gridwg.o gns_gridwg_bwd.o gns_api.o gns_topology.o; }
build nodw -DGNS_ABLATE_PASS -DGNS_ABLATE_REC
build nohbm -DGNS_ABLATE_HBM
build nohbm_nodw -DGNS_ABLATE_HBM -DGNS_ABLATE_PASS -DGNS_ABLATE_REC
build nosload -DGNS_ABLATE_SLOAD
build nosload_nohbm_nodw -DGNS_ABLATE_SLOAD -DGNS_ABLATE_HBM -DGNS_ABLATE_PASS -DGNS_ABLATE_REC
