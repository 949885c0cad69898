#!/bin/bash
# Builds tools/var_<name>.so: the library with gns_backward.hip (or the file named by SRC=) compiled with extra -D flags,
# for A/B timing on one box:  GNS_LIB=tools/var_<name>.so python tools/gpu_time.py 118 16384 4 1:0:2
# usage: bash tools/build_variant.sh <name> [-DFLAG ...]      (the other objects must have been built by make)
set -e
cd "$(dirname "$0")/../opf-graph-neural-solver_amd/csrc"
SRC=${SRC:-gns_backward}
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -ffp-contract=off -Wno-unused-function"
name=$1; shift
OBJS=""
for o in gns_forward gns_backward gns_backward_split gns_gridwg gns_gridwg_bwd gns_api; do
  if [ "$o" = "$SRC" ]; then OBJS="$OBJS /tmp/var_$name.o"; else OBJS="$OBJS $o.o"; fi
done
hipcc $F "$@" -Rpass-analysis=kernel-resource-usage -c $SRC.hip -o /tmp/var_$name.o 2> /tmp/var_$name.remarks || { cat /tmp/var_$name.remarks | grep -v remark | head -40; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/var_$name.so $OBJS gns_topology.o
grep -A12 "${KERNEL:-gns_backward_kernelILi20ELi10ELb1ELb1ELi2}" /tmp/var_$name.remarks | grep -E "VGPRs:|Spill|Scratch|LDS" | sed 's/ \[-Rpass.*//; s/.*remark: *//' | tr '\n' ';'; echo " <- $name"
