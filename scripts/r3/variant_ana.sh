#!/bin/bash
# Build a variant of librcflow.so with extra -D flags on analysis_kernels.hip :
#   scripts/r3/variant_ana.sh NAME "-DRC_HIST_ABL=1"   ->  ripcurrents_amd/librcflow_NAME.so   (git-ignored; use with RCFLOW_LIB)
set -e
cd "$(dirname "$0")/../../ripcurrents_amd/csrc"
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-inline-asm \
    $2 -c analysis_kernels.hip -o /tmp/analysis_kernels_$1.o
OBJS="rcflow_api.o pyr_polyexp_kernels.o exact_kernels.o flow_iter_kernels.o lk_kernels.o comm_rccl.o flow_iter_kernels_exact.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librcflow_$1.so $OBJS /tmp/analysis_kernels_$1.o -ldl
echo built ripcurrents_amd/librcflow_$1.so
