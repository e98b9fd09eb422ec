#!/bin/bash
# Build a variant of librcflow.so with extra -D flags on pyr_polyexp_kernels.hip:
#   scripts/r3/variant_poly.sh NAME "-DRC_POLY_EPI32=1"   ->  ripcurrents_amd/librcflow_NAME.so   (use with RCFLOW_LIB)
set -e
cd "$(dirname "$0")/../../ripcurrents_amd/csrc"
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-result \
    -fno-slp-vectorize $2 -c pyr_polyexp_kernels.hip -o /tmp/pyr_polyexp_kernels_$1.o
OBJS="rcflow_api.o flow_iter_kernels.o exact_kernels.o analysis_kernels.o lk_kernels.o comm_rccl.o flow_iter_kernels_exact.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librcflow_$1.so $OBJS /tmp/pyr_polyexp_kernels_$1.o -ldl
echo built ripcurrents_amd/librcflow_$1.so
