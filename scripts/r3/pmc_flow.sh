#!/bin/bash
# SQ counter passes over the flow kernels of one bench configuration; usage: pmc_flow.sh <tag> [bench args]
tag=$1; shift
(rocprofv3 -L > gpurun_out/rocprof_counters_list.txt 2>&1 || true)
PMC_PASSES=scripts/r3/pmc_sq_passes.txt scripts/pmc_multi.sh $tag --steps 2 --warmup 1 --no-roof "$@"
for f in gpurun_out/pmc_${tag}_*.txt; do echo "--- $f"; grep -A24 "k_flow_iter2_rr" $f | head -60; done
