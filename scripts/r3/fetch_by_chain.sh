#!/bin/bash
# FETCH_SIZE of the fused flow kernel at scale 0 by tile-chain length (one rocprofv3 --pmc pass each)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 1 2 4 8; do
  timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_flow" --output-format csv -d gpurun_out/pmc_fc_$c -- python3 bench.py --steps 2 --warmup 1 --warmup-seconds 0 --repeats 1 --pairs 32 --no-roof --no-cpu-baseline --no-kernel-events --opt chain=$c > gpurun_out/pmc_fc_$c.log 2>&1
  python3 scripts/pmc_summary.py gpurun_out/pmc_fc_$c > gpurun_out/pmc_fc_$c.txt 2>&1
  rm -rf gpurun_out/pmc_fc_$c
  echo "chain $c"; grep -A1 "rrc<2, 0, 4, 3, 4, 32, 8> grid" gpurun_out/pmc_fc_$c.txt | head -4
done
