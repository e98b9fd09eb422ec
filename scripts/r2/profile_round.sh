#!/bin/bash
# Round-2 profile set, run on the GPU box from the repo root:  bash scripts/r2/profile_round.sh <commit>
#   1. bench.py with the driver's arguments                      -> gpurun_out/r02_bench.json
#   2. rocprofv3 --kernel-trace of the same command              -> gpurun_out/r02_bench_kernel_summary.md
#   3. FETCH_SIZE / WRITE_SIZE passes (one counter per pass)     -> gpurun_out/r02_pmc_traffic.json
commit=${1:-unknown}
python3 -c 'import torch' > /dev/null 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roof > gpurun_out/prof_r02.log 2>&1 || exit 1
python3 scripts/rocprof_summary.py gpurun_out/prof_r02 gpurun_out/r02_bench_kernel_summary.md > /dev/null
f=$(find gpurun_out/prof_r02 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/r02_bench_kernel_stats.csv
rm -rf gpurun_out/prof_r02
PMC_PASSES=scripts/pmc_passes_traffic.txt PMC_TIMEOUT=300 bash scripts/pmc_multi.sh r02t --steps 2 --warmup 1 --warmup-seconds 0 --pairs 32 --no-roof
python3 scripts/pmc_traffic_json.py gpurun_out/pmc_r02t_1.txt gpurun_out/pmc_r02t_2.txt 96 32 $commit gpurun_out/r02_pmc_traffic.json
