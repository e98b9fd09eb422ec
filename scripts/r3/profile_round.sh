#!/bin/bash
# Round-3 profile set, run on the GPU box from the repo root:  bash scripts/r3/profile_round.sh <commit>
#   1. bench.py with the driver's arguments                      -> gpurun_out/r03_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command      -> gpurun_out/r03_bench_kernel_summary.md (+ the stats csv)
#   3. FETCH_SIZE / WRITE_SIZE passes (one counter per pass)     -> gpurun_out/r03_pmc_traffic.json; SQ passes -> gpurun_out/r03_sq_all.md, r03_sq_counters.json
#   4. the other BASELINE.md rows (c1, c3, c5, frame, host, host-stateless, gaussian, exact) -> gpurun_out/r03_bench_<row>.json
#   5. rocprofv3 kernel summary of C3 (4K, five scales)          -> gpurun_out/r03_c3_kernel_summary.md
commit=${1:-unknown}
python3 -c 'import torch' > /dev/null 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || exit 1
echo "bench done"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roof > gpurun_out/prof_r03.log 2>&1 || exit 1
python3 scripts/rocprof_summary.py gpurun_out/prof_r03 gpurun_out/r03_bench_kernel_summary.md > /dev/null
f=$(find gpurun_out/prof_r03 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/r03_bench_kernel_stats.csv
rm -rf gpurun_out/prof_r03
echo "trace done"
PMC_PASSES=scripts/pmc_passes_traffic.txt PMC_TIMEOUT=300 bash scripts/pmc_multi.sh r03t --steps 2 --warmup 1 --warmup-seconds 0 --repeats 1 --pairs 32 --no-roof
python3 scripts/pmc_traffic_json.py gpurun_out/pmc_r03t_1.txt gpurun_out/pmc_r03t_2.txt 96 32 $commit gpurun_out/r03_pmc_traffic.json
PMC_PASSES=scripts/r3/pmc_sq2.txt PMC_TIMEOUT=300 bash scripts/pmc_multi.sh r03s --steps 2 --warmup 1 --warmup-seconds 0 --repeats 1 --pairs 32 --no-roof
python3 scripts/r3/sq_summary.py gpurun_out/pmc_r03s_1.txt gpurun_out/pmc_r03s_2.txt > gpurun_out/r03_sq_all.md
python3 scripts/r3/sq_json.py gpurun_out/pmc_r03s_1.txt gpurun_out/pmc_r03s_2.txt 32 $commit gpurun_out/r03_sq_counters.json
echo "pmc done"
for row in "c1:--config c1" "c3:--config c3" "c5:--config c5" "frame:--mode frame" "host:--mode host" "host_stateless:--mode host-stateless --steps 5 --repeats 3" "gaussian:--gaussian" "exact:--exact --repeats 5"; do
  name=${row%%:*}; args=${row#*:}
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roof $args > gpurun_out/r03_bench_$name.json 2> gpurun_out/r03_bench_$name.err || echo "row $name failed"
  echo "row $name done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03c3 -- python3 bench.py --config c3 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-roof > gpurun_out/prof_r03c3.log 2>&1 || exit 1
python3 scripts/rocprof_summary.py gpurun_out/prof_r03c3 gpurun_out/r03_c3_kernel_summary.md > /dev/null
rm -rf gpurun_out/prof_r03c3
echo "all done"
