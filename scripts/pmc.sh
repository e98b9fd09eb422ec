#!/bin/bash
# usage: scripts/pmc.sh <tag> "<counters>" -- runs bench briefly under rocprofv3 --pmc
tag=$1; shift
ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 2 --warmup 1 --pairs 8 --no-cpu-baseline --no-kernel-events > gpurun_out/pmc_$tag.log 2>&1
echo "exit=$?" >> gpurun_out/pmc_$tag.log
