#!/bin/bash
# usage: scripts/r2/pmc_one.sh <tag> "<counters>" "<kernel regex>" <python script> [args...]
tag=$1; ctrs=$2; rex=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c 'import torch' > /dev/null 2>&1
timeout -k 5 300 rocprofv3 --pmc $ctrs --kernel-trace --kernel-include-regex "$rex" --output-format csv -d gpurun_out/pmc_$tag -- python3 "$@" > gpurun_out/pmc_$tag.log 2>&1
python3 scripts/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/pmc_$tag.txt 2>&1
rm -rf gpurun_out/pmc_$tag
cat gpurun_out/pmc_$tag.txt
