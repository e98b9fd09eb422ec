#!/bin/bash
# usage: scripts/pmc_multi.sh <tag> <bench args...>   (one rocprofv3 --pmc pass per line of scripts/pmc_passes.txt)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  echo "pass $i: $line"
  timeout -k 5 ${PMC_TIMEOUT:-120} rocprofv3 --pmc $line --kernel-trace --kernel-include-regex "k_(flow|poly|pyr|polar|hist|thresh|exact|classify|advect)" --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 bench.py "$@" --no-cpu-baseline --no-kernel-events > gpurun_out/pmc_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -2 gpurun_out/pmc_${tag}_$i.log; }
  python3 scripts/pmc_summary.py gpurun_out/pmc_${tag}_$i > gpurun_out/pmc_${tag}_$i.txt 2>&1
  [ -n "$PMC_KEEP_RAW" ] || rm -rf gpurun_out/pmc_${tag}_$i
done < ${PMC_PASSES:-scripts/pmc_passes.txt}
echo done $i passes
