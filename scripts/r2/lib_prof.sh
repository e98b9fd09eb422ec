#!/bin/bash
# per-kernel times (rocprofv3 kernel trace) of library variants on the 32-pair 1080p clip:
#   ABL=<ablate value> scripts/r2/lib_prof.sh ripcurrents_amd/librcflow_x.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  export RCFLOW_LIB=$GRAFT_REPO_ROOT/$lib
  n=$(basename $lib .so)
  rm -rf /tmp/lp_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp_$n -- python3 scripts/r2/abl_ab.py ${ABL:-32} > gpurun_out/lp_$n.log 2>&1
  f=$(find /tmp/lp_$n -name '*kernel_stats.csv' | head -1)
  echo "== $n ablate ${ABL:-32}: $(grep 'us per pair' gpurun_out/lp_$n.log | tail -1)"; python3 scripts/r2/stats_top.py $f 12 | grep "flow_iter"
done
