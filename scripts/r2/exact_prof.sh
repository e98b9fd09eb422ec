#!/bin/bash
# kernel breakdown of the exact path: scripts/r2/exact_prof.sh RC215 32
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/ep; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ep -- python3 scripts/r2/exact_clip.py $1 1 ${2:-32} > gpurun_out/exact_prof.log 2>&1
f=$(find /tmp/ep -name '*kernel_stats.csv' | head -1)
python3 scripts/r2/stats_top.py $f 40 | grep -v "at::native"
