#!/bin/bash
# same-box A/B of the tile-chain length of the fused winsize-3 flow kernel (option "chain"); run on the GPU box.
# usage: chain_ab.sh "1 8 4 1 8"  [extra bench args]
set -e
mkdir -p gpurun_out
LOG=gpurun_out/chain_ab.log
: > $LOG
for c in ${1:-1 8 1 8}; do
  lib=""; ch=$c
  case $c in *:*) lib=${c%%:*}; ch=${c##*:};; esac
  echo "== chain $ch lib ${lib:-default}" >> $LOG
  if [ -n "$lib" ]; then export RCFLOW_LIB=$PWD/ripcurrents_amd/librcflow_$lib.so; else unset RCFLOW_LIB; fi
  python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roof --opt chain=$ch $2 >> $LOG 2>&1
done
python3 - <<'PY'
import json
for l in open("gpurun_out/chain_ab.log"):
    if l.startswith("=="): print(l.strip()); continue
    if l.startswith("{"):
        d = json.loads(l)
        k = {x["kernel"]: x["avg_us"] for x in d["kernels"]}
        print("  fps %.0f  ms/step %.4f  flow@0 %.1f us flow@1 %.1f flow@2 %.1f poly@0 %.1f hist %.1f" % (d["value"], d["ms_per_step"], k.get("flow_iter_x2@0", 0), k.get("flow_iter_x2@1", 0), k.get("flow_iter_x2@2", 0), k.get("polyexp@0", 0), k.get("polar_hist@0", 0)))
PY
