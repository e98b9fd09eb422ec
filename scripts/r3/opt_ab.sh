#!/bin/bash
# same-box A/B of library options / library builds; run on the GPU box.
# usage: opt_ab.sh "<arm> <arm> ..." [extra bench args]     arm = [lib:]name=value[,name=value...]  ("-" = defaults)
set -e
mkdir -p gpurun_out
LOG=gpurun_out/opt_ab.log
: > $LOG
for arm in ${1:-"- -"}; do
  lib=""; opts=$arm
  case $arm in *:*) lib=${arm%%:*}; opts=${arm#*:};; esac
  echo "== $arm" >> $LOG
  if [ -n "$lib" ]; then export RCFLOW_LIB=$PWD/ripcurrents_amd/librcflow_$lib.so; else unset RCFLOW_LIB; fi
  o=""
  if [ "$opts" != "-" ]; then for kv in ${opts//,/ }; do o="$o --opt $kv"; done; fi
  python3 bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-roof $o $2 >> $LOG 2>&1
done
python3 - <<'PY'
import json
for l in open("gpurun_out/opt_ab.log"):
    if l.startswith("=="): print(l.strip()); continue
    if l.startswith("{"):
        d = json.loads(l)
        k = {x["kernel"]: x["avg_us"] for x in d["kernels"]}
        print("  fps %.0f (%.0f-%.0f)  ms/step %.4f  flow@0 %.1f us flow@1 %.1f flow@2 %.1f poly@0 %.1f poly@1 %.1f hist %.1f" % (d["value"], d["value_min"], d["value_max"], d["ms_per_step"], k.get("flow_iter_x2@0", 0), k.get("flow_iter_x2@1", 0), k.get("flow_iter_x2@2", 0), k.get("polyexp@0", 0), k.get("polyexp@1", 0), k.get("polar_hist@0", 0)))
    elif "Error" in l or "error" in l: print("  ", l.strip()[:200])
PY
