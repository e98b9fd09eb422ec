#!/usr/bin/env python3
"""profiles/rNN_sq_counters.json from the SQ passes of scripts/r3/profile_round.sh (pmc_summary.py outputs): per kernel
launch the wave count, the vector / scalar / LDS instruction counts, the launch's own cycle count (GRBM_GUI_ACTIVE / 8 XCDs)
and SQ_ACTIVE_INST_VALU (quad-cycles).  bench.py reads it for `roofline.valu_issue` the way it reads the traffic file.
usage: sq_json.py <pass1.txt> <pass2.txt> <pairs_per_launch> <commit> <out.json>"""
import collections
import json
import re
import sys

vals = collections.defaultdict(dict)
for path in sys.argv[1:3]:
    cur = None
    for line in open(path):
        m = re.match(r"(\S.*) grid (\d+)", line)
        if m:
            cur = (m.group(1), int(m.group(2)))
            continue
        m = re.match(r"\s+(\S+)\s+n=(\d+)\s+mean=(\S+)", line)
        if m and cur:
            vals[cur][m.group(1)] = float(m.group(3))
FAMILY = [("k_flow_iter2_rr", "flow_iter_x2"), ("k_polyexp<", "polyexp"), ("k_polar_hist", "polar_hist")]
fam = collections.defaultdict(list)
for (name, grid), v in vals.items():
    if v.get("SQ_WAVES", 0) < 1000:
        continue
    for prefix, f in FAMILY:
        if name.startswith(prefix):
            fam[f].append((grid, name, v))
out = {"_comment": "SQ counters from rocprofv3 --pmc (scripts/r3/pmc_sq2.txt: two passes, means over the launches of `bench.py --steps 2 --warmup 1 "
                   "--pairs 32`); cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the XCDs); a SIMD issues one wave64 vector instruction "
                   "per 2-cycle pass (simple fp32 / integer ops on VGPRs in a pure stream) or per 4 cycles (everything else, and any "
                   "mixed stream): profiles/r03_valu_rates.md",
       "commit": sys.argv[4], "pairs_per_launch": int(sys.argv[3]), "simds": 1024, "kernels": {}}
for f, rows in fam.items():
    for level, (grid, name, v) in enumerate(sorted(rows, reverse=True)[:3]):
        cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
        iv = v.get("SQ_INSTS_VALU", 0)
        out["kernels"]["%s@%d" % (f, level)] = {
            "kernel": name, "grid_threads": grid, "waves": v.get("SQ_WAVES"), "insts_valu": iv, "insts_salu": v.get("SQ_INSTS_SALU"),
            "insts_lds": v.get("SQ_INSTS_LDS"), "launch_cycles": cyc, "active_inst_valu_quadcycles": v.get("SQ_ACTIVE_INST_VALU"),
            "cycles_per_valu_inst_per_simd": round(cyc * 1024 / iv, 3) if iv else None,
            "valu_active_share_of_launch": round(4 * v.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * cyc), 3) if cyc else None}
json.dump(out, open(sys.argv[5], "w"), indent=1)
print(json.dumps({k: (v["cycles_per_valu_inst_per_simd"], v["valu_active_share_of_launch"]) for k, v in out["kernels"].items()}))
