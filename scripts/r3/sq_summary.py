#!/usr/bin/env python3
"""Per-kernel issue statistics from the SQ passes of scripts/r3/profile_round.sh (pmc_summary.py outputs):
VALU / SALU / LDS instructions per wave, how a wave's life divides into issuing / issue-stalled / parked, and the share
of the launch's cycles in which a SIMD's VALU was issuing (SQ_ACTIVE_INST_VALU counts quad-cycles, one per instruction
and wave; 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs)."""
import collections
import re
import sys

vals = collections.defaultdict(dict)
for path in sys.argv[1:]:
    cur = None
    for line in open(path):
        m = re.match(r"(\S.*) grid (\d+)", line)
        if m:
            cur = (m.group(1), int(m.group(2)))
            continue
        m = re.match(r"\s+(\S+)\s+n=(\d+)\s+mean=(\S+)", line)
        if m and cur:
            vals[cur][m.group(1)] = float(m.group(3))
print("| kernel | grid (threads) | waves | VALU / wave | SALU / wave | LDS / wave | issuing | issue-stalled | parked | VALU issue share of launch cycles | clock GHz (if us given) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for (name, grid), v in sorted(vals.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if "SQ_WAVES" not in v or v["SQ_WAVES"] < 1000:
        continue
    wv = v["SQ_WAVES"]
    wc = v.get("SQ_WAVE_CYCLES", 0)
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    share = 4 * v.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * cyc) if cyc else 0
    f = lambda k: (v.get(k, 0) / wc) if wc else 0
    print("| %s | %d | %d | %.0f | %.0f | %.0f | %.0f %% | %.0f %% | %.0f %% | %.0f %% | |" % (
        name[:60], grid, wv, v.get("SQ_INSTS_VALU", 0) / wv, v.get("SQ_INSTS_SALU", 0) / wv, v.get("SQ_INSTS_LDS", 0) / wv,
        100 * f("SQ_ACTIVE_INST_ANY"), 100 * f("SQ_WAIT_INST_ANY"), 100 * f("SQ_WAIT_ANY"), 100 * share))
