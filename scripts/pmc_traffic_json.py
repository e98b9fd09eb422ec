#!/usr/bin/env python3
"""Builds profiles/rNN_pmc_traffic.json (what bench.py reports as roofline.traffic) from the FETCH_SIZE /
WRITE_SIZE passes of scripts/pmc_multi.sh (PMC_PASSES=scripts/pmc_passes_traffic.txt).

usage: pmc_traffic_json.py <fetch.txt> <write.txt> <fields> <pairs_per_launch> <commit> <out.json>
  fetch.txt / write.txt: pmc_summary.py outputs of the two passes; fields = flow fields the profiled command
  produced (steps + warmup) x pairs.  bytes = 2 x FETCH_SIZE KB x 1024 + WRITE_SIZE KB x 1024
  (MI355X_MICROARCH.md, HBM: gfx950 tallies 128-B read requests at 64 B; WRITE_SIZE is exact)."""
import collections
import json
import re
import sys


def parse(path, counter):
    out, key = {}, None
    for line in open(path):
        m = re.match(r"(\S.*) grid (\d+)", line)
        if m:
            key = (m.group(1), int(m.group(2)))
            continue
        m = re.match(r"\s+%s\s+n=(\d+)\s+mean=(\S+)" % counter, line)
        if m and key:
            out[key] = (int(m.group(1)), float(m.group(2)))
    return out


FAMILY = [("k_flow_iter2_rr", "flow_iter_x2"), ("k_polyexp", "polyexp"), ("k_polar_hist", "polar_hist"),
          ("k_pyr", "pyr_level"), ("k_flow_iter", "flow_iter")]


def main():
    fetch, write = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
    fields, pairs, commit, dst = int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    total = 0.0
    by_family = collections.defaultdict(list)
    for key in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(key, (0, 0.0))
        nw, w = write.get(key, (0, 0.0))
        n = max(nf, nw)
        total += n * (2 * f + w) * 1024
        for prefix, fam in FAMILY:
            if key[0].startswith(prefix):
                by_family[fam].append((key[1], key[0], f, w, n))
                break
    kernels = {}
    for fam, rows in by_family.items():
        # largest grid = scale 0; keep the full-batch launches only (the one-frame priming launch is smaller)
        grids = sorted({g for g, *_ in rows}, reverse=True)
        for level, g in enumerate(grids[:3]):
            r = max((x for x in rows if x[0] == g), key=lambda x: x[4])
            kernels["%s@%d" % (fam, level)] = {"kernel": r[1], "grid_threads": g, "launches": r[4], "FETCH_SIZE_KB": r[2],
                                               "WRITE_SIZE_KB": r[3], "traffic_bytes_per_launch": (2 * r[2] + r[3]) * 1024}
    json.dump({"_comment": "HBM traffic from rocprofv3 --pmc, one pass per counter (FETCH_SIZE, WRITE_SIZE; "
                           "scripts/pmc_multi.sh) over `python3 bench.py --steps 2 --warmup 1 --warmup-seconds 0 --pairs %d "
                           "--no-cpu-baseline --no-kernel-events --no-roof`; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 "
                           "(MI355X_MICROARCH.md: gfx950 tallies 128-B read requests at 64 B, WRITE_SIZE is exact; "
                           "Infinity-Cache hits are included)" % pairs,
               "collected": "separate rocprofv3 --pmc passes, %d flow fields, %d pairs per launch" % (fields, pairs),
               "commit": commit, "pairs_per_launch": pairs, "fields": fields,
               "pipeline_bytes_per_field": total / fields, "kernels": kernels}, open(dst, "w"), indent=1)
    print("pipeline bytes per field: %.1f MB" % (total / fields / 1e6))
    for k, v in sorted(kernels.items()):
        print(k, v["kernel"][:50], "%.3f GB per launch" % (v["traffic_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
