#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per (kernel, grid)."""
import csv, glob, sys, collections
d = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "::k_" in name:
            name = name.split("::")[-1]
        if not name.startswith("k_"):
            continue
        key = (name, r.get("Grid_Size", "") or r.get("Grid_Size_X", ""))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(rows):
    print(key[0], "grid", key[1])
    for c, v in sorted(rows[key].items()):
        print("   %-28s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
