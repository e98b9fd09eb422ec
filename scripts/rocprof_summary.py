#!/usr/bin/env python3
"""Per-(kernel, grid) summary of a rocprofv3 --kernel-trace CSV (durations in us).

rocprofv3 --stats aggregates by kernel name only; the Farneback kernels run once per
pyramid scale with the same name, so this groups by grid size as well.
usage: rocprof_summary.py <dir with *_kernel_trace.csv> [out.md]
"""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            short = name.split("(")[0].replace("void ", "")
            short = short.split("::")[-1] if "::k_" in short else short
            if not short.startswith("k_"):
                short = "(torch) " + short[:60]
            key = (short, "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]),
                   r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"])
            rows[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in rows.values())
    lines = ["| kernel | grid (threads) | wg | LDS B | VGPR | calls | avg us | min us | max us | total ms | % |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    for key, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        if key[0].startswith("(torch)") and sum(v) / total < 0.01:
            continue
        lines.append("| %s | %s | %s | %s | %s | %d | %.2f | %.2f | %.2f | %.3f | %.1f |" % (
            key[0], key[1], key[2], key[3], key[4], len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1e3,
            100 * sum(v) / total))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
