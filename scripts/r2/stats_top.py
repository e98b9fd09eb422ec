"""print the top rows of a rocprofv3 kernel_stats.csv (names shortened)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for r in rows[:n]:
    name = r["Name"].replace("void ", "").replace("rc_flow_fast::", "")[:60]
    print("%-60s calls %5s avg %9.2f us  %5.1f %%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
