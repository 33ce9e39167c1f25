"""Print a rocprofv3 kernel_stats.csv compactly: kernel (template arguments kept), calls, average us, share."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    m = re.match(r"(?:void )?(\w+)(<[^(]*>)?", r["Name"])
    print("%-56s calls %6s avg %8.1f us  %5.1f%%" % ((m.group(1) + (m.group(2) or ""))[:56], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total %.1f ms" % (tot / 1e6))
