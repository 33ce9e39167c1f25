"""GPU busy time against wall time from a rocprofv3 kernel trace (csv): union of the kernels' [start, end] intervals over the last `frac` of the
trace -- how much of a latency-bound loop (BASIS at 30 tiles) is the device idle between launches?   python scripts/trace_gaps.py trace.csv [frac=0.2]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    r = csv.DictReader(f)
    for row in r:
        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
rows = rows[int(len(rows) * (1 - frac)):]
t0, t1 = rows[0][0], max(e for _, e, _ in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
over = 0
for s, e, _ in rows[1:]:
    if s <= cur_e:
        over += min(e, cur_e) - s if e > s else 0
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
print("kernels %d, wall %.2f ms, device busy (union) %.2f ms = %.1f %%, sum of kernel times %.2f ms (%.2f x wall: concurrency), idle %.2f ms"
      % (len(rows), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), tot / 1e6, tot / (t1 - t0), (t1 - t0 - busy) / 1e6))
gaps = []
cur_e = rows[0][1]
for s, e, n in rows[1:]:
    if s > cur_e: gaps.append((s - cur_e, n))
    cur_e = max(cur_e, e)
gaps.sort(reverse=True)
print("largest gaps (us, next kernel):", [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:8]])
import statistics
if gaps: print("gaps: %d, median %.1f us, mean %.1f us" % (len(gaps), statistics.median(g for g, _ in gaps) / 1e3, sum(g for g, _ in gaps) / len(gaps) / 1e3))
