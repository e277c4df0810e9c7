"""GPU idle time between consecutive kernels in a rocprofv3 kernel trace, attributed to the kernel that follows the gap."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
rows = rows[n // 2: n - n // 10]          # steady-state part of the run
gap_by = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0; idle = 0.0
prev_end = int(rows[0]["End_Timestamp"])
for r in rows[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = max(0, s - prev_end)
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:50]
    gap_by[name][0] += 1; gap_by[name][1] += g
    idle += g; busy += e - max(s, prev_end) if e > prev_end else 0
    prev_end = max(prev_end, e)
print("busy %.1f ms idle %.1f ms (%.1f%%) over %d kernels; avg gap %.2f us" % (busy / 1e6, idle / 1e6, 100 * idle / (busy + idle), len(rows), idle / len(rows) / 1e3))
for k, (c, g) in sorted(gap_by.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-50s n=%6d gap total %7.2f ms avg %6.2f us" % (k, c, g / 1e6, g / c / 1e3))
big = sorted(((max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])), a["Kernel_Name"][:40], b["Kernel_Name"][:40]) for a, b in zip(rows, rows[1:])), reverse=True)[:8]
for g, a, b in big:
    print("  gap %.1f us between %s -> %s" % (g / 1e3, a, b))
