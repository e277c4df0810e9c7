"""Timeline of the replayed P step from a rocprofv3 kernel trace (`*_kernel_trace.csv`): per step the span, the union of busy
intervals (any kernel running), the idle time between kernels, and which kernels the idle time follows / precedes.
Steps are delimited by `seed_word_kernel` (one launch at the head of every step).

    python3 tools/step_gaps.py /tmp/ps/s_kernel_trace.csv [first_step_to_use]
"""
import csv, sys, collections

def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:44]

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
rows.sort()
heads = [i for i, r in enumerate(rows) if r[2].startswith('seed_word_kernel')]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(heads) - 12)
heads = heads[skip:]
spans, busys, sums, conc = [], [], [], []
after = collections.defaultdict(lambda: [0, 0.0]); before = collections.defaultdict(lambda: [0, 0.0])
hist = collections.Counter()
for a, b in zip(heads[:-1], heads[1:]):
    ks = rows[a:b]
    if len(ks) < 500: continue
    t0 = ks[0][0]; t1 = max(k[1] for k in ks)
    if (t1 - t0) > 60e6: continue
    spans.append(t1 - t0); sums.append(sum(k[1] - k[0] for k in ks))
    busy = 0; cur_end = ks[0][0]; last = None; c2 = 0
    for s, e, n in ks:
        if s > cur_end:
            g = s - cur_end
            x = after[last]; x[0] += 1; x[1] += g
            y = before[n]; y[0] += 1; y[1] += g
            hist[min(int(g / 1000), 20)] += 1
            busy += e - s; cur_end = e; last = n
        else:
            if e > cur_end:
                busy += e - cur_end; cur_end = e; last = n
    busys.append(busy)
n = len(spans)
if not n:
    print("no steps found (%d heads)" % len(heads)); sys.exit(0)
print("steps analysed %d: span %.2f ms, busy (union) %.2f ms, idle %.2f ms, kernel sum %.2f ms" % (
    n, sum(spans) / n / 1e6, sum(busys) / n / 1e6, (sum(spans) - sum(busys)) / n / 1e6, sum(sums) / n / 1e6))
print("gap histogram (us : gaps per step):", ", ".join("%d:%.0f" % (k, v / n) for k, v in sorted(hist.items())))
print("idle time by the kernel that FOLLOWS the gap (per step):")
for k, (c, t) in sorted(before.items(), key=lambda kv: -kv[1][1])[:18]:
    print("  %-46s %6.1f gaps  %7.3f ms  avg %5.2f us" % (k, c / n, t / n / 1e6, t / c / 1e3))
print("idle time by the kernel that PRECEDES the gap (per step):")
for k, (c, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:18]:
    print("  %-46s %6.1f gaps  %7.3f ms  avg %5.2f us" % (k, c / n, t / n / 1e6, t / c / 1e3))
