"""Round 6: how full is the chip over the replayed P step?  From a rocprofv3 kernel trace (tools/prof_gaps.sh leaves one in /tmp/ps): every
instant of a step is classified by the LARGEST launch running then -- full (>= 256 workgroups), three quarters (176-255), half (96-175),
small (< 96), idle -- and the time per class is reported, with the kernels that own the small / half time.
    python3 tools/lab/occupancy_timeline.py /tmp/ps/s_kernel_trace.csv"""
import csv, sys, collections

def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    wg = max(1, int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1))
    grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), grid // wg))
rows.sort()
heads = [i for i, r in enumerate(rows) if r[2].startswith('seed_word_kernel')]
heads = heads[max(0, len(heads) - 9):]
def cls(w):
    return 3 if w >= 256 else (2 if w >= 176 else (1 if w >= 96 else 0))
names = ["small (< 96 wg)", "half (96-175)", "3/4 (176-255)", "full (>= 256)"]
tot = collections.Counter(); own = [collections.Counter() for _ in range(4)]; nsteps = 0; span_sum = 0
for a, b in zip(heads[:-1], heads[1:]):
    ks = rows[a:b]
    if len(ks) < 500: continue
    t0, t1 = ks[0][0], max(k[1] for k in ks)
    if t1 - t0 > 60e6: continue
    nsteps += 1; span_sum += t1 - t0
    ev = []
    for i, (s, e, n, w) in enumerate(ks):
        ev.append((s, 1, i)); ev.append((e, 0, i))
    ev.sort()
    live = set(); last = t0
    for t, kind, i in ev:
        if t > last:
            if live:
                top = max(live, key=lambda j: ks[j][3])
                c = cls(ks[top][3]); tot[c] += t - last; own[c][ks[top][2]] += t - last
            else:
                tot['idle'] += t - last
            last = t
        if kind: live.add(i)
        else: live.discard(i)
print("steps %d, span %.2f ms" % (nsteps, span_sum / nsteps / 1e6))
for c in (3, 2, 1, 0):
    print("  %-18s %6.2f ms / step" % (names[c], tot[c] / nsteps / 1e6))
print("  %-18s %6.2f ms / step" % ("idle", tot['idle'] / nsteps / 1e6))
for c in (0, 1, 2):
    print("time owned while the largest running launch is %s:" % names[c])
    for k, v in own[c].most_common(14):
        print("    %-46s %6.3f ms" % (k, v / nsteps / 1e6))
