"""Per-queue (HIP stream) busy time of the replayed P step from a rocprofv3 kernel trace, and the kernels of the busiest queue
(the chain that bounds the step).   python3 tools/step_streams.py /tmp/ps/s_kernel_trace.csv"""
import csv, sys, collections
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:40]
rows = list(csv.DictReader(open(sys.argv[1])))
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else ('Stream_Id' if 'Stream_Id' in rows[0] else None)
print("columns:", list(rows[0].keys()))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get(qkey, '0'), r.get('Stream_Id', '')) for r in rows)
heads = [i for i, e in enumerate(ev) if e[2].startswith('seed_word_kernel')]
heads = heads[max(0, len(heads) - 9):]
perq = collections.defaultdict(float); perq_n = collections.Counter(); nst = 0; span = 0
kq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for a, b in zip(heads[:-1], heads[1:]):
    ks = ev[a:b]
    if len(ks) < 500: continue
    nst += 1; span += max(k[1] for k in ks) - ks[0][0]
    for s, e, n, q, st in ks:
        key = (q, st)
        perq[key] += e - s; perq_n[key] += 1
        x = kq[key][n]; x[0] += 1; x[1] += e - s
print("steps %d, span %.2f ms" % (nst, span / nst / 1e6))
for key, t in sorted(perq.items(), key=lambda kv: -kv[1]):
    print("queue/stream %s: %.2f ms busy, %d kernels per step" % (key, t / nst / 1e6, perq_n[key] / nst))
    if t / nst / 1e6 > 0.3:
        for n, (c, tt) in sorted(kq[key].items(), key=lambda kv: -kv[1][1])[:14]:
            print("      %-42s %5.1f  %6.3f ms" % (n, c / nst, tt / nst / 1e6))
