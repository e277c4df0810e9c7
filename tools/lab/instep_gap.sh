#!/bin/bash
# Round 6: the same GEMM launches INSIDE an eager P step and BACK TO BACK -- elapsed cycles (GRBM_GUI_ACTIVE / 8),
# L2 hit rate (TCC_HIT / (HIT + MISS)) and bytes requested from the fabric (FETCH_SIZE x 2 on gfx950) per dispatch, per SHAPE
# (the step's launches are identified by their order: tools/lab/instep_gemm_step.py lists them).  Separate PMC passes, kernel-trace
# only; WRITE_SIZE tells the fused epilogues of the step (pre-activation stores, ...) from the plain ones of the back-to-back run.
# Under PMC every dispatch runs alone, so "in the step" here is the step's CACHE STATE, not its concurrency.
R=$PWD
cd /tmp && export TMPDIR=/tmp
for pass in GRBM_GUI_ACTIVE "TCC_HIT_sum TCC_MISS_sum" FETCH_SIZE WRITE_SIZE; do
  tag=$(echo $pass | cut -c1-7 | tr ' ' '_')
  rm -rf /tmp/ig_s_$tag /tmp/ig_b_$tag
  INSTEP_RECORDS=/tmp/instep_records_$tag.json rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/ig_s_$tag -o s -- python3 $R/tools/lab/instep_gemm_step.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/ig_b_$tag -o b -- python3 $R/tools/lab/gemm_b2b.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, collections, glob, json
def gemm_rows(f):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'gemm_gl' not in n and 'gemm_pp' not in n: continue
        rows.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
    return [rows[k] for k in sorted(rows)]
def med(v):
    v = sorted(v); return v[len(v) // 2] if v else float('nan')
step = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ("GRBM_GU", "TCC_HIT", "FETCH_S", "WRITE_S"):
    recs = json.load(open('/tmp/instep_records_%s.json' % tag))
    rows = gemm_rows(glob.glob('/tmp/ig_s_%s/*_counter_collection.csv' % tag)[0])[-len(recs):]
    assert len(rows) == len(recs), (len(rows), len(recs))
    for rec, c in zip(recs, rows):
        key = tuple(rec[:3]) + (rec[6], rec[7], rec[8])          # M, N, K, precision, a k-major, b k-major
        for k, v in c.items(): step[key][k].append(v)
b2b = collections.defaultdict(lambda: collections.defaultdict(list))
shapes = [(4608, 1024, 1024, 3, 0, 0), (4608, 1024, 1024, 3, 0, 1), (4608, 4096, 1024, 3, 0, 0), (4608, 1024, 4096, 3, 0, 0), (4608, 4096, 1024, 3, 0, 1),
          (4608, 1024, 4096, 3, 0, 1), (1024, 1024, 4608, 4, 1, 1), (1024, 4096, 4608, 4, 1, 1), (4096, 1024, 4608, 4, 1, 1), (2304, 1024, 1024, 3, 0, 0)]
for tag in ("GRBM_GU", "TCC_HIT", "FETCH_S", "WRITE_S"):
    rows = gemm_rows(glob.glob('/tmp/ig_b_%s/*_counter_collection.csv' % tag)[0])
    per = len(rows) // len(shapes)                                # 12 launches per shape, in order
    for i, key in enumerate(shapes):
        for c in rows[i * per + 4:(i + 1) * per]:                 # (skip the first launches of a shape: they meet the pack's cache state)
            for k, v in c.items(): b2b[key][k].append(v)
print("%-34s | %-48s | %-48s" % ("M x N x K (prec, A km, B km)", "in the step: n, cycles, L2 hit, MB read, MB written", "back to back (plain epilogue): the same"))
for key in shapes:
    if key not in step: continue
    cells = []
    for D in (step, b2b):
        c = D[key]
        h, m = med(c['TCC_HIT_sum']), med(c['TCC_MISS_sum'])
        cells.append("%3d %9.0f  %5.1f %%  %8.1f %8.1f" % (len(c['GRBM_GUI_ACTIVE']), med(c['GRBM_GUI_ACTIVE']) / 8, 100 * h / max(h + m, 1),
                                                          med(c['FETCH_SIZE']) * 2 / 1024, med(c['WRITE_SIZE']) / 1024))
    print("%-34s | %-48s | %-48s" % ("%d x %d x %d (%d, %d, %d)" % key, cells[0], cells[1]))
PY
