"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, launches and counter sum/mean."""
import csv, sys, collections, json
path, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
with open(path) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:90]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
rows = [{"kernel": k, "launches": len(cnt[k]), **{c: v for c, v in acc[k].items()}} for k in acc]
rows.sort(key=lambda r: -max(v for kk, v in r.items() if kk not in ("kernel", "launches")))
json.dump(rows, open(out, "w"), indent=1)
for r in rows[:12]:
    print(r)
