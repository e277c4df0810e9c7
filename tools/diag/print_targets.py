import json, sys
tag = sys.argv[1]
d = json.load(sys.stdin)
for m in d["qkv_pre"]:
    print(tag, m["shape"], m["stride"], "%.1f us  %.0f GB/s  frac %.3f" % (m["us"], m["GBps"], m["hbm_frac"]))
