"""Fills DESIGN.md section 5 from tools/design_s5.tmpl and the committed measurement files of a round tag:
python tools/fill_design.py r02_z   (profiles/<tag>_bench.json, profiles/<tag>_bench_kernel_stats.csv)."""
import csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03_z"
d = json.loads(open(os.path.join(ROOT, "profiles", tag + "_bench.json")).read().strip().splitlines()[-1])
r = d["roofline"]
stats = os.path.join(ROOT, "profiles", tag + "_bench_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
ng = sum(int(x["Calls"]) for x in rows if "gemm_pp_kernel" in x["Name"] or "gemm_gl_" in x["Name"] or "gemm_sp_kernel" in x["Name"])
gps = int(r.get("launches_per_step", 321))      # GEMM launches per step, counted live by bench.py
ns = max(1, round(ng / gps))
gemm_avg = sum(float(x["TotalDurationNs"]) for x in rows if "gemm_pp_kernel" in x["Name"] or "gemm_gl_" in x["Name"] or "gemm_sp_kernel" in x["Name"]) / ng / 1e3
table = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_table.py"), stats, str(ns), "22"],
                       capture_output=True, text=True).stdout.strip().splitlines()
head, body = table[0], table[1:]
launches = head.split("launches/step")[1].strip()
side = {x["config"]: x for x in d.get("side_configs", [])}
rep = {
    "@MS@": "%.1f" % d["ms_per_step"], "@CPS@": "%.1f" % d["value"], "@CPU@": "%.2f" % d["cpu_baseline"]["value"],
    "@B8@": "%.1f" % d["larger_batch"]["clips_per_s"], "@GUS@": "%.1f" % r["avg_launch_us"], "@GTF@": "%.0f" % r["achieved"],
    "@GFRAC@": "%.3f" % r["frac"], "@RUS@": "%.1f" % gemm_avg, "@GPS@": str(gps),
    "@GTFLOP@": "%.2f" % (r["algorithmic_gflop_per_launch"] * gps / 1e3),
    "@TABLE@": "| kernel | launches / step | ms / step | avg µs | share |\n|---|---|---|---|---|\n" + "\n".join(body) +
               "\n\n(" + head + ")",
    "@LAUNCHES@": launches,
    "@CFG1@": "%.1f" % side["cfg1"]["clips_per_s"] if "cfg1" in side else "n/a",
    "@CFGW@": "%.1f" % side["W"]["clips_per_s"] if "W" in side else "n/a",
    "@OPTMS@": "%.1f" % d["optimizer_step"]["ms"], "@OPTTB@": "%.1f" % (d["optimizer_step"]["hbm_GBps"] / 1e3),
    "@ITMS@": "%.1f" % d["optimizer_step"].get("train_iteration_ms_measured", float("nan")),
    "@ITCPS@": "%.1f" % d["optimizer_step"].get("train_iteration_clips_per_s", float("nan")),
}
tg = d.get("targets", {})
q = tg.get("qkv_pre_projection", {}).get("measured", [])
c = tg.get("cross_attention", {}).get("measured", [])
if len(q) >= 2 and len(c) >= 2:
    rep.update({"@T2A@": "%.1f" % (q[0]["GBps"] / 1e3), "@T2AF@": "%.0f %%" % (100 * q[0]["hbm_frac"]),
                "@T2BF@": "%.0f %%" % (100 * q[1]["hbm_frac"]), "@T2U@": "%.1f" % (q[0]["unfused_GBps_same_algorithmic_bytes"] / 1e3),
                "@T1A@": "%.0f" % c[0]["fwd_tflops"], "@T1B@": "%.0f" % c[1]["fwd_tflops"],
                "@T1AF@": "%.1f %%" % (100 * c[0]["fwd_mfma_issue_frac"]), "@T1BF@": "%.1f %%" % (100 * c[1]["fwd_mfma_issue_frac"])})
t = open(os.path.join(ROOT, "tools", "design_s5.tmpl")).read()
for k, v in rep.items():
    t = t.replace(k, v)
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a, b = s.index("## 5. Measurement (round"), s.index("## 6. Next (ordered)")
open(p, "w").write(s[:a] + t + s[b:])
print("DESIGN.md section 5 filled from", tag, ": %.1f ms/step, %s launches/step" % (d["ms_per_step"], launches))
