"""DESIGN.md section 5 from tools/design_s5_r06.tmpl and the committed bundle of a tag:  python tools/fill_design_r06.py r06_z
(profiles/<tag>_bench.json, <tag>_bench_kernel_stats.csv; the per-step kernel table comes from <tag>_step_kernel_stats.csv, a
tools/prof_step.sh run of the same build)."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06_z"
d = json.loads(open(os.path.join(ROOT, "profiles", tag + "_bench.json")).read().strip().splitlines()[-1])
r = d["roofline"]
rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_bench_kernel_stats.csv"))))
isg = lambda n: "gemm_pp_kernel" in n or "gemm_gl_" in n
ng = sum(int(x["Calls"]) for x in rows if isg(x["Name"]))
rus = sum(float(x["TotalDurationNs"]) for x in rows if isg(x["Name"])) / ng / 1e3
side = {x["config"]: x for x in d.get("side_configs", [])}
q = d["targets"]["qkv_pre_projection"]["measured"]; c = d["targets"]["cross_attention"]["measured"]
o = d["optimizer_step"]; st = d["strict"]
rep = {"@MS@": "%.2f" % d["ms_per_step"], "@CPS@": "%.1f" % d["value"], "@EAGER@": "%.1f" % d["eager"]["ms_per_step"],
       "@HOST@": "%.1f" % d["host_enqueue_ms"], "@CPU@": "%.3f" % d["cpu_baseline"]["value"], "@OPT@": "%.2f" % o["ms"],
       "@ITER@": "%.1f" % o["train_iteration_ms_measured"], "@ITCPS@": "%.1f" % o["train_iteration_clips_per_s"],
       "@STRICT@": "%.1f" % st["fwd_bwd_ms_static_weights"], "@B8MS@": "%.1f" % d["larger_batch"]["ms_per_step"],
       "@B8@": "%.1f" % d["larger_batch"]["clips_per_s"], "@CFG1@": "%.2f" % side["cfg1"]["ms_per_step"], "@W@": "%.1f" % side["W"]["ms_per_step"],
       "@GPS@": str(int(r["launches_per_step"])), "@GGF@": "%.2f" % r["algorithmic_gflop_per_launch"], "@GUS@": "%.1f" % r["avg_launch_us"],
       "@GTF@": "%.0f" % r["achieved"], "@FRAC@": "%.3f" % r["frac"], "@RUS@": "%.1f" % rus, "@ISSUE@": "%.3f" % r["mfma_issue_frac"],
       "@TRAF@": "%.0f" % (r["traffic"] / 1e6), "@MODEL@": "%.3f" % d["model_mfma_frac"],
       "@Q2@": "%.2f" % q[0]["hbm_frac"], "@Q8@": "%.2f" % q[1]["hbm_frac"], "@C2@": "%.2f" % c[0]["fwd_mfma_issue_frac"],
       "@C8@": "%.2f" % c[1]["fwd_mfma_issue_frac"], "@GMS@": "%.1f" % r["kernel_ms_per_step"]}
t = open(os.path.join(ROOT, "tools", "design_s5_r06.tmpl")).read()
for k, v in rep.items():
    t = t.replace(k, v)
left = re.findall(r"@[A-Z0-9_]+@", t)
assert not left, left
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
i = s.index("## 5. Measurement"); i = s.index("\n", i) + 1
j = s.index("## 6. Multi-GPU")
open(p, "w").write(s[:i] + "\n" + t.rstrip("\n") + "\n\n" + s[j:])
print("DESIGN.md section 5 filled from", tag)
