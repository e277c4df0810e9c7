"""Per-step kernel table from a rocprofv3 kernel_stats.csv of `bench.py` (steps = gemm launches / 325 style normalisation
is avoided: pass the number of steps the trace contains)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = sum(float(r['TotalDurationNs']) for r in rows)
groups = {}
def grp(n):
    n = n.replace('(anonymous namespace)::', '')
    for k in ('gemm_gl_group_kernel', 'gemm_gl_kernel', 'gemm_sp_kernel', 'gemm_pp_kernel', 'amax_kernel', 'pack_kc', 'pack_tr', 'splitk_reduce', 'attn_bwd_dkdv', 'attn_bwd_dq', 'attn_fwd', 'attn_delta',
              'reduce_rows', 'ln_bwd', 'ln_fwd', 'act_bwd', 'relshift', 'dwconv3', 'colsum', 'permute3', 'scale_add', 'adamw', 'sqnorm',
              'maxpool', 'transpose2d', 'add_pe', 'axpby', 'mask_rows'):
        if k in n: return k
    if n.startswith('void at::') or 'rocprim' in n or 'rocclr' in n: return 'ATen glue'
    return n[:40]
for r in rows:
    g = grp(r['Name']); a = groups.setdefault(g, [0, 0.0]); a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
print('GPU kernel ms/step %.2f, launches/step %.0f' % (tot / steps / 1e6, sum(c for c, t in groups.values()) / steps))
for g, (c, t) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 20]:
    print("| `%s` | %.0f | %.2f | %.1f | %.1f %% |" % (g, c / steps, t / steps / 1e6, t / c / 1e3, 100 * t / tot))
