"""gpurun_out/<TAG>_* -> profiles/ (bench line, kernel stats, PMC traffic summary) and a printed recap.  TAG from argv[1] (default r02_z)."""
import json, csv, shutil, os, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03_z"
RND = TAG.split("_")[0]
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
g = lambda f: os.path.join(R, 'gpurun_out', f)
f = json.load(open(g(TAG + '_pmc_fetch.json'))); w = json.load(open(g(TAG + '_pmc_write.json')))
def agg(rows, key):
    tot = n = 0
    for r in rows:
        if 'gemm_pp_kernel' in r['kernel'] or 'gemm_gl_' in r['kernel'] or 'gemm_sp_kernel' in r['kernel']:
            tot += r[key]; n += r['launches']
    return tot, n
ft, fn = agg(f, 'FETCH_SIZE'); wt, wn = agg(w, 'WRITE_SIZE')
per = 2 * ft * 1024 / fn + wt * 1024 / wn
rows = list(csv.DictReader(open(g(TAG + '_bench_kernel_stats.csv'))))
gg = [r for r in rows if 'gemm_pp_kernel' in r['Name'] or 'gemm_gl_' in r['Name'] or 'gemm_sp_kernel' in r['Name']]
tot = sum(float(r['TotalDurationNs']) for r in gg); calls = sum(int(r['Calls']) for r in gg)
out = {"gemm_kernels": {"launches_fetch_pass": fn, "launches_write_pass": wn, "FETCH_SIZE_KB_sum": ft, "WRITE_SIZE_KB_sum": wt,
       "hbm_bytes_per_launch": per, "rocprof_avg_launch_us": tot / calls / 1e3, "rocprof_launches": calls,
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --no-cpu-baseline --no-targets --extra-batch 0 "
               "--steps 2 --warmup 1`; FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section), "
               "units KB; summed over all gemm_gl_kernel + gemm_pp_kernel instantiations; rocprof_avg from `rocprofv3 --kernel-trace --stats -- python3 "
               "bench.py --no-cpu-baseline --no-targets --extra-batch 0`"}}
json.dump(out, open(os.path.join(R, 'profiles', RND + '_pmc_traffic.json'), 'w'), indent=1)
for n in ('bench_kernel_stats.csv', 'pmc_fetch.json', 'pmc_write.json'):
    shutil.copy(g(TAG + '_' + n), os.path.join(R, 'profiles', TAG + '_' + n))
for n in ('gemm_shapes.txt', 'gemm_mfma.txt', 'targets_kernel_stats.csv', 'targets_pmc_fetch.json', 'targets_pmc_write.json', 'attn_insts.txt', 'gpu_tests.txt'):       # optional extras
    if os.path.exists(g(TAG + '_' + n)):
        shutil.copy(g(TAG + '_' + n), os.path.join(R, 'profiles', TAG + '_' + n))
d = json.loads(open(g(TAG + '_bench.json')).read().strip().split('\n')[-1])
d['roofline']['traffic'] = per            # the PMC passes of THIS bundle (bench.py read the previous summary)
open(os.path.join(R, 'profiles', TAG + '_bench.json'), 'w').write(json.dumps(d) + '\n')
print(json.dumps({k: d[k] for k in ('value', 'ms_per_step', 'roofline', 'cpu_baseline', 'optimizer_step', 'larger_batch')}, indent=1))
print('HBM MB/launch %.1f  rocprof avg us %.1f over %d launches (%.1f steps)' % (per / 1e6, tot / calls / 1e3, calls, calls / d['roofline']['launches_per_step']))
