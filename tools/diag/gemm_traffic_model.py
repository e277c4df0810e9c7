"""What the PMC traffic of the GEMM launches is made of (bench line: roofline.traffic ~ 100 MB per launch against ~46 MB of
algorithmic bytes).  CPU-only model over the per-shape table of a profiled step (profiles/rNN_z_gemm_shapes.txt): every XCD has
its own 4 MB L2, the kernel hands each XCD a contiguous run of tiles (row-major over (row tile, column tile)), so the operand
the run shares is fetched once per XCD and the other operand once; split-K products also write and re-read their fp32 slabs.
FETCH_SIZE counts what the L2s request from the fabric -- Infinity-Cache hits included -- so this replication shows up in it
without being HBM traffic."""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_z_gemm_shapes.txt"
rows = []
for l in open(path):
    m = re.match(r'\s*(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+\|\s+([\d.]+)\s+([\d.]+)', l)
    if m:
        rows.append([float(x) for x in m.groups()])
alg = rep = n = 0.0
for M, N, K, nb, BM, ks, pr, ak, bk, tap, cnt, us in rows:
    parts = 1 if pr == 4 else 2
    a, b = M * K * 2 * parts * nb, N * K * 2 * parts * nb
    c = M * N * 4 * nb * (1 if ks == 1 else ks + 2)          # split-K: slabs written, read back, result written
    tm, tn = -(-M // BM), -(-N // 128)
    chunk = max(1.0, tm * tn / 8.0)                          # tiles per XCD
    fetch_b = b * max(1.0, min(8, tm * tn) * min(1.0, chunk / tn))
    fetch_a = a * (1.0 if chunk >= tn else min(8.0, tn / chunk))
    alg += (a + b + M * N * 4 * nb) * cnt
    rep += (fetch_a + fetch_b + c) * cnt
    n += cnt
print("GEMM launches per step %.0f | algorithmic %.1f MB per launch | with per-XCD operand fetches and split-K slabs %.1f MB per launch"
      % (n, alg / n / 1e6, rep / n / 1e6))
