// 1-D NMS / soft-NMS on the device: one 1024-thread workgroup per class, all classes in one launch.
// Replaces the CPU extension nms_1d_cpu (MQ/libs/utils/csrc/nms_cpu.cpp) and the per-class python
// loop around it (MQ/libs/utils/nms.py:124-152).  Index outputs are bit-exact: the kernels keep the
// reference's array semantics (stable descending order for hard NMS; first-maximum pick, swap and
// swap-with-last removal for soft-NMS), only the inner O(N) loops run in parallel.
// Compiled with -ffp-contract=off so the IoU / decay arithmetic rounds like the reference's C++.
#include "common.h"

namespace {

constexpr int NT = 1024;
constexpr int NW = NT / 64;

struct Ws {
  float* x1; float* x2; float* sc; float* ar;
  int* ind; int* tmp; unsigned char* dead;
};

__device__ __forceinline__ Ws carve(void* ws, long n_total, long off) {
  float* f = reinterpret_cast<float*>(ws);
  Ws w;
  w.x1 = f + 0 * n_total + off;
  w.x2 = f + 1 * n_total + off;
  w.sc = f + 2 * n_total + off;
  w.ar = f + 3 * n_total + off;
  w.ind = reinterpret_cast<int*>(f + 4 * n_total) + off;
  w.tmp = reinterpret_cast<int*>(f + 5 * n_total) + off;
  w.dead = reinterpret_cast<unsigned char*>(f + 6 * n_total) + off;
  return w;
}

// expf exactly as the host libm computes it.  The reference's soft-NMS decay is std::exp(float) = glibc expf
// (nms_cpu.cpp:141); with 30 000 candidates two decayed scores regularly land within 1 ulp of each other, so a
// 1-ulp different exp flips pick order.  glibc >= 2.27 uses the table-driven double-precision algorithm of the
// ARM optimized routines (N = 32 entries of 2^(i/32), cubic polynomial, one final rounding to float); restated
// here in fp64 it is bit-identical to libm on the 500 000 inputs checked in tests/test_oracle_nms.py.
__device__ const unsigned long long kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

template <typename TabT>
__device__ __forceinline__ float expf_libm_tab(float x, const TabT* tab) {
  if (!(x > -80.f && x < 80.f)) return expf(x);          // far tails / NaN: not reachable from an IoU
  const double N = 32.0;
  const double z = (0x1.71547652b82fep+0 * N) * (double)x;
  double kd = z + 0x1.8p+52;
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= 0x1.8p+52;
  const double r = z - kd;
  const unsigned long long t = tab[ki % 32] + (ki << 47);
  const double s = __longlong_as_double((long long)t);
  const double c0 = 0x1.c6af84b912394p-5 / N / N / N, c1 = 0x1.ebfce50fac4f3p-3 / N / N, c2 = 0x1.62e42ff0c52d6p-1 / N;
  const double zz = c0 * r + c1;
  const double r2 = r * r;
  double y = c2 * r + 1.0;
  y = zz * r2 + y;
  y = y * s;
  return (float)y;
}

__device__ __forceinline__ float expf_libm(float x) { return expf_libm_tab(x, kExp2fTab); }

__device__ __forceinline__ float iou_1d(float ix1, float ix2, float iarea, float jx1, float jx2, float jarea) {
  const float xx1 = fmaxf(ix1, jx1);
  const float xx2 = fminf(ix2, jx2);
  const float inter = fmaxf(0.f, xx2 - xx1);
  return inter / (iarea + jarea - inter);
}

// block-wide exclusive scan of one int per thread; returns the exclusive prefix, *total = sum
__device__ int block_excl_scan(int v, int* total, int* lds /* >= NW+1 ints */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();  // protect lds reuse
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int w = 0; w < NW; ++w) { const int t = lds[w]; lds[w] = run; run += t; }
    lds[NW] = run;
  }
  __syncthreads();
  *total = lds[NW];
  return lds[wave] + inc - v;
}

// ------------------------------------------------------------------------------------ hard NMS
// Greedy NMS keeps a candidate iff no HIGHER-ranked kept candidate overlaps it by >= thr (nms_cpu.cpp:36-55); the kept set
// is a function of the score order and that predicate alone, so the work can be regrouped freely as long as every
// (kept i, later j) pair is tested with the reference's arithmetic.  Three kernels (round 4; rounds 1-3 ran everything in
// one workgroup per class: 65 ms for one class of 30 000 candidates, slower than the CPU extension's 39 ms):
//   nms_rank_kernel   all CUs.  One thread per candidate: its stable descending rank inside its class (ties: lower input
//                     index first == ATen's CPU sort), scatter of (x1, x2, area, index) into score order.
//   nms_sweep_kernel  one workgroup per class, chunk c of CH candidates in score order: the candidates that survived the
//                     earlier chunks' keeps are swept greedily inside the chunk -- the loop visits KEPT candidates only
//                     (next set bit of an LDS bit mask), each visit tests all later candidates of the chunk in parallel.
//   nms_apply_kernel  all CUs.  The keeps chunk c just produced are applied to every candidate of the later chunks.
// The host issues rank, then (sweep, apply) per chunk; a class shorter than c * CH leaves its workgroups at once.
constexpr int CH = 4096;                 // candidates per chunk: 4 per thread, 64 mask words
constexpr int CHW = CH / 64;

struct HardAux { long* kbase; };         // per class: number of keeps before the current chunk

// BY_X1 (round 5, the pre-pass of softnms_reg_kernel): the key is -x1 (rank = position in ascending-start order, ties by input
// index) and the score travels with the candidate; no counters are touched.
template <bool BY_X1>
__global__ __launch_bounds__(256) void nms_rank_kernel(const float* __restrict__ segs, const float* __restrict__ scores,
                                                       const long* __restrict__ seg_off, int nseg,
                                                       long* __restrict__ out_cnt, long* __restrict__ kbase,
                                                       void* ws_raw, long n_total) {
  __shared__ float s_sc[1024];
  const long g0 = (long)blockIdx.x * 256;
  if (!BY_X1 && blockIdx.x == 0)
    for (int k = threadIdx.x; k < nseg; k += 256) { out_cnt[k] = 0; kbase[k] = 0; }
  // classes this block's 256 candidates belong to: [c_lo, c_hi]; candidates are class-sorted, so a block spans few classes
  const long g = g0 + threadIdx.x;
  int cls = -1;
  if (g < n_total) {
    int lo = 0, hi = nseg - 1;             // last class with seg_off[c] <= g
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (seg_off[mid] <= g) lo = mid; else hi = mid - 1;
    }
    cls = lo;
  }
  // walk the classes present in this block one after the other (uniform loop: every thread takes part in the staging)
  const long glast = (g0 + 255 < n_total ? g0 + 255 : n_total - 1);
  int c_first, c_last;
  {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (seg_off[mid] <= g0) lo = mid; else hi = mid - 1; }
    c_first = lo;
    lo = 0; hi = nseg - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (seg_off[mid] <= glast) lo = mid; else hi = mid - 1; }
    c_last = lo;
  }
  for (int c = c_first; c <= c_last; ++c) {
    const long off = seg_off[c];
    const int n = (int)(seg_off[c + 1] - off);
    const bool mine = cls == c;
    const int i = mine ? (int)(g - off) : 0;
    const float si = mine ? (BY_X1 ? -segs[2 * (off + i)] : scores[off + i]) : 0.f;
    int rank = 0;
    for (int t0 = 0; t0 < n; t0 += 1024) {
      __syncthreads();
      for (int k = threadIdx.x; k < 1024; k += 256)
        s_sc[k] = t0 + k < n ? (BY_X1 ? -segs[2 * (off + t0 + k)] : scores[off + t0 + k]) : -INFINITY;
      __syncthreads();
      const int cnt = min(1024, n - t0);
      if (mine) {
        // j < i <=> t0 + k < i: split the tile at i so the inner loops carry one compare each
        const int before = max(0, min(cnt, i - t0));
        int r = 0;
        for (int k = 0; k < before; ++k) r += s_sc[k] >= si;        // j < i: ahead on ties
        for (int k = before; k < cnt; ++k) r += s_sc[k] > si;       // j >= i (j == i: not greater)
        rank += r;
      }
    }
    if (mine) {
      Ws w = carve(ws_raw, n_total, off);
      const float a = segs[2 * (off + i)], b = segs[2 * (off + i) + 1];
      w.ind[rank] = i;
      w.x1[rank] = a;
      w.x2[rank] = b;
      w.ar[rank] = b - a + 1e-6f;
      w.dead[rank] = 0;
      if (BY_X1) w.sc[rank] = scores[off + i];
    }
  }
}

__global__ __launch_bounds__(NT) void nms_sweep_kernel(const long* __restrict__ seg_off, float thr, int chunk,
                                                       long* __restrict__ out_idx, long* __restrict__ out_cnt,
                                                       long* __restrict__ kbase, void* ws_raw, long n_total) {
  __shared__ float s_x1[CH], s_x2[CH], s_ar[CH];
  __shared__ unsigned long long s_alive[CHW];
  __shared__ int s_cur;
  const int cls = blockIdx.x;
  const long off = seg_off[cls];
  const int n = (int)(seg_off[cls + 1] - off);
  const int c0 = chunk * CH;
  if (c0 >= n) return;
  const int m = min(CH, n - c0);
  Ws w = carve(ws_raw, n_total, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // candidate b of the chunk belongs to thread b % 1024; mask word b / 64 = wave + 16 r is written by this wave alone
#pragma unroll
  for (int r = 0; r < CH / NT; ++r) {
    const int b = threadIdx.x + NT * r;
    bool alive = false;
    if (b < m) {
      s_x1[b] = w.x1[c0 + b]; s_x2[b] = w.x2[c0 + b]; s_ar[b] = w.ar[c0 + b];
      alive = !w.dead[c0 + b];
    }
    const unsigned long long bal = __ballot(alive);
    if (lane == 0) s_alive[wave + NW * r] = bal;
  }
  long kcount = out_cnt[cls];             // keeps so far (uniform)
  if (threadIdx.x == 0) kbase[cls] = kcount;
  __syncthreads();
  // first alive candidate: wave 0 looks at the 64 words
  auto next_alive = [&](int after) {      // smallest set bit > after, CH when none; call from every thread of wave 0
    unsigned long long v = s_alive[lane];
    const int wa = (after + 1) >> 6, ba = (after + 1) & 63;
    if (lane < wa) v = 0;
    else if (lane == wa) v &= ba ? (~0ull << ba) : ~0ull;
    const unsigned long long nz = __ballot(v != 0);
    if (nz == 0) return CH;
    const int fw = __ffsll((long long)nz) - 1;
    const unsigned long long word = __shfl(v, fw, 64);
    return fw * 64 + __ffsll((long long)word) - 1;
  };
  if (wave == 0) {
    const int c = next_alive(-1);
    if (lane == 0) s_cur = c;
  }
  __syncthreads();
  int cur = s_cur;
  while (cur < m) {
    const float ix1 = s_x1[cur], ix2 = s_x2[cur], ia = s_ar[cur];
    if (threadIdx.x == 0) {
      out_idx[off + kcount] = (long)w.ind[c0 + cur];
      w.tmp[kcount] = c0 + cur;           // keep list in score order (positions): read by nms_apply_kernel
    }
    ++kcount;
#pragma unroll
    for (int r = 0; r < CH / NT; ++r) {
      const int b = threadIdx.x + NT * r;
      const int wi = wave + NW * r;
      if ((wi << 6) + 63 > cur) {         // (uniform per wave) this word has candidates after cur
        const unsigned long long old = s_alive[wi];
        bool kill = false;
        if (b > cur && ((old >> lane) & 1ull)) kill = iou_1d(ix1, ix2, ia, s_x1[b], s_x2[b], s_ar[b]) >= thr;
        const unsigned long long k = __ballot(kill);
        if (lane == 0 && k) s_alive[wi] = old & ~k;
      }
    }
    __syncthreads();
    if (wave == 0) {
      const int c = next_alive(cur);
      if (lane == 0) s_cur = c;
    }
    __syncthreads();
    cur = s_cur;
  }
  if (threadIdx.x == 0) out_cnt[cls] = kcount;
}

__global__ __launch_bounds__(256) void nms_apply_kernel(const long* __restrict__ seg_off, float thr, int chunk,
                                                        const long* __restrict__ out_cnt, const long* __restrict__ kbase,
                                                        void* ws_raw, long n_total) {
  __shared__ float s_x1[1024], s_x2[1024], s_ar[1024];
  const int cls = blockIdx.y;
  const long off = seg_off[cls];
  const int n = (int)(seg_off[cls + 1] - off);
  const int first = (chunk + 1) * CH;     // candidates of the later chunks
  const int j0 = first + blockIdx.x * 256;
  if (j0 >= n) return;
  Ws w = carve(ws_raw, n_total, off);
  const int k0 = (int)kbase[cls], k1 = (int)out_cnt[cls];       // the keeps chunk `chunk` added
  const int j = j0 + threadIdx.x;
  bool alive = j < n && !w.dead[j];
  const float jx1 = alive ? w.x1[j] : 0.f, jx2 = alive ? w.x2[j] : 0.f, ja = alive ? w.ar[j] : 1.f;
  for (int t0 = k0; t0 < k1; t0 += 1024) {
    __syncthreads();
    const int cnt = min(1024, k1 - t0);
    for (int k = threadIdx.x; k < cnt; k += 256) {
      const int p = w.tmp[t0 + k];
      s_x1[k] = w.x1[p]; s_x2[k] = w.x2[p]; s_ar[k] = w.ar[p];
    }
    __syncthreads();
    if (__ballot(alive) != 0) {
      for (int k = 0; k < cnt && alive; ++k)
        if (iou_1d(s_x1[k], s_x2[k], s_ar[k], jx1, jx2, ja) >= thr) alive = false;
    }
  }
  if (j < n && !alive) w.dead[j] = 1;
}

// ------------------------------------------------------------------------------------ soft NMS
__global__ __launch_bounds__(NT) void softnms_kernel(const float* __restrict__ segs,
                                                     const float* __restrict__ scores,
                                                     const long* __restrict__ seg_off, float thr,
                                                     float sigma, float min_score, int method,
                                                     long max_num, float* __restrict__ dets,
                                                     long* __restrict__ out_idx,
                                                     long* __restrict__ out_cnt, void* ws_raw,
                                                     long n_total, int n_lo, int n_hi) {
  __shared__ int s_scan[NW + 1];
  __shared__ float s_val[NW];
  __shared__ int s_pos[NW];
  const int cls = blockIdx.x;
  const long off = seg_off[cls];
  const int n = (int)(seg_off[cls + 1] - off);
  if (n <= 0) {
    if (threadIdx.x == 0) out_cnt[cls] = 0;
    return;
  }
  if (n <= n_lo || n > n_hi) return;                // (another soft kernel of this call owns the class)
  Ws w = carve(ws_raw, n_total, off);
  const float* sg = segs + off * 2;
  for (int i = threadIdx.x; i < n; i += NT) {
    w.x1[i] = sg[2 * i];
    w.x2[i] = sg[2 * i + 1];
    w.ar[i] = sg[2 * i + 1] - sg[2 * i] + 1e-6f;
    w.sc[i] = scores[off + i];
    w.ind[i] = i;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int nsegs = n;  // uniform
  int i = 0;
  for (; i < nsegs; ++i) {
    if (max_num > 0 && i >= max_num) break;
    // (a) first maximum of sc[i:nsegs)  (nms_cpu.cpp:91-101: strict '<' keeps the lowest position)
    float bv = -INFINITY;
    int bp = 0x7fffffff;
    for (int p = i + threadIdx.x; p < nsegs; p += NT) {
      const float v = w.sc[p];
      if (bp == 0x7fffffff || v > bv) { bv = v; bp = p; }  // ascending p per thread: ties keep lowest
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int op = __shfl_xor(bp, o, 64);
      if (op != 0x7fffffff && (bp == 0x7fffffff || ov > bv || (ov == bv && op < bp))) { bv = ov; bp = op; }
    }
    if (lane == 0) { s_val[wave] = bv; s_pos[wave] = bp; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float v = s_val[0];
      int p = s_pos[0];
      for (int k = 1; k < NW; ++k) {
        const int op = s_pos[k];
        const float ov = s_val[k];
        if (op != 0x7fffffff && (p == 0x7fffffff || ov > v || (ov == v && op < p))) { v = ov; p = op; }
      }
      // (b) swap slot i <-> p, emit det row i  (nms_cpu.cpp:103-121)
      const float ax1 = w.x1[p], ax2 = w.x2[p], asc = w.sc[p], aar = w.ar[p];
      const int aind = w.ind[p];
      w.x1[p] = w.x1[i]; w.x2[p] = w.x2[i]; w.sc[p] = w.sc[i]; w.ar[p] = w.ar[i]; w.ind[p] = w.ind[i];
      w.x1[i] = ax1; w.x2[i] = ax2; w.sc[i] = asc; w.ar[i] = aar; w.ind[i] = aind;
      float* d = dets + (off + i) * 3;
      d[0] = ax1; d[1] = ax2; d[2] = asc;
      out_idx[off + i] = (long)aind;
    }
    __syncthreads();
    // (c) decay every remaining score once (nms_cpu.cpp:124-143)
    const float ix1 = w.x1[i], ix2 = w.x2[i], ia = w.ar[i];
    const int first = i + 1;
    const int span = nsegs - first;
    const int chunk = (span + NT - 1) / NT;
    const int lo = first + min(span, (int)threadIdx.x * chunk), hi = min(nsegs, lo + chunk);
    int alive_cnt = 0;
    for (int p = lo; p < hi; ++p) {
      const float ovr = iou_1d(ix1, ix2, ia, w.x1[p], w.x2[p], w.ar[p]);
      float weight = 1.f;
      if (method == 0) { if (ovr >= thr) weight = 0.f; }
      else if (method == 1) { if (ovr >= thr) weight = 1.f - ovr; }
      else if (method == 2) { weight = expf_libm(-(ovr * ovr) / sigma); }
      const float ns = w.sc[p] * weight;
      w.sc[p] = ns;
      const bool dd = ns < min_score;
      w.dead[p] = dd;
      alive_cnt += !dd;
    }
    // (d) swap-with-last removals (nms_cpu.cpp:146-154), done as: k-th hole (ascending) <- k-th
    // surviving element counted from the end; identical final array to the sequential loop.
    int n_alive;
    int a_before = block_excl_scan(alive_cnt, &n_alive, s_scan);  // alive in [first, lo)
    const int new_n = first + n_alive;
    if (n_alive != span) {
      int a = a_before;
      for (int p = lo; p < hi; ++p) {
        if (!w.dead[p]) {
          if (p >= new_n) w.tmp[n_alive - a - 1] = p;  // rank from the end among sources
          ++a;
        }
      }
      __syncthreads();
      a = a_before;
      for (int p = lo; p < hi; ++p) {
        if (w.dead[p]) {
          if (p < new_n) {
            const int k = (p - first) - a;  // dead positions before p
            const int src = w.tmp[k];
            w.x1[p] = w.x1[src]; w.x2[p] = w.x2[src]; w.sc[p] = w.sc[src]; w.ar[p] = w.ar[src];
            w.ind[p] = w.ind[src];
          }
        } else {
          ++a;
        }
      }
      nsegs = new_n;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out_cnt[cls] = i;
}


// ------------------------------------------------------------------------------------ soft NMS, row-strided (round 4)
// Same array semantics as softnms_kernel above, restructured around what bounded it (76 us per pick at n = 30 000, i.e. one
// memory round trip per loop iteration): position p of the remaining range belongs to thread (p - first) % 1024 -- every
// access of a wavefront is one coalesced line -- and every loop requests the values of several rows before it uses the
// first (clamped, unconditional loads; the stores follow the loads of a batch).  The ordered prefix "alive before p" that
// the swap-with-last emulation needs comes from wavefront ballots: count per (row, wave) -> one exclusive scan in LDS.
// n <= SROWS * 1024; larger classes take softnms_kernel.
constexpr int SROWS = 64;

__global__ __launch_bounds__(NT) void softnms_rows_kernel(const float* __restrict__ segs,
                                                          const float* __restrict__ scores,
                                                          const long* __restrict__ seg_off, float thr,
                                                          float sigma, float min_score, int method,
                                                          long max_num, float* __restrict__ dets,
                                                          long* __restrict__ out_idx,
                                                          long* __restrict__ out_cnt, void* ws_raw,
                                                          long n_total, int n_lo, int n_hi) {
  __shared__ int s_cnt[SROWS * NW + 1];
  __shared__ float s_val[NW];
  __shared__ int s_pos[NW];
  __shared__ float s_pick[3];
  const int cls = blockIdx.x;
  const long off = seg_off[cls];
  const int n = (int)(seg_off[cls + 1] - off);
  if (n <= 0) {
    if (threadIdx.x == 0) out_cnt[cls] = 0;
    return;
  }
  if (n <= n_lo || n > n_hi) return;                // (another soft kernel of this call owns the class)
  Ws w = carve(ws_raw, n_total, off);
  float* __restrict__ X1 = w.x1; float* __restrict__ X2 = w.x2; float* __restrict__ SC = w.sc; float* __restrict__ AR = w.ar;
  int* __restrict__ IND = w.ind; int* __restrict__ TMP = w.tmp;
  const float* sg = segs + off * 2;
  for (int i = threadIdx.x; i < n; i += NT) {
    const float a = sg[2 * i], b = sg[2 * i + 1];
    X1[i] = a; X2[i] = b; AR[i] = b - a + 1e-6f; SC[i] = scores[off + i]; IND[i] = i;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  int nsegs = n;  // uniform
  int i = 0;
  for (; i < nsegs; ++i) {
    if (max_num > 0 && i >= max_num) break;
    // (a) first maximum of sc[i:nsegs)  (nms_cpu.cpp:91-101: strict '<' keeps the lowest position)
    float bv = -INFINITY;
    int bp = 0x7fffffff;
    for (int p0 = i + (int)threadIdx.x; p0 < nsegs; p0 += 8 * NT) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = SC[min(p0 + u * NT, nsegs - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + u * NT;
        if (p < nsegs && (bp == 0x7fffffff || v[u] > bv)) { bv = v[u]; bp = p; }   // ascending p per thread: ties keep lowest
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int op = __shfl_xor(bp, o, 64);
      if (op != 0x7fffffff && (bp == 0x7fffffff || ov > bv || (ov == bv && op < bp))) { bv = ov; bp = op; }
    }
    if (lane == 0) { s_val[wave] = bv; s_pos[wave] = bp; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float v = s_val[0];
      int p = s_pos[0];
      for (int k = 1; k < NW; ++k) {
        const int op = s_pos[k];
        const float ov = s_val[k];
        if (op != 0x7fffffff && (p == 0x7fffffff || ov > v || (ov == v && op < p))) { v = ov; p = op; }
      }
      // (b) swap slot i <-> p, emit det row i  (nms_cpu.cpp:103-121): all ten values requested before the first store
      const float ax1 = X1[p], ax2 = X2[p], asc = SC[p], aar = AR[p];
      const int aind = IND[p];
      const float bx1 = X1[i], bx2 = X2[i], bsc = SC[i], bar = AR[i];
      const int bind = IND[i];
      X1[p] = bx1; X2[p] = bx2; SC[p] = bsc; AR[p] = bar; IND[p] = bind;
      X1[i] = ax1; X2[i] = ax2; SC[i] = asc; AR[i] = aar; IND[i] = aind;
      float* d = dets + (off + i) * 3;
      d[0] = ax1; d[1] = ax2; d[2] = asc;
      out_idx[off + i] = (long)aind;
      s_pick[0] = ax1; s_pick[1] = ax2; s_pick[2] = aar;
    }
    __syncthreads();
    // (c) decay every remaining score once (nms_cpu.cpp:124-143); row r = positions first + r * 1024 + thread
    const float ix1 = s_pick[0], ix2 = s_pick[1], ia = s_pick[2];
    const int first = i + 1;
    const int span = nsegs - first;
    const int rows = (span + NT - 1) / NT;
    unsigned long long deadmask = 0, inmask = 0;          // bit r: this thread's candidate of row r is dead / exists
    for (int r0 = 0; r0 < rows; r0 += 8) {
      float jx1[8], jx2[8], ja[8], js[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = min(first + (r0 + u) * NT + (int)threadIdx.x, nsegs - 1);
        jx1[u] = X1[p]; jx2[u] = X2[p]; ja[u] = AR[p]; js[u] = SC[p];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + u;
        const int p = first + r * NT + (int)threadIdx.x;
        const bool in = r < rows && p < nsegs;
        bool dd = false;
        if (in) {
          const float ovr = iou_1d(ix1, ix2, ia, jx1[u], jx2[u], ja[u]);
          float weight = 1.f;
          if (method == 0) { if (ovr >= thr) weight = 0.f; }
          else if (method == 1) { if (ovr >= thr) weight = 1.f - ovr; }
          else if (method == 2) { weight = expf_libm(-(ovr * ovr) / sigma); }
          const float ns = js[u] * weight;
          SC[p] = ns;
          dd = ns < min_score;
        }
        if (r < rows) {                                    // (uniform)
          const unsigned long long al = __ballot(in && !dd);
          if (lane == 0) s_cnt[r * NW + wave] = __popcll(al);
          if (in) inmask |= 1ull << r;
          if (dd) deadmask |= 1ull << r;
        }
      }
    }
    __syncthreads();
    // exclusive scan of the (row, wave) counts in position order, by wave 0
    const int ent = rows * NW;
    if (wave == 0) {
      const int per = (ent + 63) / 64;
      const int e0 = lane * per, e1 = min(ent, e0 + per);
      int sum = 0;
      for (int e = e0; e < e1; ++e) sum += s_cnt[e];
      int inc = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      int run = inc - sum;
      for (int e = e0; e < e1; ++e) { const int t = s_cnt[e]; s_cnt[e] = run; run += t; }
      if (lane == 63) s_cnt[ent] = inc;
    }
    __syncthreads();
    const int n_alive = s_cnt[ent];
    const int new_n = first + n_alive;
    // (d) swap-with-last removals (nms_cpu.cpp:146-154) as: k-th hole (ascending) <- k-th surviving element from the end
    if (n_alive != span) {
      for (int r = 0; r < rows; ++r) {
        const bool in = (inmask >> r) & 1ull, dd = (deadmask >> r) & 1ull;
        const unsigned long long al = __ballot(in && !dd);
        const int p = first + r * NT + (int)threadIdx.x;
        if (in && !dd && p >= new_n) {
          const int a = s_cnt[r * NW + wave] + __popcll(al & lt);
          TMP[n_alive - a - 1] = p;
        }
      }
      __syncthreads();
      for (int r = 0; r < rows; ++r) {
        const bool in = (inmask >> r) & 1ull, dd = (deadmask >> r) & 1ull;
        const unsigned long long al = __ballot(in && !dd);
        const int p = first + r * NT + (int)threadIdx.x;
        if (in && dd && p < new_n) {
          const int a = s_cnt[r * NW + wave] + __popcll(al & lt);
          const int src = TMP[(p - first) - a];
          const float mx1 = X1[src], mx2 = X2[src], msc = SC[src], mar = AR[src];
          const int mind = IND[src];
          X1[p] = mx1; X2[p] = mx2; SC[p] = msc; AR[p] = mar; IND[p] = mind;
        }
      }
      nsegs = new_n;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out_cnt[cls] = i;
}

// ------------------------------------------------------------------------------------ soft NMS, register-resident (round 5)
// softnms_rows_kernel keeps the reference's arrays in global memory: a pick is four dependent L2 round trips (argmax, swap,
// decay, compaction), ~22 us at n = 30 000 whatever the arithmetic.  Here a candidate never moves: it lives at a HOME (thread,
// slot) with its score and its two ends in REGISTERS for the whole kernel, and what the reference's array positions are still
// needed for -- the tie order of the first-maximum pick (nms_cpu.cpp:91-101: strict '<' keeps the lowest position) -- is kept as
// two 16-bit maps in LDS, position -> home and home -> position, updated exactly as the reference moves its elements:
//   pick:      the winner goes to position i, the element that sat there to the winner's old position (:103-121);
//   removals:  swap-with-last in ascending position order (:146-154) == the k-th hole gets the k-th survivor from the end;
//              holes and tail survivors are ranked through a position bitmap + per-word prefix popcounts (one wave).
// Homes are dealt in ascending-start order (nms_rank_kernel<true>: wave w owns a contiguous stretch of the time axis), so the
// fp64 exp of the Gaussian decay runs only in the few (wave, slot) pairs that overlap the pick; everything else is one
// interval test per candidate.  Same IoU / decay / threshold expressions and the same fp64 exp as the other kernels: indices
// and decayed scores bit for bit.  One 512-thread workgroup per class, n <= 60 * 512 per class, Gaussian decay (otherwise: rows kernel).
// 8 waves x 60 slots (two waves per SIMD, 256 registers each: 180 hold the candidates).  The slot loops are unrolled (register
// arrays), so their bodies are kept minimal: the first build was 81 KB of code -- more than the 64 KB instruction cache -- and
// spent ~7 us per pick fetching instructions whatever n.  Gaussian decay only (method 2, what batched_nms uses); the deaths of a
// pick are handled by a rolled loop after the decay.
constexpr int NTR = 512, NWR = NTR / 64;
constexpr int RSLOT = 60;
constexpr int REG_MAXN = RSLOT * NTR;
constexpr int FILL_CAP = 16000;                     // tail survivors of one pick kept in LDS (more: the global scratch array)

// workgroup barrier for LDS traffic only: the kernel's global accesses (det rows out, nothing read back) are not waited for --
// __syncthreads() drains them, and a pick's index store sat behind a global round trip on every pick's critical path
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// (score, position) argmax over a wavefront without the LDS crossbar: four DPP steps inside each row of 16 lanes (both registers
// moved alike), then the four row results through v_readlane.  Order: higher score first, ties to the LOWER position; position
// 0x7fffffff = no candidate.
__device__ __forceinline__ void amax_take(float& v, int& p, float ov, int op) {
  if (op != 0x7fffffff && (p == 0x7fffffff || ov > v || (ov == v && op < p))) { v = ov; p = op; }
}
template <int CTRL>
__device__ __forceinline__ void amax_dpp(float& v, int& p) {
  const float ov = vilco_dpp<CTRL>(v);
  const int op = __builtin_amdgcn_update_dpp(0, p, CTRL, 0xF, 0xF, false);
  amax_take(v, p, ov, op);
}
__device__ __forceinline__ void wave_argmax(float& v, int& p) {
  amax_dpp<0xB1>(v, p);
  amax_dpp<0x4E>(v, p);
  amax_dpp<0x141>(v, p);
  amax_dpp<0x140>(v, p);
  float rv = vilco_lane(v, 0);
  int rp = __builtin_amdgcn_readlane(p, 0);
  amax_take(rv, rp, vilco_lane(v, 16), __builtin_amdgcn_readlane(p, 16));
  amax_take(rv, rp, vilco_lane(v, 32), __builtin_amdgcn_readlane(p, 32));
  amax_take(rv, rp, vilco_lane(v, 48), __builtin_amdgcn_readlane(p, 48));
  v = rv; p = rp;
}

struct RegLds {
  unsigned short pos_of[REG_MAXN];                  // home -> array position
  unsigned short cand_at[REG_MAXN];                 // array position -> home
  unsigned bits[REG_MAXN / 32];                     // bit p: the element at position p is alive
  unsigned pref[REG_MAXN / 32 + 1];                 // alive positions in [first, 32 w)
  unsigned short fill[FILL_CAP];
  unsigned long long exptab[32];                    // kExp2fTab: a table look-up per exp is an LDS read, not a global round trip
  float s_val[NWR];
  int s_pos[NWR];
  float s_pick[4];
  int s_flag, s_nalive;
};

#ifdef VILCO_LAB_NMS   // tools/lab only: cycle stamps of picks 100..131 of class 0, wave 0 (never compiled into the product)
__device__ unsigned long long vilco_lab_nms_stamps[32 * 8];
#define NSTAMP(j)                                                                                   \
  do {                                                                                              \
    if (blockIdx.x == 0 && wave == 0 && i >= 100 && i < 132) {                                      \
      const unsigned long long c_ = __builtin_amdgcn_s_memtime();                                   \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
      if (lane == 0) vilco_lab_nms_stamps[(i - 100) * 8 + (j)] = c_;                                \
    }                                                                                               \
  } while (0)
#else
#define NSTAMP(j) do {} while (0)
#endif
__global__ __launch_bounds__(NTR) void softnms_reg_kernel(const long* __restrict__ seg_off, float thr, float sigma, float min_score,
                                                         int method, long max_num, float* __restrict__ dets,
                                                         long* __restrict__ out_idx, long* __restrict__ out_cnt, void* ws_raw,
                                                         long n_total, int n_hi) {
  extern __shared__ __attribute__((aligned(16))) unsigned char reg_lds_raw[];
  RegLds& L = *reinterpret_cast<RegLds*>(reg_lds_raw);
  const int cls = blockIdx.x;
  const long off = seg_off[cls];
  const int n = (int)(seg_off[cls + 1] - off);
  if (n <= 0) {
    if (threadIdx.x == 0) out_cnt[cls] = 0;
    return;
  }
  if (n > n_hi) return;                             // (a class beyond the register file: softnms_rows_kernel / softnms_kernel own it)
  Ws w = carve(ws_raw, n_total, off);               // x1 / x2 / sc / ind in ascending-start order (nms_rank_kernel<true>)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rmax = (n + NTR - 1) / NTR;             // slots in use (uniform)
  // Homes: block q of 64 consecutive starts sits on the 64 lanes of slot q / 8 of wave q % 8 -- a (wave, slot) group covers
  // one stretch of the time axis (a pick overlaps few groups), and the groups a pick does overlap are spread over all waves.
  const int hbase = wave * 64 + lane;               // home of slot r: hbase + r * NTR
  const float NINF = -INFINITY;
  float sc[RSLOT], x1[RSLOT], x2[RSLOT];            // sc = -inf: no candidate here (never was, picked, or dead)
  float gmn = INFINITY, gmx = NINF;                 // lane r: the stretch [min start, max end] of this wave's slot r
#pragma unroll
  for (int r = 0; r < RSLOT; ++r) {
    sc[r] = NINF; x1[r] = 0.f; x2[r] = 0.f;
    if (r < rmax) {
      const int h = hbase + r * NTR;
      if (h < n) {
        x1[r] = w.x1[h]; x2[r] = w.x2[h]; sc[r] = w.sc[h];
        const int orig = w.ind[h];
        L.pos_of[h] = (unsigned short)orig;
        L.cand_at[orig] = (unsigned short)h;
      }
      const float lo = -wave_max(h < n ? -x1[r] : NINF), hi = wave_max(h < n ? x2[r] : NINF);
      if (lane == r) { gmn = lo; gmx = hi; }
    }
  }
  for (int k = tid; k < (n + 31) / 32; k += NTR) L.bits[k] = (k * 32 + 32 <= n) ? 0xffffffffu : ((1u << (n & 31)) - 1u);
  if (tid == 0) L.s_flag = 0;
  if (tid < 32) L.exptab[tid] = kExp2fTab[tid];
  lds_barrier();

  int nsegs = n, i = 0;
  for (; i < nsegs; ++i) {
    if (max_num > 0 && i >= max_num) break;
    NSTAMP(0);
    // ---- (a1) the maximum SCORE: one v_max per candidate, DPP wave reduction, 8-entry fold
    float lm = sc[0];
#pragma unroll
    for (int r = 1; r < RSLOT; ++r) lm = fmaxf(lm, sc[r]);
    lm = wave_max(lm);
    if (lane == 0) L.s_val[wave] = lm;
    lds_barrier();                                                        // #1
    float gv = L.s_val[0];
#pragma unroll
    for (int q = 1; q < NWR; ++q) gv = fmaxf(gv, L.s_val[q]);
    NSTAMP(1);
    // ---- (a2) who holds it: ties by the lowest array position (nms_cpu.cpp:91-101); a match is rare code with static r
    int myp = 0x7fffffff, mine = -1;
    float bx1 = 0.f, bx2 = 0.f;
#pragma unroll
    for (int r = 0; r < RSLOT; ++r) {
      const bool eq = sc[r] == gv;
      if (__builtin_expect(__ballot(eq) != 0ull, 0)) {
        if (eq) {
          const int p = (int)L.pos_of[hbase + r * NTR];
          if (p < myp) { myp = p; mine = r; bx1 = x1[r]; bx2 = x2[r]; }
        }
      }
    }
    // lowest matching position in this wave: DPP max of the negated position (positions < 2^24 are exact in fp32; no match: 2^31)
    const int wp = (int)(-wave_max(-(float)myp));
    if (lane == 0) L.s_pos[wave] = wp;
    lds_barrier();                                                        // #2
    int rp = L.s_pos[0];
#pragma unroll
    for (int q = 1; q < NWR; ++q) rp = min(rp, L.s_pos[q]);
    NSTAMP(2);
    // ---- (b) the owner emits det row i and the two elements trade positions; its slot leaves the running
    const bool owner = mine >= 0 && myp == rp;
    if (owner) {
      const int hw = hbase + mine * NTR;
      float* d = dets + (off + i) * 3;
      d[0] = bx1; d[1] = bx2; d[2] = gv;
      out_idx[off + i] = (long)w.ind[hw];            // (a global round trip of the owner's wave only: the barriers do not wait for it)
      L.s_pick[0] = bx1; L.s_pick[1] = bx2; L.s_pick[2] = bx2 - bx1 + 1e-6f;
      if (rp != i) {
        const unsigned short ci = L.cand_at[i];
        L.cand_at[rp] = ci;
        L.pos_of[ci] = (unsigned short)rp;
      }
      L.cand_at[i] = (unsigned short)hw;
      L.pos_of[hw] = (unsigned short)i;
    }
    const int kill = owner ? mine : -1;
#pragma unroll
    for (int r = 0; r < RSLOT; ++r) sc[r] = kill == r ? NINF : sc[r];
    lds_barrier();                                                        // #3
    NSTAMP(3);
    // ---- (c) decay (nms_cpu.cpp:124-143) in the (wave, slot) groups whose stretch meets the pick; the first pick visits
    // everybody (an initial score below min_score dies there, overlap or not)
    const float ix1 = L.s_pick[0], ix2 = L.s_pick[1], ia = L.s_pick[2];
    const int first = i + 1;
    unsigned long long died = 0;
    unsigned long long ovm = __ballot(lane < rmax && ix2 > gmn && ix1 < gmx);
    if (i == 0) ovm = rmax >= 64 ? ~0ull : ((1ull << rmax) - 1ull);
#pragma unroll
    for (int g8 = 0; g8 < (RSLOT + 7) / 8; ++g8) {
      if ((ovm >> (8 * g8)) & 0xffull) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = 8 * g8 + u;
          if (r < RSLOT && ((ovm >> r) & 1ull)) {
            if (sc[r] != NINF) {
              const float xx1 = fmaxf(ix1, x1[r]);
              const float xx2 = fminf(ix2, x2[r]);
              const float inter = fmaxf(0.f, xx2 - xx1);
              if (inter > 0.f) {                                            // (IoU 0: the weight is exp(-0) = 1 exactly)
                const float ja = x2[r] - x1[r] + 1e-6f;
                const float ovr = inter / (ia + ja - inter);
                sc[r] = sc[r] * expf_libm_tab(-(ovr * ovr) / sigma, L.exptab);
              }
              if (sc[r] < min_score) {
                sc[r] = NINF;
                died |= 1ull << r;
                const int pd = L.pos_of[hbase + r * NTR];
                atomicAnd(&L.bits[pd >> 5], ~(1u << (pd & 31)));
                L.s_flag = 1;
              }
            }
          }
        }
      }
    }
    lds_barrier();                                                        // #4
    NSTAMP(5);
    if (L.s_flag) {                                                         // (uniform)
      // ---- (d) swap-with-last removals (nms_cpu.cpp:146-154): ranks through the bitmap
      const int span = nsegs - first;
      const int w0 = first >> 5, w1 = (nsegs - 1) >> 5;                     // span > 0 here (somebody died)
      if (wave == 0) {
        const int nw = w1 - w0 + 1, per = (nw + 63) / 64;
        const int e0 = w0 + lane * per, e1 = min(w1 + 1, e0 + per);
        int sum = 0;
        for (int e = e0; e < e1; ++e) {
          unsigned b = L.bits[e];
          if (e == w0) b &= ~((1u << (first & 31)) - 1u);
          if (e == w1 && (nsegs & 31)) b &= (1u << (nsegs & 31)) - 1u;
          sum += __popc(b);
        }
        int inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(inc, o, 64);
          if (lane >= o) inc += t;
        }
        int run = inc - sum;
        for (int e = e0; e < e1; ++e) {
          unsigned b = L.bits[e];
          if (e == w0) b &= ~((1u << (first & 31)) - 1u);
          if (e == w1 && (nsegs & 31)) b &= (1u << (nsegs & 31)) - 1u;
          L.pref[e] = (unsigned)run;
          run += __popc(b);
        }
        if (lane == 63) L.s_nalive = inc;
      }
      lds_barrier();                                                      // #4b
      // (reset only now: every wave has read the flag -- it read it before it arrived at #4b -- and the next pick's deaths
      // are three barriers away.  Resetting inside the block above raced with a late wave's read of it: ADVICE r5)
      if (tid == 0) L.s_flag = 0;
      const int n_alive = L.s_nalive;
      const int new_n = first + n_alive;
      // alive positions in [first, p)
      auto alive_before = [&](int p) {
        const int e = p >> 5;
        unsigned b = L.bits[e] & ((1u << (p & 31)) - 1u);
        if (e == w0) b &= ~((1u << (first & 31)) - 1u);
        return (int)L.pref[e] + __popc(b);
      };
      const int tail = nsegs - new_n;                                       // == the number of deaths of this pick
      unsigned short* fill = tail <= FILL_CAP ? L.fill : nullptr;
      int* gfill = w.tmp;
      if (n_alive != span) {
        for (int p = new_n + tid; p < nsegs; p += NTR) {
          if ((L.bits[p >> 5] >> (p & 31)) & 1u) {
            const int t = n_alive - alive_before(p) - 1;                    // rank from the end among the survivors
            if (fill) fill[t] = L.cand_at[p]; else gfill[t] = (int)L.cand_at[p];
          }
        }
        if (fill) lds_barrier(); else __syncthreads();                    // #5 (the global scratch list needs the full barrier)
        unsigned long long dd = died;
        while (dd) {
          const int r = __builtin_ctzll(dd);
          dd &= dd - 1;
          const int pd = L.pos_of[hbase + r * NTR];
          if (pd < new_n) {
            const int hidx = (pd - first) - alive_before(pd);               // rank among the holes, ascending
            const unsigned short c = fill ? fill[hidx] : (unsigned short)gfill[hidx];
            L.cand_at[pd] = c;
            L.pos_of[c] = (unsigned short)pd;
          }
        }
        lds_barrier();                                                    // #6 (every rank is taken before the bitmap changes)
        dd = died;
        while (dd) {
          const int r = __builtin_ctzll(dd);
          dd &= dd - 1;
          const int pd = L.pos_of[hbase + r * NTR];
          if (pd < new_n) atomicOr(&L.bits[pd >> 5], 1u << (pd & 31));
        }
      }
      NSTAMP(6);
      nsegs = new_n;            // (no barrier: the bitmap is next written after two more barriers and read after three)
    }
  }
  if (threadIdx.x == 0) out_cnt[cls] = i;
}

}  // namespace

#ifdef VILCO_LAB_NMS
extern "C" int vilco_lab_nms_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vilco_lab_nms_stamps), sizeof(unsigned long long) * 32 * 8);
}
#endif

// Which soft-NMS kernel takes a class is decided PER CLASS on the device (the host knows the total only): every launched kernel
// covers a range of class sizes (n_lo, n_hi] and its workgroups leave at once for classes outside it.
//   kind 0  softnms_reg_kernel   Gaussian decay (method 2), n <= REG_MAXN
//   kind 1  softnms_rows_kernel  any method, n <= SROWS * NT
//   kind 2  softnms_kernel       any method, any n
// vilco_nms_set_kernel(-1) = automatic (the fastest kernel that can take the class); k >= 0 = nothing faster than kind k.
static int g_soft_kind = [] {
  const char* l = getenv("VILCO_SOFTNMS_LEGACY");
  if (l && l[0] == '1') return 2;
  const char* r = getenv("VILCO_SOFTNMS_REG");
  if (r && atoi(r) == 0) return 1;
  return -1;
}();
static int g_soft_last = 0;

extern "C" int vilco_nms_set_kernel(int32_t kind) {
  const int prev = g_soft_kind;
  if (kind >= -1 && kind <= 2) g_soft_kind = kind;
  return prev;
}

extern "C" int vilco_nms_last_kernels(void) { return g_soft_last; }

extern "C" size_t vilco_nms_workspace(int64_t n_total, int32_t nseg) {
  // 7 words per candidate (x1, x2, score, area, index, keep list / scratch, dead flag) + one counter per class
  return (size_t)(n_total > 0 ? n_total : 1) * 7 * sizeof(float) + 256 + (size_t)(nseg > 0 ? nseg : 1) * sizeof(long);
}

extern "C" int vilco_nms_1d(const float* segs, const float* scores, const int64_t* seg_off,
                            int32_t nseg, int64_t n_total, float iou_threshold, int64_t* out_idx,
                            int64_t* out_cnt, void* workspace, size_t workspace_bytes, void* stream) {
  if (nseg < 0 || !seg_off || !out_cnt) return VILCO_ERR_BADARG;
  if (nseg == 0) return VILCO_OK;
  if (!segs || !scores || !out_idx || !workspace) return VILCO_ERR_BADARG;
  if (n_total < 0 || workspace_bytes < vilco_nms_workspace(n_total, nseg)) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long* so = reinterpret_cast<const long*>(seg_off);
  long* oi = reinterpret_cast<long*>(out_idx);
  long* oc = reinterpret_cast<long*>(out_cnt);
  const size_t per = (size_t)(n_total > 0 ? n_total : 1) * 7 * sizeof(float);
  long* kbase = reinterpret_cast<long*>((reinterpret_cast<uintptr_t>(workspace) + per + 255) / 256 * 256);
  const int rblocks = (int)((n_total + 255) / 256);
  hipLaunchKernelGGL(nms_rank_kernel<false>, dim3(rblocks > 0 ? rblocks : 1), dim3(256), 0, s, segs, scores, so, nseg, oc, kbase,
                     workspace, (long)n_total);
  // a class cannot be longer than n_total: ceil(n_total / CH) chunk rounds cover every class (the workgroups of a class
  // that ended earlier return at once)
  const int chunks = (int)((n_total + CH - 1) / CH);
  for (int c = 0; c < chunks; ++c) {
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(nseg), dim3(NT), 0, s, so, iou_threshold, c, oi, oc, kbase, workspace, (long)n_total);
    const long rest = n_total - (long)(c + 1) * CH;
    if (rest > 0)
      hipLaunchKernelGGL(nms_apply_kernel, dim3((unsigned)((rest + 255) / 256), nseg), dim3(256), 0, s, so, iou_threshold, c,
                         oc, kbase, workspace, (long)n_total);
  }
  return vilco_launch_status();
}

extern "C" int vilco_softnms_1d(const float* segs, const float* scores, const int64_t* seg_off,
                                int32_t nseg, int64_t n_total, float iou_threshold, float sigma,
                                float min_score, int32_t method, int64_t max_num, float* dets,
                                int64_t* out_idx,
                                int64_t* out_cnt, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (nseg < 0 || !seg_off || !out_cnt) return VILCO_ERR_BADARG;
  if (method < 0 || method > 2) return VILCO_ERR_BADARG;
  if (nseg == 0) return VILCO_OK;
  if (!segs || !scores || !dets || !out_idx || !workspace) return VILCO_ERR_BADARG;
  if (n_total < 0 || workspace_bytes < vilco_nms_workspace(n_total, nseg)) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long* so = reinterpret_cast<const long*>(seg_off);
  long* oc = reinterpret_cast<long*>(out_cnt);
  long* oi = reinterpret_cast<long*>(out_idx);
  const int first = g_soft_kind < 0 ? 0 : g_soft_kind;
  const long cap = n_total < 0x7fffffffL ? n_total : 0x7fffffffL;      // no class is longer than the total
  int lo = 0, used = 0;                                                   // classes of size <= lo are taken
  if (first <= 0 && method == 2) {
    const size_t per = (size_t)(n_total > 0 ? n_total : 1) * 7 * sizeof(float);
    long* kbase = reinterpret_cast<long*>((reinterpret_cast<uintptr_t>(workspace) + per + 255) / 256 * 256);
    const int rblocks = (int)((n_total + 255) / 256);
    hipLaunchKernelGGL(nms_rank_kernel<true>, dim3(rblocks > 0 ? rblocks : 1), dim3(256), 0, s, segs, scores, so, nseg, oc, kbase,
                       workspace, (long)n_total);
    static const bool once = [] {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&softnms_reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)sizeof(RegLds));
      (void)hipGetLastError();
      return true;
    }();
    (void)once;
    hipLaunchKernelGGL(softnms_reg_kernel, dim3(nseg), dim3(NTR), sizeof(RegLds), s, so, iou_threshold, sigma, min_score, method,
                       (long)max_num, dets, oi, oc, workspace, (long)n_total, (int)REG_MAXN);
    lo = REG_MAXN;
    used |= 1;
  }
  if (first <= 1 && (cap > lo || !used)) {
    hipLaunchKernelGGL(softnms_rows_kernel, dim3(nseg), dim3(NT), 0, s, segs, scores, so, iou_threshold, sigma, min_score, method,
                       (long)max_num, dets, oi, oc, workspace, (long)n_total, lo, SROWS * NT);
    lo = SROWS * NT;
    used |= 2;
  }
  if (cap > lo || !used) {
    hipLaunchKernelGGL(softnms_kernel, dim3(nseg), dim3(NT), 0, s, segs, scores, so, iou_threshold, sigma, min_score, method,
                       (long)max_num, dets, oi, oc, workspace, (long)n_total, lo, 0x7fffffff);
    used |= 4;
  }
  g_soft_last = used;
  return vilco_launch_status();
}
