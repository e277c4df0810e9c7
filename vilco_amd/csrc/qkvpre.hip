// q/k/v pre-projection of MaskedMHCA fused with the block's first LayerNorm (MQ/libs/modeling/blocks.py:561-563 ln1,
// :363-369 query/key/value_conv + *_norm):
//
//     h = LN1(x);   for j in {q, k, v}:   y_j = LN_j( dwconv3_{stride}(h; w_j) * mask )
//
// ONE read of x and three writes (plus h itself when another consumer needs it -- the channel-attention branch of the
// stem blocks, the time adapters), instead of LN1 (r+w), three depthwise convs (r+w each) and three LayerNorms (r+w
// each): 4*C*T read + 3*4*C*T' written per clip against 15 passes.  HBM-bound; the north-star target kernel
// (">= 60 % of HBM bandwidth on the 1-D conv FPN at T = C = 2304").
//
// Forward mapping (r03, qkv_pre_fwd_ring_kernel; the three-pass qkv_pre_fwd_kernel serves runs too short to fill the
// pipeline): one workgroup of 3 + C/256 waves walks a run of consecutive output tokens of one clip.  Three producer
// waves bring the x rows in (loads in flight for 5 barrier intervals), normalise them ONCE and park them in a two-group
// LDS ring; C/256 consumer waves, each owning one 256-channel chunk with its 15 parameters in registers, read the ring,
// convolve, reduce and write.  One workgroup barrier per token, every x row read once, every output row written once:
// 111 us at [8, 2304, 2304] = 6.1 TB/s of algorithmic bytes = 0.77 of the 8 TB/s HBM peak (three-pass kernel: 226 us).
//
// Backward (qkv_pre_bwd_*): the conv outputs are never stored -- they are recomputed from h in registers.
//   rows:    per output token, x_hat_j from (h, stats), LayerNorm backward -> d(conv out)_j written (3 tensors)
//   params:  per channel, d gamma_j / d beta_j / d w_j accumulated over row chunks (no reductions: statistics are saved)
//   dh:      dh = sum_j dwconv^T(dc_j) + dh from the other consumers, one pass
// followed by the ordinary LayerNorm backward of LN1 (norm.hip).
#include <cstdlib>
#include "common.h"

void vilco_reduce_rows(const float* ws, float* out0, float* out1, int nrows, int ncols, int split, hipStream_t s);

namespace {

constexpr int QT = 256;

struct QkvArgs {
  const float* x;            // [B][T][C]
  const float* g1; const float* b1;                 // LN1
  const float* w[3];         // [C][3] depthwise taps of q, k, v
  const float* gam[3]; const float* bet[3];         // LN_j
  const int* len;            // [B] valid input length
  float* h;                  // [B][T][C] or null
  float* y[3];               // [B][Tout][C]
  float* mean1; float* rstd1;                       // [B*T]
  float* mean[3]; float* rstd[3];                   // [B*Tout]
  float* amax[3];            // optional: one partial max |y_j| per WAVE (the consumers' operand packs fold them)
  int B, T, Tout, C, stride, seg;
  float eps1, eps;
};

// block-wide sums of N values (every thread gets the totals); red: [NW][N] floats of LDS, NW = waves per block
template <int N, int NW>
__device__ __forceinline__ void block_sums(float (&v)[N], float* red) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();                      // previous readers of `red` are done
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < N; ++i) red[(threadIdx.x >> 6) * N + i] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += red[w * N + i];
    v[i] = t;
  }
}

// Forward, third version.  ONE WAVE owns TB consecutive output tokens and all C channels of them; a lane owns 4
// consecutive channels of every 256-channel chunk (16-byte accesses, 1 KB per wave instruction); no block barrier
// anywhere, every reduction is a wavefront reduction.  Three passes over the S*TB + 2 input rows:
//   0: LayerNorm statistics of the input rows            (x from HBM)
//   1: h = LN1(x), the three convs, THEIR statistics     (x again: L2 / Infinity Cache; h written when asked for)
//   2: h and the convs once more, normalise, write q,k,v (x again)
// The conv outputs are recomputed instead of kept (9 FMAs per element against 3 x C floats of registers per token).
// Statistics are shifted one-pass sums -- sum(v - k), sum((v - k)^2) with k = the row's first element, so the
// subtraction in E[d^2] - E[d]^2 cancels nothing unless |mean - k| >> std, which a member of the row is not.
// (Version 1 walked tokens sequentially with four block barriers each: 1.4 TB/s.  Version 2 batched the rows of a block
// through registers: 1.3 TB/s at C = 2304 with spills, 2.7 TB/s at C = 1024.)
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float hsum(const float4& v) { return (v.x + v.y) + (v.z + v.w); }

template <int TB, int S>
__global__ __launch_bounds__(256) void qkv_pre_fwd_kernel(QkvArgs a) {
  constexpr int NR = S * TB + 2;
  const int lane = threadIdx.x & 63;
  const int groups = (a.Tout + TB - 1) / TB;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long)a.B * groups) return;                 // whole waves only: no barrier below
  const int b = (int)(wid / groups), t0 = (int)(wid % groups) * TB;
  const int C = a.C, T = a.T, NG = C >> 8;
  const float invC = 1.f / (float)C;
  const int len = a.len[b];
  const int r0 = S * t0 - 1;
  const float* xb = a.x + (long)b * T * C + lane * 4;

  // Rows outside [0, T) are the conv's zero padding.  Their loads are issued anyway, from the clamped row, and the
  // result is discarded: a guarded load is a branch with its own s_waitcnt, which serialises the NR row loads of a chunk
  // (measured: NR x NG dependent round trips per pass); unconditional loads are all in flight together, and the next
  // chunk's rows are requested before this chunk is consumed.
  bool ok[NR];
  const float* xr[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int t = r0 + r;
    ok[r] = t >= 0 && t < T;
    xr[r] = xb + (long)(t < 0 ? 0 : (t >= T ? T - 1 : t)) * C;
  }
  auto load_x = [&](int g, float4 (&v)[NR]) {
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = ldg4(xr[r] + g * 256);
  };

  // ---- pass 0: statistics of the input rows
  float mu[NR], rs[NR];
  {
    float k[NR], s1[NR], s2[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      k[r] = vilco_lane(*(xr[r]), 0);         // the row's first element (lane 0's first channel)
      s1[r] = 0.f; s2[r] = 0.f;
    }
    float4 v[NR], vn[NR];
    load_x(0, v);
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) load_x(g + 1, vn);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float d0 = v[r].x - k[r], d1 = v[r].y - k[r], d2 = v[r].z - k[r], d3 = v[r].w - k[r];
        s1[r] += (d0 + d1) + (d2 + d3);
        s2[r] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) v[r] = vn[r];
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const float m = wave_sum(s1[r]) * invC, q = wave_sum(s2[r]) * invC;
      mu[r] = k[r] + m;
      rs[r] = 1.0f / sqrtf(fmaxf(q - m * m, 0.f) + a.eps1);
      const int t = r0 + r;
      if (lane == 0 && a.mean1 && r >= 1 && r <= S * TB && t < T) { a.mean1[(long)b * T + t] = mu[r]; a.rstd1[(long)b * T + t] = rs[r]; }
    }
  }

  // h rows of chunk g out of its x rows (zero outside [0, T): the conv's padding)
  auto make_h = [&](int g, const float4 (&v)[NR], float4 (&h)[NR]) {
    const float4 g1 = a.g1 ? ldg4(a.g1 + g * 256 + lane * 4) : f4(1.f);
    const float4 b1 = a.b1 ? ldg4(a.b1 + g * 256 + lane * 4) : f4(0.f);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      h[r].x = ok[r] ? (v[r].x - mu[r]) * rs[r] * g1.x + b1.x : 0.f; h[r].y = ok[r] ? (v[r].y - mu[r]) * rs[r] * g1.y + b1.y : 0.f;
      h[r].z = ok[r] ? (v[r].z - mu[r]) * rs[r] * g1.z + b1.z : 0.f; h[r].w = ok[r] ? (v[r].w - mu[r]) * rs[r] * g1.w + b1.w : 0.f;
    }
  };
  // conv j of chunk g for the TB tokens: w[c][3] of 4 consecutive channels = 12 floats
  auto conv = [&](int j, int g, const float4 (&h)[NR], float4 (&c)[TB]) {
    const float* wp = a.w[j] + (g * 256 + lane * 4) * 3;
    const float4 wa = ldg4(wp), wb = ldg4(wp + 4), wc = ldg4(wp + 8);       // x: a.x a.y a.z | y: a.w b.x b.y | z: b.z b.w c.x | w: c.y c.z c.w
#pragma unroll
    for (int i = 0; i < TB; ++i) {
      const int t = t0 + i;
      if (!(t < a.Tout && S * t < len)) { c[i] = f4(0.f); continue; }
      const float4 &p = h[S * i], &q = h[S * i + 1], &n = h[S * i + 2];
      c[i].x = wa.x * p.x + wa.y * q.x + wa.z * n.x;
      c[i].y = wa.w * p.y + wb.x * q.y + wb.y * n.y;
      c[i].z = wb.z * p.z + wb.w * q.z + wc.x * n.z;
      c[i].w = wc.y * p.w + wc.z * q.w + wc.w * n.w;
    }
  };

  // ---- pass 1: h (stored when asked for), conv statistics
  float cm[3][TB], cr[3][TB];
  {
    float k[3][TB], s1[3][TB], s2[3][TB];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < TB; ++i) { k[j][i] = 0.f; s1[j][i] = 0.f; s2[j][i] = 0.f; }
    float4 xv[NR], xn[NR];
    load_x(0, xv);
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) load_x(g + 1, xn);
      float4 h[NR];
      make_h(g, xv, h);
#pragma unroll
      for (int r = 0; r < NR; ++r) xv[r] = xn[r];
      if (a.h) {
#pragma unroll
        for (int r = 1; r <= S * TB; ++r) {
          const int t = r0 + r;
          if (t < T) *reinterpret_cast<float4*>(a.h + ((long)b * T + t) * C + g * 256 + lane * 4) = h[r];
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float4 c[TB];
        conv(j, g, h, c);
#pragma unroll
        for (int i = 0; i < TB; ++i) {
          if (g == 0) k[j][i] = vilco_lane(c[i].x, 0);
          const float d0 = c[i].x - k[j][i], d1 = c[i].y - k[j][i], d2 = c[i].z - k[j][i], d3 = c[i].w - k[j][i];
          s1[j][i] += (d0 + d1) + (d2 + d3);
          s2[j][i] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        const float m = wave_sum(s1[j][i]) * invC, q = wave_sum(s2[j][i]) * invC;
        cm[j][i] = k[j][i] + m;
        cr[j][i] = 1.0f / sqrtf(fmaxf(q - m * m, 0.f) + a.eps);
        const int t = t0 + i;
        if (lane == 0 && a.mean[j] && t < a.Tout) { a.mean[j][(long)b * a.Tout + t] = cm[j][i]; a.rstd[j][(long)b * a.Tout + t] = cr[j][i]; }
      }
  }

  // ---- pass 2: normalise and write
  float omax[3] = {0.f, 0.f, 0.f};
  float4 xv[NR], xn[NR];
  load_x(0, xv);
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) load_x(g + 1, xn);
    float4 h[NR];
    make_h(g, xv, h);
#pragma unroll
    for (int r = 0; r < NR; ++r) xv[r] = xn[r];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float4 c[TB];
      conv(j, g, h, c);
      const float4 gm = a.gam[j] ? ldg4(a.gam[j] + g * 256 + lane * 4) : f4(1.f);
      const float4 bt = a.bet[j] ? ldg4(a.bet[j] + g * 256 + lane * 4) : f4(0.f);
#pragma unroll
      for (int i = 0; i < TB; ++i) {
        const int t = t0 + i;
        if (t >= a.Tout) continue;
        float4 o;
        o.x = (c[i].x - cm[j][i]) * cr[j][i] * gm.x + bt.x; o.y = (c[i].y - cm[j][i]) * cr[j][i] * gm.y + bt.y;
        o.z = (c[i].z - cm[j][i]) * cr[j][i] * gm.z + bt.z; o.w = (c[i].w - cm[j][i]) * cr[j][i] * gm.w + bt.w;
        omax[j] = fmaxf(fmaxf(omax[j], fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        *reinterpret_cast<float4*>(a.y[j] + ((long)b * a.Tout + t) * C + g * 256 + lane * 4) = o;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j)
    if (a.amax[j]) {
      const float m = wave_max(omax[j]);
      if (lane == 0) a.amax[j][wid] = m;
    }
}

// Forward, fourth version (r03): a ROW RING in LDS, producer / consumer waves, every wave channel-split.
// One workgroup walks a run of `seg` consecutive output tokens of one clip; 3 + C/256 waves:
//   * A-waves (3): each owns every third 256-channel chunk of every input row.  Rows travel global -> registers KS row
//     groups ahead (a group = the S new rows one output token needs; nothing waits on a load that was not issued ~KS
//     barrier intervals earlier), then: LayerNorm-1 statistics as per-wave shifted sums, combined over the three waves
//     with the exact pairwise mean / M2 update -> h = LN1(x) written ONCE into a two-group LDS ring (and to HBM when
//     another consumer wants h).
//   * B-waves (C/256): each owns ONE 256-channel chunk of every output token, its 15 per-channel parameters live in
//     registers for the whole run.  Per interval: read the new h rows of its chunk from the ring (the previous two stay in
//     registers), the three convs, six partial sums -> LDS; one interval later the combined statistics normalise the
//     token and q, k, v leave as three 1 KB stores per wave.
// ONE workgroup barrier per token; x is read once (plus two halo rows per run), nothing is recomputed, no row is
// re-fetched.  Pipeline (interval i): A finalises group i, B convolves token i - D and writes token i - D - 1.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void stg4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCH>
__device__ __forceinline__ float sum_parts(const float* p) {      // p[0 .. NCH): fixed order, the same in every wave
  const float4 a = *reinterpret_cast<const float4*>(p);
  float s = NCH >= 4 ? (a.x + a.y) + (a.z + a.w) : NCH == 3 ? (a.x + a.y) + a.z : NCH == 2 ? a.x + a.y : a.x;
  if (NCH > 4) {
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    s += NCH >= 8 ? (b.x + b.y) + (b.z + b.w) : NCH == 7 ? (b.x + b.y) + b.z : NCH == 6 ? b.x + b.y : b.x;
  }
  if (NCH > 8) s += p[8];
  return s;
}

// Joint wavefront sums: v_permlane32_swap / v_permlane16_swap exchange half the lanes of TWO registers in one instruction,
// so each step halves the number of live registers instead of spending a full butterfly per value (18 instructions for
// six sums instead of 66).  Results land in 16-lane rows: wave_sum4 -> row 0: sum v0, row 1: sum v2, row 2: sum v1, row 3:
// sum v3; wave_sum2 -> rows 0-1: sum v0, rows 2-3: sum v1.  Fixed order, bitwise reproducible.
__device__ __forceinline__ void swap32(float& x, float& y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& x, float& y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]); y = __uint_as_float(r[1]);
}
__device__ __forceinline__ float row_sum(float s) {
  s += vilco_dpp<0xB1>(s); s += vilco_dpp<0x4E>(s); s += vilco_dpp<0x141>(s); s += vilco_dpp<0x140>(s);
  return s;
}
__device__ __forceinline__ float wave_sum4(float v0, float v1, float v2, float v3) {
  swap32(v0, v1); float s01 = v0 + v1;
  swap32(v2, v3); float s23 = v2 + v3;
  swap16(s01, s23);
  return row_sum(s01 + s23);
}
__device__ __forceinline__ float wave_sum2(float v0, float v1) {
  swap32(v0, v1);
  float s = v0 + v1, t = s;
  swap16(s, t);
  return row_sum(s + t);
}

constexpr int RING_NA = 3;
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 lo2(const f32x4& v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(const f32x4& v) { return __builtin_shufflevector(v, v, 2, 3); }
__device__ __forceinline__ f32x4 ld4s(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }

template <int NCH, int S>
__global__ __launch_bounds__((RING_NA + NCH) * 64) void qkv_pre_fwd_ring_kernel(QkvArgs a, int npart) {
  // KS: register slots of row groups per A-wave = unroll of both loops (the B-waves' row window rotates with period 3 at
  // stride 1, their conv rows with period 2; KS = 6 / 4 is a multiple of both)
  constexpr int NA = RING_NA, ACH = (NCH + NA - 1) / NA, KS = (S == 1) ? 6 : 4, D = (S == 1) ? 3 : 2, C = NCH * 256;
  constexpr bool ALLV = NCH % NA == 0;                         // every A-wave owns ACH whole chunks
  __shared__ __align__(16) float ring[2][S][C];
  __shared__ __align__(16) float bfin[2][3][2];                // (mean, rstd) of a token's three conv rows, combined by the A-waves
  __shared__ __align__(16) float apart[2][S][NA][4];
  __shared__ __align__(16) float bpart[2][6][12];
  __shared__ float wmax[3][12];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nseg = (a.Tout + a.seg - 1) / a.seg;
  const int b = blockIdx.x / nseg, t0 = (blockIdx.x % nseg) * a.seg;
  const int t1 = t0 + a.seg < a.Tout ? t0 + a.seg : a.Tout, nsteps = t1 - t0;
  const int NI = nsteps + D + 2, NQ = nsteps + (S == 1 ? 2 : 1);        // barrier intervals; row groups the run needs
  const int NIp = (NI + KS - 1) / KS * KS;                              // whole rounds of the KS slots: the surplus intervals touch nothing that is used
  const int T = a.T, rb = S * t0 - 1;                                    // group q = rows rb + q S + u, u < S
  const float invC = 1.f / (float)C;

  if (wave < NA) {
    // ------------------------------------------------------------------------------------------------ A: x -> h ring
    // The row loads are inline asm (uniform row base in SGPRs + the lane's constant byte offset) with hand-placed
    // s_waitcnt vmcnt(N): the compiler's counter model merges control-flow joins conservatively (a conditional store
    // between issue and use costs one load of prefetch depth each), and the whole point of these waves is loads that stay
    // in flight for KS - 1 barrier intervals.
    const int aw = wave;
    bool cv[ACH]; int co[ACH];
#pragma unroll
    for (int m = 0; m < ACH; ++m) { const int ch = aw + NA * m; cv[m] = ALLV || ch < NCH; co[m] = (cv[m] ? ch : 0) * 256 + lane * 4; }
    f32x4 g1v[ACH], b1v[ACH];
#pragma unroll
    for (int m = 0; m < ACH; ++m) { g1v[m] = a.g1 ? ld4s(a.g1 + co[m]) : splat4(1.f); b1v[m] = a.b1 ? ld4s(a.b1 + co[m]) : splat4(0.f); }
    const float* xb = a.x + (long)b * T * C;
    f32x4 xs[KS][S][ACH];
    auto load = [&](int q, f32x4 (&dst)[S][ACH]) {
      q = q < NQ - 1 ? q : NQ - 1;                            // past the run: the last group again (cache hits, no new bytes)
#pragma unroll
      for (int u = 0; u < S; ++u) {
        int r = rb + q * S + u;
        r = r < 0 ? 0 : (r >= T ? T - 1 : r);                 // padding rows: loaded from the clamped row, discarded
        const float* p = xb + (long)r * C;
#pragma unroll
        for (int m = 0; m < ACH; ++m)
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst[u][m]) : "v"(co[m] * 4), "s"(p) : "memory");
      }
    };
    auto arrived = [&](f32x4 (&v)[S][ACH]) {                  // all but the (KS - 1) younger groups' loads have landed
#pragma unroll
      for (int u = 0; u < S; ++u)
#pragma unroll
        for (int m = 0; m < ACH; ++m) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v[u][m]) : "n"((KS - 1) * S * ACH));
    };
    auto stats = [&](int q, f32x4 (&v)[S][ACH]) {             // shifted sums of this wave's slice of the group's rows
      arrived(v);
      float ks[S], p1[S], p2[S];
#pragma unroll
      for (int u = 0; u < S; ++u) {
        const float k = vilco_lane(v[u][0].x, 0);              // shift: the first element of this wave's slice
        const f32x2 k2 = {k, k};
        f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
        for (int m = 0; m < ACH; ++m)
          if (cv[m]) {
            const f32x2 d0 = lo2(v[u][m]) - k2, d1 = hi2(v[u][m]) - k2;
            s1 += d0 + d1;
            s2 = __builtin_elementwise_fma(d0, d0, s2);
            s2 = __builtin_elementwise_fma(d1, d1, s2);
          }
        ks[u] = k; p1[u] = s1.x + s1.y; p2[u] = s2.x + s2.y;
      }
      if (S == 1) {
        const float r = wave_sum2(p1[0], p2[0]);               // lanes < 32: s1, lanes >= 32: s2
        if (lane == 0) { apart[q & 1][0][aw][0] = ks[0]; apart[q & 1][0][aw][1] = r; }
        if (lane == 32) apart[q & 1][0][aw][2] = r;
      } else {
        const float r = wave_sum4(p1[0], p2[0], p1[S - 1], p2[S - 1]);     // rows: s1 of row 0, s1 of row 1, s2 of row 0, s2 of row 1
        if ((lane & 15) == 0) {
          const int row = lane >> 4;
          apart[q & 1][row & 1][aw][1 + (row >> 1)] = r;
          if (row < 2) apart[q & 1][row][aw][0] = row ? ks[S - 1] : ks[0];
        }
      }
    };
    auto finalize = [&](int q, const f32x4 (&v)[S][ACH]) {    // the three waves' sums -> mean, rstd (pairwise update) -> h
#pragma unroll
      for (int u = 0; u < S; ++u) {
        const int r = rb + q * S + u;
        const bool in = r >= 0 && r < T, own = in && r >= S * t0 && r < S * t1;
        float mw[NA], m2[NA], tot = 0.f;
#pragma unroll
        for (int w = 0; w < NA; ++w) {
          const int nv = (NCH - w + NA - 1) / NA;              // chunks of A-wave w (0 when w >= NCH)
          if (nv > 0) {
            const float4 p = *reinterpret_cast<const float4*>(apart[q & 1][u][w]);
            const float n = 256.f * nv, d = p.y * (1.f / n);
            mw[w] = p.x + d; m2[w] = p.z - p.y * d; tot += n * mw[w];
          }
        }
        const float mu = tot * invC;
        float M2 = 0.f;
#pragma unroll
        for (int w = 0; w < NA; ++w) {
          const int nv = (NCH - w + NA - 1) / NA;
          if (nv > 0) { const float d = mw[w] - mu; M2 += m2[w] + 256.f * nv * d * d; }
        }
        const float rs = __builtin_amdgcn_rsqf(fmaxf(M2 * invC, 0.f) + a.eps1);
        const f32x2 mu2 = {mu, mu}, rs2 = {rs, rs};
#pragma unroll
        for (int m = 0; m < ACH; ++m)
          if (cv[m]) {                                           // rows outside [0, T) (the conv's zero padding) are zeroed by their readers
            const f32x2 hl = __builtin_elementwise_fma((lo2(v[u][m]) - mu2) * rs2, lo2(g1v[m]), lo2(b1v[m]));
            const f32x2 hh = __builtin_elementwise_fma((hi2(v[u][m]) - mu2) * rs2, hi2(g1v[m]), hi2(b1v[m]));
            const f32x4 h = __builtin_shufflevector(hl, hh, 0, 1, 2, 3);
            *reinterpret_cast<f32x4*>(&ring[q & 1][u][co[m]]) = h;
            if (a.h && own) *reinterpret_cast<f32x4*>(a.h + ((long)b * T + r) * C + co[m]) = h;
          }
        if (aw == 0 && lane == 0 && a.mean1 && own) { a.mean1[(long)b * T + r] = mu; a.rstd1[(long)b * T + r] = rs; }
      }
    };
#pragma unroll
    for (int u = 0; u < KS; ++u) load(u, xs[u]);
    stats(0, xs[0]);                                           // (waits for group 0 only: KS - 1 groups stay in flight)
    lds_barrier();
    for (int i0 = 0; i0 < NIp; i0 += KS) {
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        const int i = i0 + u;
        finalize(i, xs[u]);
        load(i + KS, xs[u]);
        {                                                      // the B-waves' partial sums of interval i - 1 -> (mean, rstd) of conv aw
          const float m = sum_parts<NCH>(bpart[(i - 1) & 1][2 * aw]) * invC, q2 = sum_parts<NCH>(bpart[(i - 1) & 1][2 * aw + 1]) * invC;
          const float rs = __builtin_amdgcn_rsqf(fmaxf(q2 - m * m, 0.f) + a.eps);
          const int s = i - 1 - D;
          if (lane == 0) {
            bfin[i & 1][aw][0] = m; bfin[i & 1][aw][1] = rs;
            float* mp = aw == 0 ? a.mean[0] : aw == 1 ? a.mean[1] : a.mean[2];
            float* rp = aw == 0 ? a.rstd[0] : aw == 1 ? a.rstd[1] : a.rstd[2];
            if (mp && s >= 0 && s < nsteps) { mp[(long)b * a.Tout + t0 + s] = m; rp[(long)b * a.Tout + t0 + s] = rs; }
          }
        }
        stats(i + 1, xs[(u + 1) % KS]);
        lds_barrier();
      }
    }
    lds_barrier();                                             // the B-waves' amax hand-over
  } else {
    // ------------------------------------------------------------------------------------------- B: ring -> q, k, v
    const int c = wave - NA, co = c * 256 + lane * 4;
    f32x4 w0[3], w1[3], w2[3], gm[3], bt[3];                   // taps as per-channel vectors
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float* wp = a.w[j] + co * 3;
      const f32x4 wa = ld4s(wp), wb = ld4s(wp + 4), wc = ld4s(wp + 8);   // channel 0: wa.xyz | 1: wa.w wb.xy | 2: wb.zw wc.x | 3: wc.yzw
      w0[j] = f32x4{wa.x, wa.w, wb.z, wc.y}; w1[j] = f32x4{wa.y, wb.x, wb.w, wc.z}; w2[j] = f32x4{wa.z, wb.y, wc.x, wc.w};
      gm[j] = a.gam[j] ? ld4s(a.gam[j] + co) : splat4(1.f);
      bt[j] = a.bet[j] ? ld4s(a.bet[j] + co) : splat4(0.f);
    }
    const int len = a.len[b];
    constexpr int NH = (S == 1) ? 3 : 2;
    f32x4 hw[NH][S];                                           // stride 1: the last three rows; stride 2: the last two groups
    f32x4 cw[2][3];                                            // conv rows of the two tokens in flight
    float om[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NH; ++k)
#pragma unroll
      for (int u = 0; u < S; ++u) hw[k][u] = splat4(0.f);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 3; ++j) cw[k][j] = splat4(0.f);
    float* yb[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) yb[j] = a.y[j] + ((long)b * a.Tout + t0) * C + co;
    lds_barrier();
    for (int i0 = 0; i0 < NIp; i0 += KS) {
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        const int i = i0 + u;
        const int sf = i - D - 2;
        const bool wr = sf >= 0 && sf < nsteps;
        f32x4 o[3];
        if (wr) {                                               // token sf: statistics complete -> normalise
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float2 fin = *reinterpret_cast<const float2*>(bfin[(i - 1) & 1][j]);
            const f32x2 m2 = {fin.x, fin.x}, r2 = {fin.y, fin.y};
            const f32x2 ol = __builtin_elementwise_fma((lo2(cw[u % 2][j]) - m2) * r2, lo2(gm[j]), lo2(bt[j]));
            const f32x2 oh = __builtin_elementwise_fma((hi2(cw[u % 2][j]) - m2) * r2, hi2(gm[j]), hi2(bt[j]));
            om[j] = fmaxf(fmaxf(fabsf(ol.x), fabsf(ol.y)), fmaxf(fmaxf(fabsf(oh.x), fabsf(oh.y)), om[j]));
            o[j] = __builtin_shufflevector(ol, oh, 0, 1, 2, 3);
          }
        }
        // The three 1 KB stores of a wave are spread over the interval (start / after the convs / before the barrier): issued
        // together, the nine B-waves' 27 KB overran the CU's store queue and the waves sat in the issue of the burst
        // instead of computing (stores only: 117 us against 97 us for a bare fill of the same bytes).
        auto put = [&](int j) { if (wr) __builtin_nontemporal_store(o[j], reinterpret_cast<f32x4*>(yb[j] + (long)sf * C)); };
        put(0);
        __builtin_amdgcn_sched_barrier(0);
        // group i - 1 of the ring (interval 0 reads nothing that is used)
#pragma unroll
        for (int r = 0; r < S; ++r) hw[(u + NH - 1) % NH][r] = *reinterpret_cast<const f32x4*>(&ring[(i - 1) & 1][r][co]);
        const int s = i - D;
        if (s >= 0 && s < nsteps) {                             // token s
          f32x4 p0 = S == 1 ? hw[u % 3][0] : hw[u % 2][0], n0 = S == 1 ? hw[(u + 2) % 3][0] : hw[(u + 1) % 2][0];
          const f32x4& p1 = S == 1 ? hw[(u + 1) % 3][0] : hw[u % 2][S - 1];
          if (S * (t0 + s) - 1 < 0) p0 = splat4(0.f);           // the conv's zero padding: rows -1 and T (first / last token of a clip)
          if (S * (t0 + s) + 1 >= T) n0 = splat4(0.f);
          float sm[6];
          if (S * (t0 + s) < len) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const f32x4 cj = w0[j] * p0 + w1[j] * p1 + w2[j] * n0;
              cw[u % 2][j] = cj;
              const f32x2 cl = lo2(cj), ch = hi2(cj);
              const f32x2 s1 = cl + ch, s2 = __builtin_elementwise_fma(cl, cl, ch * ch);
              sm[2 * j] = s1.x + s1.y;
              sm[2 * j + 1] = s2.x + s2.y;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) { cw[u % 2][j] = splat4(0.f); sm[2 * j] = 0.f; sm[2 * j + 1] = 0.f; }
          }
          __builtin_amdgcn_sched_barrier(0);
          put(1);
          __builtin_amdgcn_sched_barrier(0);
          const float r4 = wave_sum4(sm[0], sm[1], sm[2], sm[3]);      // rows: sm0, sm2, sm1, sm3
          const float r2 = wave_sum2(sm[4], sm[5]);                    // rows 0-1: sm4, rows 2-3: sm5
          if ((lane & 15) == 0) {
            const int row = lane >> 4;
            bpart[i & 1][((row & 1) << 1) | (row >> 1)][c] = r4;
            if (!(row & 1)) bpart[i & 1][4 + (row >> 1)][c] = r2;
          }
        } else {
          put(1);
        }
        __builtin_amdgcn_sched_barrier(0);
        put(2);
        lds_barrier();
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float m = wave_max(om[j]);
      if (lane == 0) wmax[j][c] = m;
    }
    lds_barrier();
    float* ap = lane == 0 ? a.amax[0] : lane == 1 ? a.amax[1] : a.amax[2];
    if (c == 0 && lane < 3 && ap) {                             // one partial per workgroup; unused slots of the table: 0
      float m = 0.f;
      for (int k = 0; k < NCH; ++k) m = fmaxf(m, wmax[lane][k]);
      ap[blockIdx.x] = m;
      for (int k = gridDim.x + blockIdx.x; k < npart; k += gridDim.x) ap[k] = 0.f;
    }
  }
}

// Other forward structures built and measured at [8, 2304, 2304] and removed (r02 / r03; DESIGN_LOG.md 3.4):
//   * rows of a 4-token tile normalised once into LDS, one token per wave: 288 us -- every wave re-loads the 17
//     per-channel parameter quads per token and pass;
//   * channel-split tiles (a wave owns 256 channels of 6 tokens, statistics combined across the waves through LDS, every
//     wave doing every stage): 295 us / 227 us -- ~55 wavefront reductions per 72 outputs of a lane, issue-bound; as a
//     persistent sliding window it spilled.  The ring kernel keeps the channel split but separates the stages into
//     producer and consumer waves and reduces six sums jointly (wave_sum4 / wave_sum2).
// What bounded the three-pass kernel is real HBM traffic: rocprofv3 FETCH_SIZE / WRITE_SIZE (profiles/r02_z_targets_pmc_*.json)
// show x fetched 3.6x (the rows a wave re-reads in passes 1 and 2 have left the 4 MB L2 of its XCD).

// ---------------------------------------------------------------------------------------------------- backward, rows
struct QkvBwdArgs {
  const float* h;            // [B][T][C]  (LN1 output, saved or recomputed)
  const float* w[3];
  const float* gam[3];
  const float* dy[3];        // [B][Tout][C]
  const float* mean[3]; const float* rstd[3];
  const int* len;
  float* dc[3];              // [B][Tout][C]  gradient wrt the (masked) conv outputs
  int B, T, Tout, C, stride, seg;
};

template <int CPT, int NT>
__global__ __launch_bounds__(NT) void qkv_pre_bwd_rows_kernel(QkvBwdArgs a) {
  constexpr int QT = NT, NW = NT / 64;
  __shared__ float red[NW * 6];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * a.seg;
  int t1 = t0 + a.seg;
  if (t1 > a.Tout) t1 = a.Tout;
  const int C = a.C, T = a.T, s = a.stride, tid = threadIdx.x;
  const float invC = 1.f / (float)C;
  const int len = a.len[b];
  float w[3][3][CPT], gam[3][CPT];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int c = tid + k * QT;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      w[j][0][k] = a.w[j][c * 3]; w[j][1][k] = a.w[j][c * 3 + 1]; w[j][2][k] = a.w[j][c * 3 + 2];
      gam[j][k] = a.gam[j] ? a.gam[j][c] : 1.f;
    }
  }
  auto load_h = [&](int t, float (&dst)[CPT]) {
    if (t < 0 || t >= T) {
#pragma unroll
      for (int k = 0; k < CPT; ++k) dst[k] = 0.f;
      return;
    }
    const float* hr = a.h + ((long)b * T + t) * C;
#pragma unroll
    for (int k = 0; k < CPT; ++k) dst[k] = hr[tid + k * QT];
  };
  float hm[CPT], h0[CPT], hp[CPT];
  load_h(s * t0 - 1, hm);
  if (s == 1) load_h(t0, h0);
  for (int t = t0; t < t1; ++t) {
    if (s == 1) load_h(t + 1, hp);
    else { load_h(2 * t, h0); load_h(2 * t + 1, hp); }
    const bool valid = s * t < len;
    const long row = (long)b * a.Tout + t;
    float g[3][CPT], xh[3][CPT];
    float sm[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float mu = a.mean[j][row], rs = a.rstd[j][row];
      const float* dyr = a.dy[j] + row * C;
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        const float cv = valid ? w[j][0][k] * hm[k] + w[j][1][k] * h0[k] + w[j][2][k] * hp[k] : 0.f;
        xh[j][k] = (cv - mu) * rs;
        g[j][k] = dyr[tid + k * QT] * gam[j][k];
        sm[2 * j] += g[j][k];
        sm[2 * j + 1] += g[j][k] * xh[j][k];
      }
    }
    block_sums<6, NW>(sm, red);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float rs = a.rstd[j][row], s1 = sm[2 * j] * invC, s2 = sm[2 * j + 1] * invC;
      float* dcr = a.dc[j] + row * C;
#pragma unroll
      for (int k = 0; k < CPT; ++k) dcr[tid + k * QT] = valid ? rs * (g[j][k] - s1 - xh[j][k] * s2) : 0.f;
    }
    if (s == 1) {
#pragma unroll
      for (int k = 0; k < CPT; ++k) { hm[k] = h0[k]; h0[k] = hp[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < CPT; ++k) hm[k] = hp[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------- backward, parameters
// thread = one channel, block = (row chunk, 256-channel group): partial[chunk][15][C] = d gamma_j, d beta_j (6) and
// d w_j[tap] (9), from (h, dy_j, dc_j, statistics) -- no reductions inside the kernel
__global__ __launch_bounds__(QT) void qkv_pre_bwd_params_kernel(QkvBwdArgs a, float* __restrict__ partial, int rows_per_block) {
  const int c = blockIdx.y * QT + threadIdx.x;
  if (c >= a.C) return;
  const int C = a.C, T = a.T, s = a.stride;
  const long R = (long)a.B * a.Tout;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  float w[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) { w[j][0] = a.w[j][c * 3]; w[j][1] = a.w[j][c * 3 + 1]; w[j][2] = a.w[j][c * 3 + 2]; }
  float acc[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) acc[i] = 0.f;
  // two rows per iteration, all 18 loads of the pair issued up front from clamped (always valid) addresses and masked
  // afterwards: a `valid ? p[i] : 0.f` load is a branch with its own wait (DESIGN_LOG.md 3.7)
  struct RowIn { float hm, h0, hp, dy[3], dc[3]; bool valid, lo, hi; };
  auto fetch = [&](long r, RowIn& x) {
    const int t = (int)(r % a.Tout), b = (int)(r / a.Tout);
    const int tc = s * t;
    x.valid = tc < a.len[b]; x.lo = tc > 0; x.hi = tc + 1 < T;
    const float* hb = a.h + (long)b * T * C + c;
    x.hm = hb[(long)(tc > 0 ? tc - 1 : 0) * C];
    x.h0 = hb[(long)tc * C];
    x.hp = hb[(long)(tc + 1 < T ? tc + 1 : T - 1) * C];
#pragma unroll
    for (int j = 0; j < 3; ++j) { x.dy[j] = a.dy[j][r * C + c]; x.dc[j] = a.dc[j][r * C + c]; }
  };
  auto use = [&](long r, const RowIn& x) {
    const float hm = (x.valid && x.lo) ? x.hm : 0.f, h0 = x.valid ? x.h0 : 0.f, hp = (x.valid && x.hi) ? x.hp : 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float cv = x.valid ? w[j][0] * hm + w[j][1] * h0 + w[j][2] * hp : 0.f;
      const float xh = (cv - a.mean[j][r]) * a.rstd[j][r];
      acc[2 * j] += x.dy[j] * xh;
      acc[2 * j + 1] += x.dy[j];
      acc[6 + 3 * j] += x.dc[j] * hm;              // dc is already zero on masked rows
      acc[6 + 3 * j + 1] += x.dc[j] * h0;
      acc[6 + 3 * j + 2] += x.dc[j] * hp;
    }
  };
  long r = r0;
  for (; r + 2 <= r1; r += 2) {
    RowIn x0, x1;
    fetch(r, x0); fetch(r + 1, x1);
    use(r, x0); use(r + 1, x1);
  }
  if (r < r1) { RowIn x0; fetch(r, x0); use(r, x0); }
  // partial row: [6][C] LayerNorm gradients, then the tap gradients in the WEIGHT's own layout [j][C][3], so that the
  // reduced row holds d w_j ready to be viewed as [C, 1, 3] (no transposing copy per weight on the host side)
  float* p = partial + (long)blockIdx.x * 15 * C;
#pragma unroll
  for (int i = 0; i < 6; ++i) p[(long)i * C + c] = acc[i];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) p[(long)(6 + 3 * j) * C + c * 3 + k] = acc[6 + 3 * j + k];
}

// dh[b][t][c] = dh_ext + sum_j sum_tap w_j[c][tap] * dc_j[b][t'][c],  s*t' + tap - 1 == t
__global__ __launch_bounds__(QT) void qkv_pre_bwd_dh_kernel(QkvBwdArgs a, const float* __restrict__ dh_ext, float* __restrict__ dh) {
  const int C = a.C, C4 = C >> 2, s = a.stride;
  const long total = (long)a.B * a.T * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const long bt = i / C4;
    const int t = (int)(bt % a.T), b = (int)(bt / a.T);
    float4 o = dh_ext ? *reinterpret_cast<const float4*>(dh_ext + bt * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      const int num = t + 1 - tap;
      if (num < 0 || (num % s) != 0) continue;
      const int to = num / s;
      if (to >= a.Tout) continue;
      const long r = (long)b * a.Tout + to;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float4 g = *reinterpret_cast<const float4*>(a.dc[j] + r * C + c);
        const float* wj = a.w[j] + c * 3 + tap;
        o.x += wj[0] * g.x; o.y += wj[3] * g.y; o.z += wj[6] * g.z; o.w += wj[9] * g.w;
      }
    }
    *reinterpret_cast<float4*>(dh + bt * C + c) = o;
  }
}

int pick_seg(int B, int Tout) {
  // >= ~2 blocks per CU when the problem allows it, segments of at least 4 tokens (halo re-reads <= 50 % of ONE of the
  // five streams), at most 32
  int seg = 32;
  while (seg > 4 && (long)B * ((Tout + seg - 1) / seg) < 512) seg >>= 1;
  return seg;
}

bool bad_common(int B, int T, int C, int stride) {
  return B < 0 || T < 0 || C <= 0 || (stride != 1 && stride != 2) || (T % stride) != 0;
}

int param_blocks(long rows) {
  long b = (rows + 31) / 32;
  if (b > 96) b = 96;
  return b < 1 ? 1 : (int)b;
}

}  // namespace

// row widths: multiples of 256 (a lane owns 4 consecutive channels of every 256-channel chunk) up to 2304 (the
// backward row kernel keeps 9 channels per thread)
extern "C" int vilco_qkv_pre_supported(int32_t C) { return C > 0 && (C % 256) == 0 && C / 256 <= 9; }

// (channels per thread, threads per block) of a row width: 256 threads up to C = 1024, 768 threads for multiples of
// 768 up to 3072 (C = 2304: 3 per thread), else 256 threads with up to 9 per thread (sequential-token kernels only)
#define QKV_CASE(KERNEL, CPT_, NT_, GRID, ...) hipLaunchKernelGGL((KERNEL<CPT_, NT_>), GRID, dim3(NT_), 0, s, __VA_ARGS__)
#define QKV_DISPATCH(C_, KERNEL, GRID, ...)                                                                          \
  do {                                                                                                              \
    const int cpt256 = (C_) / 256;                                                                                  \
    if (cpt256 > 4 && (C_) % 768 == 0 && (C_) / 768 <= 4) {                                                         \
      switch ((C_) / 768) {                                                                                         \
        case 1: QKV_CASE(KERNEL, 1, 768, GRID, __VA_ARGS__); break;                                                 \
        case 2: QKV_CASE(KERNEL, 2, 768, GRID, __VA_ARGS__); break;                                                 \
        case 3: QKV_CASE(KERNEL, 3, 768, GRID, __VA_ARGS__); break;                                                 \
        default: QKV_CASE(KERNEL, 4, 768, GRID, __VA_ARGS__); break;                                                \
      }                                                                                                             \
    } else {                                                                                                        \
      switch (cpt256) {                                                                                             \
        case 1: QKV_CASE(KERNEL, 1, 256, GRID, __VA_ARGS__); break;                                                 \
        case 2: QKV_CASE(KERNEL, 2, 256, GRID, __VA_ARGS__); break;                                                 \
        case 3: QKV_CASE(KERNEL, 3, 256, GRID, __VA_ARGS__); break;                                                 \
        case 4: QKV_CASE(KERNEL, 4, 256, GRID, __VA_ARGS__); break;                                                 \
        case 5: QKV_CASE(KERNEL, 5, 256, GRID, __VA_ARGS__); break;                                                 \
        case 6: QKV_CASE(KERNEL, 6, 256, GRID, __VA_ARGS__); break;                                                 \
        case 7: QKV_CASE(KERNEL, 7, 256, GRID, __VA_ARGS__); break;                                                 \
        case 8: QKV_CASE(KERNEL, 8, 256, GRID, __VA_ARGS__); break;                                                 \
        default: QKV_CASE(KERNEL, 9, 256, GRID, __VA_ARGS__); break;                                                \
      }                                                                                                             \
    }                                                                                                               \
  } while (0)

template <int TB, int S>
void launch_fwd_tb(const QkvArgs& a, hipStream_t s) {
  const long waves = (long)a.B * ((a.Tout + TB - 1) / TB);
  hipLaunchKernelGGL((qkv_pre_fwd_kernel<TB, S>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
}

// tokens per wave: more tokens = fewer halo re-reads and parameter loads, fewer = more waves in flight and fewer
// registers.  Measured at [8, 2304, 2304] (r02): TB = 1 / 2 / 4 -> 1.90 / 2.49 / 2.16 TB/s.  VILCO_QKV_TB overrides (tuning).
int forced_tb() {
  static const int forced = [] { const char* e = getenv("VILCO_QKV_TB"); return e ? atoi(e) : 0; }();
  return forced;
}

// Ring kernel (fourth version): one workgroup per run of `seg` output tokens; runs sized so that one round of workgroups
// covers the 256 CUs (a run pays D + 1 pipeline intervals and two halo rows, so short runs are worth nothing).
int ring_seg(int B, int Tout) {
  int per_clip = 256 / B;
  if (per_clip < 1) per_clip = 1;
  int seg = (Tout + per_clip - 1) / per_clip;
  return seg < 8 ? 8 : seg;
}
int ring_mode() {          // VILCO_QKV_RING: 1 = always, 0 = never, unset = where it measured faster
  static const int m = [] { const char* e = getenv("VILCO_QKV_RING"); return e ? atoi(e) : -1; }();
  return m;
}
// Device time, forward, ring vs three-pass (r03, graph replays): [8,2304,2304] 112 vs 226 us, [2,2304,2304] 34 vs 66,
// [2,2304,1024] 20 vs 31, [2,576,1024] stride 2 13.5 vs 18; only runs too short to fill the pipeline lose ([2,256,512]: 12 vs 10).
bool use_ring(int B, int T, int stride) {
  const int m = ring_mode();
  if (m >= 0) return m != 0;
  return (long)B * (T / stride) >= 256;
}
template <int S>
void launch_ring(QkvArgs a, int npart, hipStream_t s) {
  a.seg = ring_seg(a.B, a.Tout);
  const dim3 grid((unsigned)(a.B * ((a.Tout + a.seg - 1) / a.seg)));
#define RING_CASE(N) case N: hipLaunchKernelGGL((qkv_pre_fwd_ring_kernel<N, S>), grid, dim3((RING_NA + N) * 64), 0, s, a, npart); break
  switch (a.C / 256) {
    RING_CASE(1); RING_CASE(2); RING_CASE(3); RING_CASE(4); RING_CASE(5); RING_CASE(6); RING_CASE(7); RING_CASE(8);
    default: RING_CASE(9);
  }
#undef RING_CASE
}

bool launch_fwd(const QkvArgs& a, hipStream_t s, int npart) {
  if (a.C % 256 != 0) return false;
  if (use_ring(a.B, a.T, a.stride) && !forced_tb()) {
    if (a.stride == 1) launch_ring<1>(a, npart, s); else launch_ring<2>(a, npart, s);
    return true;
  }
  const int forced = forced_tb();
  if (a.stride == 1) {
    int tb = forced ? forced : 2;
    if (tb >= 4) launch_fwd_tb<4, 1>(a, s);
    else if (tb == 1) launch_fwd_tb<1, 1>(a, s);
    else launch_fwd_tb<2, 1>(a, s);
  } else {
    int tb = forced ? forced : 2;
    if (tb >= 2) launch_fwd_tb<2, 2>(a, s);
    else launch_fwd_tb<1, 2>(a, s);
  }
  return true;
}

extern "C" int vilco_qkv_pre_amax_parts(int32_t B, int32_t T, int32_t stride) {
  if (B <= 0 || T <= 0 || (stride != 1 && stride != 2)) return 0;
  if (use_ring(B, T, stride) && !forced_tb()) {                 // ring kernel: one partial per workgroup
    const int Tout = T / stride;
    return B * ((Tout + ring_seg(B, Tout) - 1) / ring_seg(B, Tout));
  }
  const long waves = (long)B * ((T / stride + 1) / 2);          // TB = 2 tokens per wave, both strides (launch_fwd)
  return waves <= 4096 ? (int)waves : 0;                        // more partials than that cost the packs more than an amax launch
}

extern "C" int vilco_qkv_pre_fwd(const float* x, const float* ln1_g, const float* ln1_b, const float* const* w,
                                 const float* const* gam, const float* const* bet, const int32_t* len, float* h,
                                 float* const* y, float* mean1, float* rstd1, float* const* mean, float* const* rstd,
                                 float* const* amax_parts, int32_t B, int32_t T, int32_t C, int32_t stride, float eps1,
                                 float eps, void* stream) {
  if (!x || !w || !gam || !bet || !len || !y || !mean || !rstd || bad_common(B, T, C, stride)) return VILCO_ERR_BADARG;
  if (!vilco_qkv_pre_supported(C)) return VILCO_ERR_UNSUPPORTED;
  if (B == 0 || T == 0) return VILCO_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  QkvArgs a;
  a.x = x; a.g1 = ln1_g; a.b1 = ln1_b; a.len = len; a.h = h; a.mean1 = mean1; a.rstd1 = rstd1;
  for (int j = 0; j < 3; ++j) {
    if (!w[j] || !y[j]) return VILCO_ERR_BADARG;
    a.w[j] = w[j]; a.gam[j] = gam[j]; a.bet[j] = bet[j]; a.y[j] = y[j]; a.mean[j] = mean[j]; a.rstd[j] = rstd[j];
    a.amax[j] = (amax_parts && !forced_tb() && vilco_qkv_pre_amax_parts(B, T, stride) > 0) ? amax_parts[j] : nullptr;
  }
  a.B = B; a.T = T; a.Tout = T / stride; a.C = C; a.stride = stride; a.seg = 0;
  a.eps1 = eps1; a.eps = eps;
  if (!launch_fwd(a, s, a.amax[0] ? vilco_qkv_pre_amax_parts(B, T, stride) : 0)) return VILCO_ERR_UNSUPPORTED;
  return vilco_launch_status();
}

extern "C" size_t vilco_qkv_pre_bwd_workspace(int32_t B, int32_t T, int32_t C, int32_t stride) {
  if (bad_common(B, T, C, stride)) return 0;
  return (size_t)param_blocks((long)B * (T / stride)) * 15 * (size_t)C * sizeof(float);
}

// dparams = 15 C floats: d gamma_q, d beta_q, d gamma_k, d beta_k, d gamma_v, d beta_v as [6][C], then d w_q, d w_k, d w_v
// as [j][C][3] -- each the weight's own [C][1][3] layout
extern "C" int vilco_qkv_pre_bwd(const float* h, const float* const* w, const float* const* gam, const float* const* dy,
                                 const float* const* mean, const float* const* rstd, const int32_t* len,
                                 const float* dh_ext, float* const* dc, float* dh, float* dparams, int32_t B, int32_t T,
                                 int32_t C, int32_t stride, void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !w || !gam || !dy || !mean || !rstd || !len || !dc || !dh || !dparams || bad_common(B, T, C, stride))
    return VILCO_ERR_BADARG;
  if (!vilco_qkv_pre_supported(C)) return VILCO_ERR_UNSUPPORTED;
  if (B == 0 || T == 0) return VILCO_OK;
  if (!workspace || workspace_bytes < vilco_qkv_pre_bwd_workspace(B, T, C, stride)) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  QkvBwdArgs a;
  a.h = h; a.len = len;
  for (int j = 0; j < 3; ++j) {
    if (!w[j] || !dy[j] || !mean[j] || !rstd[j] || !dc[j]) return VILCO_ERR_BADARG;
    a.w[j] = w[j]; a.gam[j] = gam[j]; a.dy[j] = dy[j]; a.mean[j] = mean[j]; a.rstd[j] = rstd[j]; a.dc[j] = dc[j];
  }
  a.B = B; a.T = T; a.Tout = T / stride; a.C = C; a.stride = stride; a.seg = pick_seg(B, a.Tout);
  const dim3 grid((a.Tout + a.seg - 1) / a.seg, B);
  QKV_DISPATCH(C, qkv_pre_bwd_rows_kernel, grid, a);
  const long rows = (long)B * a.Tout;
  const int nb = param_blocks(rows);
  const int rpb = (int)((rows + nb - 1) / nb);
  float* partial = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(qkv_pre_bwd_params_kernel, dim3(nb, (C + QT - 1) / QT), dim3(QT), 0, s, a, partial, rpb);
  vilco_reduce_rows(partial, dparams, nullptr, nb, 15 * C, 15 * C, s);
  long blocks = ((long)B * T * (C / 4) + QT - 1) / QT;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(qkv_pre_bwd_dh_kernel, dim3((int)blocks), dim3(QT), 0, s, a, dh_ext, dh);
  return vilco_launch_status();
}
