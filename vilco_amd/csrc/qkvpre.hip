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
// Mapping: a 256-thread block walks a segment of SEG consecutive output tokens of one clip; thread i owns channels
// i, i + 256, ... (CPT = C / 256 of them), so every global access is a fully coalesced 1 KB line per instruction and a
// token's LayerNorm statistics are two block reductions (exact two-pass mean / variance, as the stand-alone kernel).
// The normalised input rows t-1, t, t+1 slide through registers: every x row is read once (plus one halo row per
// segment side), every h row normalised once per segment.
//
// Backward (qkv_pre_bwd_*): the conv outputs are never stored -- they are recomputed from h in registers.
//   rows:    per output token, x_hat_j from (h, stats), LayerNorm backward -> d(conv out)_j written (3 tensors)
//   params:  per channel, d gamma_j / d beta_j / d w_j accumulated over row chunks (no reductions: statistics are saved)
//   dh:      dh = sum_j dwconv^T(dc_j) + dh from the other consumers, one pass
// followed by the ordinary LayerNorm backward of LN1 (norm.hip).
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

// NT threads per block (256, or 768 for wide rows: C = 2304 is then 3 channels per thread), thread i owns channels
// i, i + NT, ...  A block takes SEG consecutive output tokens and works on ALL of them at once: the SEG*s + 2 input rows
// are loaded up front (every load of the block in flight together), their LayerNorm statistics are TWO block reductions
// for the whole batch of rows (not two per row), the 3*SEG conv rows two more.  Four barriers-with-reduction per
// segment instead of four per token: the first version (one token at a time) ran at 1.4 TB/s, latency-bound.
template <int CPT, int NT, int SEG, int S>
__global__ __launch_bounds__(NT) void qkv_pre_fwd_kernel(QkvArgs a) {
  constexpr int QT = NT, NW = NT / 64, NR = SEG * S + 2;        // input rows s*t0 - 1 .. s*t0 + SEG*S
  __shared__ float red[NW * (3 * SEG > NR ? 3 * SEG : NR)];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * SEG;
  const int C = a.C, T = a.T, tid = threadIdx.x;
  const float invC = 1.f / (float)C;
  const int len = a.len[b];
  const int r0 = S * t0 - 1;

  // ---- LN1 of the NR input rows (rows outside [0, T) are the conv's zero padding)
  float h[NR][CPT];
  float sm[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int t = r0 + r;
    sm[r] = 0.f;
    if (t >= 0 && t < T) {
      const float* xr = a.x + ((long)b * T + t) * C;
#pragma unroll
      for (int k = 0; k < CPT; ++k) { h[r][k] = xr[tid + k * QT]; sm[r] += h[r][k]; }
    } else {
#pragma unroll
      for (int k = 0; k < CPT; ++k) h[r][k] = 0.f;
    }
  }
  block_sums<NR, NW>(sm, red);
  float sq[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    sm[r] *= invC;
    sq[r] = 0.f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) { h[r][k] -= sm[r]; sq[r] += h[r][k] * h[r][k]; }
  }
  block_sums<NR, NW>(sq, red);
  {
    float g1[CPT], b1[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) { g1[k] = a.g1 ? a.g1[tid + k * QT] : 1.f; b1[k] = a.b1 ? a.b1[tid + k * QT] : 0.f; }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int t = r0 + r;
      const bool in = t >= 0 && t < T;
      const float rs = 1.0f / sqrtf(sq[r] * invC + a.eps1);
#pragma unroll
      for (int k = 0; k < CPT; ++k) h[r][k] = in ? h[r][k] * rs * g1[k] + b1[k] : 0.f;
      // rows [S*t0, S*t0 + S*SEG) are OWNED by this block (the halo rows belong to the neighbours)
      if (in && r >= 1 && r <= SEG * S) {
        if (a.h) {
          float* hr = a.h + ((long)b * T + t) * C;
#pragma unroll
          for (int k = 0; k < CPT; ++k) hr[tid + k * QT] = h[r][k];
        }
        if (tid == 0 && a.mean1) { a.mean1[(long)b * T + t] = sm[r]; a.rstd1[(long)b * T + t] = rs; }
      }
    }
  }

  // ---- the three depthwise convs of the SEG output tokens, masked, then their LayerNorms
  float c[3][SEG][CPT];
  float cs[3 * SEG];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float w0[CPT], w1[CPT], w2[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const float* wp = a.w[j] + (tid + k * QT) * 3;
      w0[k] = wp[0]; w1[k] = wp[1]; w2[k] = wp[2];
    }
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
      const int t = t0 + i;
      const bool valid = t < a.Tout && S * t < len;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        c[j][i][k] = valid ? w0[k] * h[S * i][k] + w1[k] * h[S * i + 1][k] + w2[k] * h[S * i + 2][k] : 0.f;
        acc += c[j][i][k];
      }
      cs[j * SEG + i] = acc;
    }
  }
  block_sums<3 * SEG, NW>(cs, red);
  float cq[3 * SEG];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
      const float mu = cs[j * SEG + i] * invC;
      cs[j * SEG + i] = mu;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < CPT; ++k) { c[j][i][k] -= mu; acc += c[j][i][k] * c[j][i][k]; }
      cq[j * SEG + i] = acc;
    }
  block_sums<3 * SEG, NW>(cq, red);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float gam[CPT], bet[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) { gam[k] = a.gam[j] ? a.gam[j][tid + k * QT] : 1.f; bet[k] = a.bet[j] ? a.bet[j][tid + k * QT] : 0.f; }
#pragma unroll
    for (int i = 0; i < SEG; ++i) {
      const int t = t0 + i;
      if (t >= a.Tout) continue;
      const long row = (long)b * a.Tout + t;
      const float rs = 1.0f / sqrtf(cq[j * SEG + i] * invC + a.eps);
      float* yr = a.y[j] + row * C;
#pragma unroll
      for (int k = 0; k < CPT; ++k) yr[tid + k * QT] = c[j][i][k] * rs * gam[k] + bet[k];
      if (tid == 0 && a.mean[j]) { a.mean[j][row] = cs[j * SEG + i]; a.rstd[j][row] = rs; }
    }
  }
}

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
  for (long r = r0; r < r1; ++r) {
    const int t = (int)(r % a.Tout), b = (int)(r / a.Tout);
    const bool valid = s * t < a.len[b];
    const int tc = s * t;
    const float* hb = a.h + (long)b * T * C + c;
    const float hm = (valid && tc > 0) ? hb[(long)(tc - 1) * C] : 0.f;
    const float h0 = valid ? hb[(long)tc * C] : 0.f;
    const float hp = (valid && tc + 1 < T) ? hb[(long)(tc + 1) * C] : 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float cv = valid ? w[j][0] * hm + w[j][1] * h0 + w[j][2] * hp : 0.f;
      const float xh = (cv - a.mean[j][r]) * a.rstd[j][r];
      const float dy = a.dy[j][r * C + c];
      acc[2 * j] += dy * xh;
      acc[2 * j + 1] += dy;
      const float dc = a.dc[j][r * C + c];          // already zero on masked rows
      acc[6 + 3 * j] += dc * hm;
      acc[6 + 3 * j + 1] += dc * h0;
      acc[6 + 3 * j + 2] += dc * hp;
    }
  }
  float* p = partial + (long)blockIdx.x * 15 * C + c;
#pragma unroll
  for (int i = 0; i < 15; ++i) p[(long)i * C] = acc[i];
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

// widths the forward kernel has an instantiation for: multiples of 256 up to 1024, of 512 up to 2048, of 768 up to 3072
extern "C" int vilco_qkv_pre_supported(int32_t C) {
  if (C <= 0 || (C % 256) != 0 || C / 256 > 9) return 0;
  return (C / 256 <= 4) || (C % 768 == 0 && C / 768 <= 4) || (C % 512 == 0 && C / 512 <= 4);
}

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

// forward: batch-of-rows kernel, (CPT, NT) in {1..4} x {256, 768}; SEG output tokens per block by register budget
template <int CPT, int NT>
void launch_fwd_seg(const QkvArgs& a, hipStream_t s) {
  constexpr int SEG1 = CPT <= 2 ? 8 : 4, SEG2 = CPT <= 2 ? 4 : 2;       // stride 1 / stride 2 (twice the input rows)
  if (a.stride == 1) {
    const dim3 grid((a.Tout + SEG1 - 1) / SEG1, a.B);
    hipLaunchKernelGGL((qkv_pre_fwd_kernel<CPT, NT, SEG1, 1>), grid, dim3(NT), 0, s, a);
  } else {
    const dim3 grid((a.Tout + SEG2 - 1) / SEG2, a.B);
    hipLaunchKernelGGL((qkv_pre_fwd_kernel<CPT, NT, SEG2, 2>), grid, dim3(NT), 0, s, a);
  }
}

bool launch_fwd(const QkvArgs& a, hipStream_t s) {
  const int C = a.C;
  if (C % 768 == 0 && C / 768 >= 2 && C / 768 <= 4) {
    switch (C / 768) {
      case 2: launch_fwd_seg<2, 768>(a, s); break;
      case 3: launch_fwd_seg<3, 768>(a, s); break;
      default: launch_fwd_seg<4, 768>(a, s); break;
    }
    return true;
  }
  if (C % 256 == 0 && C / 256 <= 4) {
    switch (C / 256) {
      case 1: launch_fwd_seg<1, 256>(a, s); break;
      case 2: launch_fwd_seg<2, 256>(a, s); break;
      case 3: launch_fwd_seg<3, 256>(a, s); break;
      default: launch_fwd_seg<4, 256>(a, s); break;
    }
    return true;
  }
  if (C % 512 == 0 && C / 512 <= 4) {      // 1280 is not, 1536 / 2048 are: 512 threads
    switch (C / 512) {
      case 3: launch_fwd_seg<3, 512>(a, s); break;
      default: launch_fwd_seg<4, 512>(a, s); break;
    }
    return true;
  }
  return false;
}

extern "C" int vilco_qkv_pre_fwd(const float* x, const float* ln1_g, const float* ln1_b, const float* const* w,
                                 const float* const* gam, const float* const* bet, const int32_t* len, float* h,
                                 float* const* y, float* mean1, float* rstd1, float* const* mean, float* const* rstd,
                                 int32_t B, int32_t T, int32_t C, int32_t stride, float eps1, float eps, void* stream) {
  if (!x || !w || !gam || !bet || !len || !y || !mean || !rstd || bad_common(B, T, C, stride)) return VILCO_ERR_BADARG;
  if (!vilco_qkv_pre_supported(C)) return VILCO_ERR_UNSUPPORTED;
  if (B == 0 || T == 0) return VILCO_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  QkvArgs a;
  a.x = x; a.g1 = ln1_g; a.b1 = ln1_b; a.len = len; a.h = h; a.mean1 = mean1; a.rstd1 = rstd1;
  for (int j = 0; j < 3; ++j) {
    if (!w[j] || !y[j]) return VILCO_ERR_BADARG;
    a.w[j] = w[j]; a.gam[j] = gam[j]; a.bet[j] = bet[j]; a.y[j] = y[j]; a.mean[j] = mean[j]; a.rstd[j] = rstd[j];
  }
  a.B = B; a.T = T; a.Tout = T / stride; a.C = C; a.stride = stride; a.seg = 0;
  a.eps1 = eps1; a.eps = eps;
  if (!launch_fwd(a, s)) return VILCO_ERR_UNSUPPORTED;
  return vilco_launch_status();
}

extern "C" size_t vilco_qkv_pre_bwd_workspace(int32_t B, int32_t T, int32_t C, int32_t stride) {
  if (bad_common(B, T, C, stride)) return 0;
  return (size_t)param_blocks((long)B * (T / stride)) * 15 * (size_t)C * sizeof(float);
}

// dparams = [15][C]: d gamma_q, d beta_q, d gamma_k, d beta_k, d gamma_v, d beta_v, then d w_q[tap 0..2], d w_k, d w_v as
// [j][tap][C] planes (the caller re-lays them to the [C][1][3] weight shape)
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
