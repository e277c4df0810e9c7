// Depthwise k=3 conv (stride 1|2) and the MaxPool1d(3,2,1) skip path, token-major.
// Both are HBM-bound streaming kernels: channels are the contiguous dim, so every access is a
// coalesced float4; the 3 taps of neighbouring tokens re-hit L1/L2, not HBM.
// Algorithmic bytes per call (fwd): 4*C*(Tin + Tout) per clip.
// Reference: blocks.py:312-334 (depthwise MaskedConv1D), :106-130 (mask), :519-523,567 (pool skip).
#include "common.h"

void vilco_reduce_rows(const float* ws, float* out0, float* out1, int nrows, int ncols, int split,
                       hipStream_t s);

namespace {

constexpr int EW_THREADS = 256;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// w[c][3] for 4 consecutive channels -> 12 floats
struct W12 { float v[12]; };
__device__ __forceinline__ W12 ldw(const float* w, int c) {
  W12 r;
  const float4 a = ld4(w + c * 3), b = ld4(w + c * 3 + 4), d = ld4(w + c * 3 + 8);
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  r.v[8] = d.x; r.v[9] = d.y; r.v[10] = d.z; r.v[11] = d.w;
  return r;
}

__global__ __launch_bounds__(EW_THREADS) void dwconv3_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const int* __restrict__ in_len,
    float* __restrict__ y, int B, int Tin, int Tout, int C, int stride) {
  const int C4 = C >> 2;
  const long total = (long)B * Tout * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const long bt = i / C4;
    const int to = (int)(bt % Tout), b = (int)(bt / Tout);
    float4 o = f4zero();
    if (stride * to < in_len[b]) {
      const W12 ww = ldw(w, c);
      const float* xb = x + (long)b * Tin * C + c;
      const int tc = stride * to;
      const float4 xm = tc > 0 ? ld4(xb + (long)(tc - 1) * C) : f4zero();
      const float4 x0 = ld4(xb + (long)tc * C);
      const float4 xp = tc + 1 < Tin ? ld4(xb + (long)(tc + 1) * C) : f4zero();
      o.x = ww.v[0] * xm.x + ww.v[1] * x0.x + ww.v[2] * xp.x;
      o.y = ww.v[3] * xm.y + ww.v[4] * x0.y + ww.v[5] * xp.y;
      o.z = ww.v[6] * xm.z + ww.v[7] * x0.z + ww.v[8] * xp.z;
      o.w = ww.v[9] * xm.w + ww.v[10] * x0.w + ww.v[11] * xp.w;
    }
    st4(y + bt * C + c, o);
  }
}

// dx[b][t][c] = sum_j w[c][j] * dym[b][t'][c]  with stride*t' + j - 1 == t, dym = dy * valid(t')
__global__ __launch_bounds__(EW_THREADS) void dwconv3_bwd_dx_kernel(
    const float* __restrict__ dy, const float* __restrict__ w, const int* __restrict__ in_len,
    float* __restrict__ dx, int B, int Tin, int Tout, int C, int stride) {
  const int C4 = C >> 2;
  const long total = (long)B * Tin * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const long bt = i / C4;
    const int t = (int)(bt % Tin), b = (int)(bt / Tin);
    const int len = in_len[b];
    const W12 ww = ldw(w, c);
    const float* dyb = dy + (long)b * Tout * C + c;
    float4 o = f4zero();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int num = t + 1 - j;  // = stride * t'
      if (num < 0 || (num % stride) != 0) continue;
      const int to = num / stride;
      if (to >= Tout || stride * to >= len) continue;
      const float4 g = ld4(dyb + (long)to * C);
      o.x += ww.v[0 + j] * g.x;
      o.y += ww.v[3 + j] * g.y;
      o.z += ww.v[6 + j] * g.z;
      o.w += ww.v[9 + j] * g.w;
    }
    st4(dx + bt * C + c, o);
  }
}

// partial dw: block handles a slab of output rows; thread = 4 channels; ws[block][C*3]
__global__ __launch_bounds__(EW_THREADS) void dwconv3_bwd_dw_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const int* __restrict__ in_len,
    float* __restrict__ ws, int B, int Tin, int Tout, int C, int stride, int rows_per_block, float* __restrict__ dw,
    unsigned* sync) {
  const int C4 = C >> 2;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  const long R = (long)B * Tout;
  if (r1 > R) r1 = R;
  for (int cg = threadIdx.x; cg < C4; cg += blockDim.x) {
    const int c = cg * 4;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (long r = r0; r < r1; ++r) {
      const int to = (int)(r % Tout), b = (int)(r / Tout);
      const int tc = stride * to;
      if (tc >= in_len[b]) continue;
      const float4 g = ld4(dy + r * C + c);
      const float* xb = x + (long)b * Tin * C + c;
      const float4 xm = tc > 0 ? ld4(xb + (long)(tc - 1) * C) : f4zero();
      const float4 x0 = ld4(xb + (long)tc * C);
      const float4 xp = tc + 1 < Tin ? ld4(xb + (long)(tc + 1) * C) : f4zero();
      acc[0] += g.x * xm.x; acc[1] += g.x * x0.x; acc[2] += g.x * xp.x;
      acc[3] += g.y * xm.y; acc[4] += g.y * x0.y; acc[5] += g.y * xp.y;
      acc[6] += g.z * xm.z; acc[7] += g.z * x0.z; acc[8] += g.z * xp.z;
      acc[9] += g.w * xm.w; acc[10] += g.w * x0.w; acc[11] += g.w * xp.w;
    }
    float* o = ws + (long)blockIdx.x * C * 3 + c * 3;
#pragma unroll
    for (int k = 0; k < 12; ++k) vilco_st_agent(o + k, acc[k]);      // crosses the in-launch barrier: write-through
  }
  if (sync) vilco_finish_colsum(ws, dw, nullptr, (int)gridDim.x, C * 3, C * 3, sync, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ float4 max4(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

// y[b][t'][c] = valid(t') * max(x[2t'-1], x[2t'], x[2t'+1])   (out-of-range = -inf, as MaxPool1d pads)
__global__ __launch_bounds__(EW_THREADS) void maxpool_fwd_kernel(
    const float* __restrict__ x, const int* __restrict__ in_len, float* __restrict__ y, int B,
    int Tin, int Tout, int C) {
  const int C4 = C >> 2;
  const long total = (long)B * Tout * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const long bt = i / C4;
    const int to = (int)(bt % Tout), b = (int)(bt / Tout);
    float4 o = f4zero();
    if (2 * to < in_len[b]) {
      const float* xb = x + (long)b * Tin * C + c;
      const int tc = 2 * to;
      o = ld4(xb + (long)tc * C);
      if (tc > 0) o = max4(o, ld4(xb + (long)(tc - 1) * C));
      if (tc + 1 < Tin) o = max4(o, ld4(xb + (long)(tc + 1) * C));
    }
    st4(y + bt * C + c, o);
  }
}

// argmax position inside window t' (first maximum wins, like aten max_pool1d): returns 2t'-1, 2t' or 2t'+1
__device__ __forceinline__ int win_argmax(const float* xb, long C, int to, int Tin) {
  const int tc = 2 * to;
  int best = tc > 0 ? tc - 1 : tc;
  float bv = xb[(long)best * C];
  for (int t = best + 1; t <= tc + 1 && t < Tin; ++t) {
    const float v = xb[(long)t * C];
    if (v > bv) { bv = v; best = t; }
  }
  return best;
}

__global__ __launch_bounds__(EW_THREADS) void maxpool_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const int* __restrict__ in_len,
    float* __restrict__ dx, int B, int Tin, int Tout, int C) {
  const long total = (long)B * Tin * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long bt = i / C;
    const int t = (int)(bt % Tin), b = (int)(bt / Tin);
    const int len = in_len[b];
    const float* xb = x + (long)b * Tin * C + c;
    const float* dyb = dy + (long)b * Tout * C + c;
    float o = 0.f;
    // windows that contain t: t even -> t/2 ; t odd -> (t-1)/2 and (t+1)/2
    const int w0 = (t & 1) ? (t - 1) / 2 : t / 2;
    const int w1 = (t & 1) ? (t + 1) / 2 : -1;
    if (w0 < Tout && 2 * w0 < len && win_argmax(xb, C, w0, Tin) == t) o += dyb[(long)w0 * C];
    if (w1 >= 0 && w1 < Tout && 2 * w1 < len && win_argmax(xb, C, w1, Tin) == t) o += dyb[(long)w1 * C];
    dx[i] = o;
  }
}

int ew_grid(long total) {
  long b = (total + EW_THREADS - 1) / EW_THREADS;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

int dw_blocks(long rows) {
  long b = (rows + 15) / 16;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int vilco_dwconv3_fwd(const float* x, const float* w, const int32_t* in_len, float* y,
                                 int32_t B, int32_t Tin, int32_t C, int32_t stride, void* stream) {
  if (!x || !w || !in_len || !y || B < 0 || Tin < 0 || C <= 0) return VILCO_ERR_BADARG;
  if (stride != 1 && stride != 2) return VILCO_ERR_UNSUPPORTED;
  if ((C % 4) != 0 || (Tin % stride) != 0) return VILCO_ERR_UNSUPPORTED;
  if (B == 0 || Tin == 0) return VILCO_OK;
  const int Tout = Tin / stride;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(dwconv3_fwd_kernel, dim3(ew_grid((long)B * Tout * (C / 4))), dim3(EW_THREADS), 0, s,
                     x, w, in_len, y, B, Tin, Tout, C, stride);
  return vilco_launch_status();
}

extern "C" size_t vilco_dwconv3_bwd_workspace(int32_t B, int32_t Tin, int32_t C, int32_t stride) {
  const long rows = (long)B * (Tin / (stride > 0 ? stride : 1));
  return (size_t)dw_blocks(rows) * (size_t)C * 3 * sizeof(float);
}

extern "C" int vilco_dwconv3_bwd(const float* dy, const float* x, const float* w,
                                 const int32_t* in_len, float* dx, float* dw, int32_t B, int32_t Tin,
                                 int32_t C, int32_t stride, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  if (!dy || !x || !w || !in_len || B < 0 || Tin < 0 || C <= 0) return VILCO_ERR_BADARG;
  if (stride != 1 && stride != 2) return VILCO_ERR_UNSUPPORTED;
  if ((C % 4) != 0 || (Tin % stride) != 0) return VILCO_ERR_UNSUPPORTED;
  if (B == 0 || Tin == 0) return VILCO_OK;
  const int Tout = Tin / stride;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dx)
    hipLaunchKernelGGL(dwconv3_bwd_dx_kernel, dim3(ew_grid((long)B * Tin * (C / 4))), dim3(EW_THREADS), 0,
                       s, dy, w, in_len, dx, B, Tin, Tout, C, stride);
  if (dw) {
    if (!workspace || workspace_bytes < vilco_dwconv3_bwd_workspace(B, Tin, C, stride)) return VILCO_ERR_WORKSPACE;
    const long rows = (long)B * Tout;
    const int nb = dw_blocks(rows);
    const int rpb = (int)((rows + nb - 1) / nb);
    float* ws = reinterpret_cast<float*>(workspace);
    unsigned* sync = vilco_sync_counter(s, VILCO_SITE_DWCONV);          // nb <= 512 blocks: co-resident
    hipLaunchKernelGGL(dwconv3_bwd_dw_kernel, dim3(nb), dim3(EW_THREADS), 0, s, dy, x, in_len, ws, B, Tin,
                       Tout, C, stride, rpb, dw, sync);
    if (!sync) vilco_reduce_rows(ws, dw, nullptr, nb, C * 3, C * 3, s);
  }
  return vilco_launch_status();
}

extern "C" int vilco_maxpool3s2_fwd(const float* x, const int32_t* in_len, float* y, int32_t B,
                                    int32_t Tin, int32_t C, void* stream) {
  if (!x || !in_len || !y || B < 0 || Tin < 0 || C <= 0) return VILCO_ERR_BADARG;
  if ((C % 4) != 0 || (Tin % 2) != 0) return VILCO_ERR_UNSUPPORTED;
  if (B == 0 || Tin == 0) return VILCO_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid((long)B * (Tin / 2) * (C / 4))), dim3(EW_THREADS), 0,
                     s, x, in_len, y, B, Tin, Tin / 2, C);
  return vilco_launch_status();
}

extern "C" int vilco_maxpool3s2_bwd(const float* dy, const float* x, const int32_t* in_len, float* dx,
                                    int32_t B, int32_t Tin, int32_t C, void* stream) {
  if (!dy || !x || !in_len || !dx || B < 0 || Tin < 0 || C <= 0) return VILCO_ERR_BADARG;
  if ((Tin % 2) != 0) return VILCO_ERR_UNSUPPORTED;
  if (B == 0 || Tin == 0) return VILCO_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid((long)B * Tin * C)), dim3(EW_THREADS), 0, s, dy, x,
                     in_len, dx, B, Tin, Tin / 2, C);
  return vilco_launch_status();
}
