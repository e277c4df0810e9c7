// LayerNorm over the channel dim of token-major rows (HBM-bound: 8*C bytes per row fwd).
// One wave per row, the row lives in registers (C <= 4096, C % 4 == 0): one read of x, one write
// of y; mean / biased variance by wavefront reductions.  Reference: blocks.py:160-175 and the
// stock nn.LayerNorm uses listed in include/vilco_hip.h.
#include "common.h"

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_WAVES = LN_THREADS / 64;

// FULL: C == 256 * NV, every lane owns NV float4 of the row -- no `c < C` guards.  A guarded load is a branch with its own
// s_waitcnt: the NV loads of a row become NV dependent round trips (seen in the ISA; same finding as qkvpre.hip).
// PLANES (round 4): the kernel also writes y as fp16 x2 operand planes (pack.h's format) for the matrix product that consumes
// it -- natural rows, or the k=3 convs' zero-padded per-sequence image.  The scale comes from a bound instead of the tensor's
// maximum (which only exists once every row is done): |xhat| < sqrt(C), so |y| < max|gamma| sqrt(C) + max|beta|, taken from the
// parameter chunks every wave already holds.  A bound k binary orders above the true maximum (k ~ 3 for C = 1024) keeps all 22
// significant bits of every element and raises the format's absolute floor from 2^-40 to 2^(k-40) of the tensor maximum.
struct LnPlanes {
  _Float16* p0;          // part 0; null = none
  long plane_stride;     // elements between the parts
  float* inv_scale;      // {1/s, s}
  long rows_out;         // rows of one plane (natural: rows32; image: vilco_tap_plane_rows)
  int seqT;              // 0: natural rows; T > 0: image rows b * (T + 2) + 1 + t
  const float* row_mask; // optional (with or without planes): y[row] *= row_mask[row % mask_rows] (the LevelCat heads' separator rows)
  long mask_rows;
};

template <int NV, bool FULL, bool PLANES>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, long rows, int C,
    float eps, int relu, float* __restrict__ amax_parts, LnPlanes po) {
  __shared__ float amax_red[LN_WAVES];
  float amax = 0.f;          // max |y| over this lane's outputs: the consumer's operand pack needs the tensor's amax
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
  const long wstride = (long)gridDim.x * LN_WAVES;
  const float invC = 1.0f / (float)C;

  float4 g[NV], b[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (FULL || c < C) {
      g[i] = gamma ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
      b[i] = beta ? *reinterpret_cast<const float4*>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }

  float fs = 0.f;
  if (PLANES) {
    float gmax = 0.f, bmax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (FULL || c < C) {
        gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(g[i].x), fabsf(g[i].y))), fmaxf(fabsf(g[i].z), fabsf(g[i].w)));
        bmax = fmaxf(fmaxf(bmax, fmaxf(fabsf(b[i].x), fabsf(b[i].y))), fmaxf(fabsf(b[i].z), fabsf(b[i].w)));
      }
    }
    const float bound = wave_max(gmax) * sqrtf((float)C) + wave_max(bmax);        // the same value in every wave of the grid
    int e = (int)((__float_as_uint(bound) >> 23) & 0xff);
    if (e < 15) e = 15;
    if (e > 250) e = 250;
    fs = __uint_as_float((unsigned)(268 - e) << 23);                               // bound * fs in [2^14, 2^15)
    if (blockIdx.x == 0 && threadIdx.x == 0) { po.inv_scale[0] = __uint_as_float((unsigned)(e - 14) << 23); po.inv_scale[1] = fs; }
  }

  for (long row = wid; row < rows; row += wstride) {
    const float* xr = x + row * C;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (FULL || c < C) {
        v[i] = *reinterpret_cast<const float4*>(xr + c);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    const float mu = wave_sum(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (FULL || c < C) {
        v[i].x -= mu; v[i].y -= mu; v[i].z -= mu; v[i].w -= mu;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
      }
    }
    const float var = wave_sum(q) * invC;
    const float rs = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
      if (mean) mean[row] = mu;
      if (rstd) rstd[row] = rs;
    }
    float* yr = y + row * C;
    const long orow = (PLANES && po.seqT) ? (row / po.seqT) * (po.seqT + 2) + 1 + row % po.seqT : row;
    const float rm = po.row_mask ? po.row_mask[row % po.mask_rows] : 1.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (FULL || c < C) {
        float4 o;
        o.x = v[i].x * rs * g[i].x + b[i].x;
        o.y = v[i].y * rs * g[i].y + b[i].y;
        o.z = v[i].z * rs * g[i].z + b[i].z;
        o.w = v[i].w * rs * g[i].w + b[i].w;
        if (relu) {
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        if (po.row_mask) { o.x *= rm; o.y *= rm; o.z *= rm; o.w *= rm; }
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        *reinterpret_cast<float4*>(yr + c) = o;
        if (PLANES) {
          typedef _Float16 h4 __attribute__((ext_vector_type(4)));
          const float xs[4] = {o.x * fs, o.y * fs, o.z * fs, o.w * fs};      // exact (power of two)
          h4 h0, h1;
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) { h0[e4] = (_Float16)xs[e4]; h1[e4] = (_Float16)(xs[e4] - (float)h0[e4]); }
          *reinterpret_cast<h4*>(po.p0 + orow * C + c) = h0;
          *reinterpret_cast<h4*>(po.p0 + po.plane_stride + orow * C + c) = h1;
        }
      }
    }
  }
  if (PLANES) {        // the planes' zero rows: below the last row (natural), around every sequence and below the last one (image)
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    const long first = po.seqT ? 0 : rows, nseq = po.seqT ? rows / po.seqT : 0;
    for (long pr = first + wid; pr < po.rows_out; pr += wstride) {
      if (po.seqT) {
        const long q = pr % (po.seqT + 2);
        if (pr < nseq * (po.seqT + 2) && q != 0 && q != po.seqT + 1) continue;
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (FULL || c < C) {
          *reinterpret_cast<h4*>(po.p0 + pr * C + c) = z;
          *reinterpret_cast<h4*>(po.p0 + po.plane_stride + pr * C + c) = z;
        }
      }
    }
  }
  if (amax_parts) {          // one partial maximum per block, folded by the pack kernel exactly like amax_kernel's
    amax = wave_max(amax);
    if (lane == 0) amax_red[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = amax_red[0];
#pragma unroll
      for (int w = 1; w < LN_WAVES; ++w) m = fmaxf(m, amax_red[w]);
      amax_parts[blockIdx.x] = m;
    }
  }
}

// dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)),  g = dy (masked by y>0 if relu)
// per-block partial dgamma/dbeta -> ws[block][2][C]
template <int NV, bool FULL, bool RELU>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
    float* __restrict__ dx, float* __restrict__ ws, long rows, int C,
    float* __restrict__ dgamma, float* __restrict__ dbeta, unsigned* sync, const float* __restrict__ dres,
    float* __restrict__ amax_parts) {          // amax_parts (round 6): max|dx| of this block, for the consumer's operand planes
  __shared__ float red[LN_WAVES][64 * 4];
  float amax = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wid = (long)blockIdx.x * LN_WAVES + wave;
  const long wstride = (long)gridDim.x * LN_WAVES;
  const float invC = 1.0f / (float)C;

  float4 gm[NV], dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    gm[i] = ((FULL || c < C) && gamma) ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  for (long row = wid; row < rows; row += wstride) {
    const float mu = mean[row], rs = rstd[row];
    float4 g[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (FULL || c < C) {
        g[i] = *reinterpret_cast<const float4*>(dy + row * C + c);
        if (RELU) {
          const float4 yy = *reinterpret_cast<const float4*>(y + row * C + c);
          if (!(yy.x > 0.f)) g[i].x = 0.f;
          if (!(yy.y > 0.f)) g[i].y = 0.f;
          if (!(yy.z > 0.f)) g[i].z = 0.f;
          if (!(yy.w > 0.f)) g[i].w = 0.f;
        }
        const float4 xv = *reinterpret_cast<const float4*>(x + row * C + c);
        xh[i].x = (xv.x - mu) * rs; xh[i].y = (xv.y - mu) * rs;
        xh[i].z = (xv.z - mu) * rs; xh[i].w = (xv.w - mu) * rs;
        dg[i].x += g[i].x * xh[i].x; dg[i].y += g[i].y * xh[i].y;
        dg[i].z += g[i].z * xh[i].z; dg[i].w += g[i].w * xh[i].w;
        db[i].x += g[i].x; db[i].y += g[i].y; db[i].z += g[i].z; db[i].w += g[i].w;
        g[i].x *= gm[i].x; g[i].y *= gm[i].y; g[i].z *= gm[i].z; g[i].w *= gm[i].w;
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
      }
    }
    s1 = wave_sum(s1) * invC;
    s2 = wave_sum(s2) * invC;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (FULL || c < C) {
        float4 o;
        o.x = rs * (g[i].x - s1 - xh[i].x * s2);
        o.y = rs * (g[i].y - s1 - xh[i].y * s2);
        o.z = rs * (g[i].z - s1 - xh[i].z * s2);
        o.w = rs * (g[i].w - s1 - xh[i].w * s2);
        if (dres) {                 // the gradient arriving over the residual connection around this LayerNorm's branch
          const float4 r = *reinterpret_cast<const float4*>(dres + row * C + c);
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        *reinterpret_cast<float4*>(dx + row * C + c) = o;
      }
    }
  }

  // block reduction of the per-wave partial sums, one 256-channel slab at a time
  float* wsb = ws + (long)blockIdx.x * 2 * C;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    for (int which = 0; which < 2; ++which) {
      const float4 v = which == 0 ? dg[i] : db[i];
      __syncthreads();
      *reinterpret_cast<float4*>(&red[wave][lane * 4]) = v;
      __syncthreads();
      if (wave == 0 && (FULL || c < C)) {
        float4 a = *reinterpret_cast<const float4*>(&red[0][lane * 4]);
#pragma unroll
        for (int w = 1; w < LN_WAVES; ++w) {
          const float4 t = *reinterpret_cast<const float4*>(&red[w][lane * 4]);
          a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        float* o = wsb + which * C + c;         // crosses the in-launch barrier: write-through stores
        vilco_st_agent(o, a.x); vilco_st_agent(o + 1, a.y); vilco_st_agent(o + 2, a.z); vilco_st_agent(o + 3, a.w);
      }
    }
  }
  if (amax_parts) {
    amax = wave_max(amax);
    __syncthreads();
    if (lane == 0) red[0][wave] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = red[0][0];
#pragma unroll
      for (int w = 1; w < LN_WAVES; ++w) m = fmaxf(m, red[0][w]);
      amax_parts[blockIdx.x] = m;
    }
  }
  // ws rows are [dgamma | dbeta] per block: finish the column sums in this launch (grid barrier), when the host could
  // give us a counter; otherwise a reduce_rows launch follows
  if (sync) vilco_finish_colsum(ws, dgamma, dbeta, (int)gridDim.x, 2 * C, C, sync, blockIdx.x, gridDim.x);
}

// out[j] = sum_r ws[r][j]  (j < ncols); also used by every two-stage column reduction.
// block = 64 columns x 4 row slices; slices are combined through LDS.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ ws,
                                                          float* __restrict__ out0,
                                                          float* __restrict__ out1, int nrows,
                                                          int ncols, int split) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (j < ncols) {
    int r = slice;
    for (; r + 28 < nrows; r += 32) {       // eight rows in flight; the sum keeps the row order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ws[(long)(r + 4 * u) * ncols + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < nrows; r += 4) s += ws[(long)r * ncols + j];
  }
  part[slice][lane] = s;
  __syncthreads();
  if (slice == 0 && j < ncols) {
    s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (out1 && j >= split) out1[j - split] = s;
    else out0[j] = s;
  }
}

int ln_blocks(long rows, long cap = 256) {
  long b = (rows + LN_WAVES - 1) / LN_WAVES;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

// shared with the other translation units
void vilco_reduce_rows(const float* ws, float* out0, float* out1, int nrows, int ncols, int split,
                       hipStream_t s) {
  if (vilco_defer_active()) {          // recorded; issued with the others by vilco_defer_flush (defer.hip)
    vilco_defer_push_rr(ws, out0, out1, nrows, ncols, split);
    return;
  }
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((ncols + 63) / 64), dim3(256), 0, s, ws, out0, out1,
                     nrows, ncols, split);
}

#define LN_CASE(N_, KERNEL, ...)                                                                              \
  if (C == (N_) * 256) hipLaunchKernelGGL((KERNEL<N_, true, LN_PLANES_>), grid, dim3(LN_THREADS), 0, s, __VA_ARGS__);      \
  else hipLaunchKernelGGL((KERNEL<N_, false, LN_PLANES_>), grid, dim3(LN_THREADS), 0, s, __VA_ARGS__);                      \
  break;
#define LN_DISPATCH(NVV, KERNEL, ...)                                                   \
  switch (NVV) {                                                                        \
    case 1: LN_CASE(1, KERNEL, __VA_ARGS__)                                             \
    case 2: LN_CASE(2, KERNEL, __VA_ARGS__)                                             \
    case 3: LN_CASE(3, KERNEL, __VA_ARGS__)                                             \
    case 4: LN_CASE(4, KERNEL, __VA_ARGS__)                                             \
    case 5: LN_CASE(5, KERNEL, __VA_ARGS__)                                             \
    case 6: LN_CASE(6, KERNEL, __VA_ARGS__)                                             \
    case 7: case 8: LN_CASE(8, KERNEL, __VA_ARGS__)                                     \
    case 9: LN_CASE(9, KERNEL, __VA_ARGS__)                                             \
    case 10: case 11: case 12: LN_CASE(12, KERNEL, __VA_ARGS__)                         \
    default: LN_CASE(16, KERNEL, __VA_ARGS__)                                           \
  }

#define LN_CASE_B(N_, KERNEL, R_, ...)                                                                        \
  if (C == (N_) * 256) hipLaunchKernelGGL((KERNEL<N_, true, R_>), grid, dim3(LN_THREADS), 0, s, __VA_ARGS__);  \
  else hipLaunchKernelGGL((KERNEL<N_, false, R_>), grid, dim3(LN_THREADS), 0, s, __VA_ARGS__);                  \
  break;
#define LN_DISPATCH_B(NVV, KERNEL, R_, ...)                                             \
  switch (NVV) {                                                                        \
    case 1: LN_CASE_B(1, KERNEL, R_, __VA_ARGS__)                                       \
    case 2: LN_CASE_B(2, KERNEL, R_, __VA_ARGS__)                                       \
    case 3: LN_CASE_B(3, KERNEL, R_, __VA_ARGS__)                                       \
    case 4: LN_CASE_B(4, KERNEL, R_, __VA_ARGS__)                                       \
    case 5: LN_CASE_B(5, KERNEL, R_, __VA_ARGS__)                                       \
    case 6: LN_CASE_B(6, KERNEL, R_, __VA_ARGS__)                                       \
    case 7: case 8: LN_CASE_B(8, KERNEL, R_, __VA_ARGS__)                               \
    case 9: LN_CASE_B(9, KERNEL, R_, __VA_ARGS__)                                       \
    case 10: case 11: case 12: LN_CASE_B(12, KERNEL, R_, __VA_ARGS__)                   \
    default: LN_CASE_B(16, KERNEL, R_, __VA_ARGS__)                                     \
  }

extern "C" int vilco_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                                   float* mean, float* rstd, int64_t rows, int32_t C, float eps,
                                   int32_t relu, void* stream) {
  return vilco_layernorm_fwd_amax(x, gamma, beta, y, mean, rstd, rows, C, eps, relu, nullptr, nullptr, stream);
}

extern "C" int vilco_layernorm_fwd_amax(const float* x, const float* gamma, const float* beta, float* y,
                                        float* mean, float* rstd, int64_t rows, int32_t C, float eps,
                                        int32_t relu, float* amax_parts, int32_t* n_parts, void* stream) {
  return vilco_layernorm_fwd_planes(x, gamma, beta, y, mean, rstd, rows, C, eps, relu, amax_parts, n_parts, nullptr, 0, 0, nullptr, 0,
                                    stream);
}

extern "C" size_t vilco_layernorm_planes_bytes(int64_t rows, int32_t C, int32_t seq_len) {
  if (rows <= 0 || C <= 0) return 0;
  if (seq_len > 0) return (size_t)(VILCO_PACK_HDR + (vilco_tap_plane_rows(rows / seq_len, seq_len) * C + 7) / 8 * 8 * 4);
  return (size_t)(VILCO_PACK_HDR + (rows + 31) / 32 * 32 * (long)C * 4);
}

extern "C" int vilco_layernorm_fwd_planes(const float* x, const float* gamma, const float* beta, float* y,
                                          float* mean, float* rstd, int64_t rows, int32_t C, float eps,
                                          int32_t relu, float* amax_parts, int32_t* n_parts, void* planes, size_t planes_bytes,
                                          int32_t seq_len, const float* row_mask, int64_t mask_rows, void* stream) {
  if (!x || !y || rows < 0 || C <= 0) return VILCO_ERR_BADARG;
  if (n_parts) *n_parts = 0;
  if (rows == 0) return VILCO_OK;
  if ((C % 4) != 0 || C > 4096) return VILCO_ERR_UNSUPPORTED;
  if (!vilco_aligned(x, 16) || !vilco_aligned(y, 16)) return VILCO_ERR_BADARG;
  if (row_mask && mask_rows <= 0) return VILCO_ERR_BADARG;
  LnPlanes po = {nullptr, 0, nullptr, 0, 0, row_mask, (long)mask_rows};
  if (planes) {
    // natural rows need C % 32 == 0 (no column padding); the convs' image C % 8 == 0 and whole sequences
    if (seq_len < 0 || !vilco_aligned(planes, 256)) return VILCO_ERR_BADARG;
    if (seq_len > 0 ? ((C % 8) != 0 || (rows % seq_len) != 0) : (C % 32) != 0) return VILCO_ERR_UNSUPPORTED;
    if (planes_bytes < vilco_layernorm_planes_bytes(rows, C, seq_len)) return VILCO_ERR_WORKSPACE;
    unsigned char* u = reinterpret_cast<unsigned char*>(planes);
    po.p0 = reinterpret_cast<_Float16*>(u + VILCO_PACK_HDR);
    po.rows_out = seq_len > 0 ? vilco_tap_plane_rows(rows / seq_len, seq_len) : (rows + 31) / 32 * 32;
    po.plane_stride = seq_len > 0 ? (po.rows_out * C + 7) / 8 * 8 : po.rows_out * (long)C;
    po.inv_scale = reinterpret_cast<float*>(u) + VILCO_AMAX_MAX_BLOCKS;
    po.seqT = seq_len;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nv = (C + 255) / 256;
  dim3 grid(ln_blocks(rows, 2048));
  if (amax_parts && n_parts) *n_parts = (int32_t)grid.x;
  if (planes) {
#define LN_PLANES_ true
    LN_DISPATCH(nv, ln_fwd_kernel, x, gamma, beta, y, mean, rstd, (long)rows, (int)C, eps, (int)relu,
                (amax_parts && n_parts) ? amax_parts : nullptr, po)
#undef LN_PLANES_
  } else {
#define LN_PLANES_ false
    LN_DISPATCH(nv, ln_fwd_kernel, x, gamma, beta, y, mean, rstd, (long)rows, (int)C, eps, (int)relu,
                (amax_parts && n_parts) ? amax_parts : nullptr, po)
#undef LN_PLANES_
  }
  return vilco_launch_status();
}

extern "C" size_t vilco_layernorm_bwd_workspace(int64_t rows, int32_t C) {
  return (size_t)ln_blocks(rows) * 2 * (size_t)C * sizeof(float);
}

extern "C" int vilco_layernorm_bwd(const float* dy, const float* x, const float* y,
                                   const float* gamma, const float* mean, const float* rstd,
                                   float* dx, float* dgamma, float* dbeta, int64_t rows, int32_t C,
                                   int32_t relu, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  return vilco_layernorm_bwd_res(dy, x, y, gamma, mean, rstd, nullptr, dx, dgamma, dbeta, rows, C, relu, workspace, workspace_bytes, stream);
}

extern "C" int vilco_layernorm_bwd_res(const float* dy, const float* x, const float* y,
                                       const float* gamma, const float* mean, const float* rstd, const float* dres,
                                       float* dx, float* dgamma, float* dbeta, int64_t rows, int32_t C,
                                       int32_t relu, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  return vilco_layernorm_bwd_res_amax(dy, x, y, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, C, relu, workspace, workspace_bytes,
                                      nullptr, nullptr, stream);
}

extern "C" int vilco_layernorm_bwd_res_amax(const float* dy, const float* x, const float* y,
                                            const float* gamma, const float* mean, const float* rstd, const float* dres,
                                            float* dx, float* dgamma, float* dbeta, int64_t rows, int32_t C,
                                            int32_t relu, void* workspace, size_t workspace_bytes,
                                            float* dx_amax_parts, int32_t* n_parts, void* stream) {
  if (n_parts) *n_parts = 0;
  if (!dy || !x || !mean || !rstd || !dx || rows < 0 || C <= 0) return VILCO_ERR_BADARG;
  if (relu && !y) return VILCO_ERR_BADARG;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return VILCO_ERR_BADARG;
  if (rows == 0) return VILCO_OK;
  if ((C % 4) != 0 || C > 4096) return VILCO_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < vilco_layernorm_bwd_workspace(rows, C)) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nv = (C + 255) / 256;
  const int nb = ln_blocks(rows);
  dim3 grid(nb);
  float* ws = reinterpret_cast<float*>(workspace);
  unsigned* sync = (dgamma && dbeta) ? vilco_sync_counter(s, VILCO_SITE_LN) : nullptr;   // nb <= 256 blocks: co-resident
  float* ap = (dx_amax_parts && n_parts) ? dx_amax_parts : nullptr;       // one partial per block (nb <= 2048: ln_blocks)
  if (ap) *n_parts = nb;
  if (relu) { LN_DISPATCH_B(nv, ln_bwd_kernel, true, dy, x, y, gamma, mean, rstd, dx, ws, (long)rows, (int)C, dgamma, dbeta, sync, dres, ap) }
  else { LN_DISPATCH_B(nv, ln_bwd_kernel, false, dy, x, y, gamma, mean, rstd, dx, ws, (long)rows, (int)C, dgamma, dbeta, sync, dres, ap) }
  if (dgamma && dbeta && !sync) vilco_reduce_rows(ws, dgamma, dbeta, nb, 2 * C, C, s);  // ws rows: [dgamma | dbeta]
  return vilco_launch_status();
}
