// Memory-bound glue kernels: masked row softmax (materialised-score attention of round 1), the
// XLNet relative shift, residual/scale/mask epilogues, activation backward, column reductions,
// layout transposes.  All token-major, float4 where the channel count allows.
#include "common.h"

void vilco_reduce_rows(const float* ws, float* out0, float* out1, int nrows, int ncols, int split,
                       hipStream_t s);

namespace {

constexpr int EW_THREADS = 256;

int ew_grid(long total) {
  long b = (total + EW_THREADS - 1) / EW_THREADS;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// ---------------------------------------------------------------- softmax over rows of S[B][H][Tq][Tk]
// one wave per row; three passes over the row (max, sum, write) -- rows are <= 9 KB and stay in L1/L2.
__global__ __launch_bounds__(EW_THREADS) void softmax_fwd_kernel(float* __restrict__ s,
                                                                 const int* __restrict__ kv_len, int B,
                                                                 int H, int Tq, int Tk, int mode) {
  const int lane = threadIdx.x & 63;
  const long nrows = (long)B * H * Tq;
  const long wid = (long)blockIdx.x * (EW_THREADS / 64) + (threadIdx.x >> 6);
  const long wstride = (long)gridDim.x * (EW_THREADS / 64);
  for (long row = wid; row < nrows; row += wstride) {
    float* p = s + row * Tk;
    const int i = (int)(row % Tq);
    const int b = (int)(row / ((long)H * Tq));
    const int len = (mode == 2) ? Tk : kv_len[b];
    // mode 0: columns >= len are excluded (-inf).  mode 1 (XLNet): "- 1e30" on masked columns,
    // except the diagonal; in fp32 score - 1e30 == -1e30 exactly for any realistic score.
    float m = -INFINITY;
    for (int j = lane; j < Tk; j += 64) {
      float v = p[j];
      if (j >= len) v = (mode == 1) ? ((j == i) ? v : v - 1e30f) : -INFINITY;
      m = fmaxf(m, v);
    }
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < Tk; j += 64) {
      float v = p[j];
      if (j >= len) v = (mode == 1) ? ((j == i) ? v : v - 1e30f) : -INFINITY;
      sum += expf(v - m);
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < Tk; j += 64) {
      float v = p[j];
      if (j >= len) v = (mode == 1) ? ((j == i) ? v : v - 1e30f) : -INFINITY;
      p[j] = expf(v - m) * inv;
    }
  }
}

__global__ __launch_bounds__(EW_THREADS) void softmax_bwd_kernel(float* __restrict__ dp,
                                                                 const float* __restrict__ p,
                                                                 long nrows, int Tk) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * (EW_THREADS / 64) + (threadIdx.x >> 6);
  const long wstride = (long)gridDim.x * (EW_THREADS / 64);
  for (long row = wid; row < nrows; row += wstride) {
    float* g = dp + row * Tk;
    const float* pr = p + row * Tk;
    float dot = 0.f;
    for (int j = lane; j < Tk; j += 64) dot += g[j] * pr[j];
    dot = wave_sum(dot);
    for (int j = lane; j < Tk; j += 64) g[j] = pr[j] * (g[j] - dot);
  }
}

// s[b][h][i][j] += scale * bd[b][h][i][T - i + j]   (bd rows are 2T long)
__global__ __launch_bounds__(EW_THREADS) void relshift_add_kernel(float* __restrict__ s,
                                                                  const float* __restrict__ bd,
                                                                  float scale, long nrows, int T) {
  const long total = nrows * T;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % T);
    const long row = idx / T;
    const int i = (int)(row % T);
    s[idx] += scale * bd[row * 2 * T + (T - i + j)];
  }
}

__global__ __launch_bounds__(EW_THREADS) void relshift_bwd_kernel(const float* __restrict__ ds,
                                                                  float* __restrict__ dbd, float scale,
                                                                  long nrows, int T) {
  const long total = nrows * 2 * T;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % (2 * T));
    const long row = idx / (2 * T);
    const int i = (int)(row % T);
    const int j = q - T + i;  // q = T - i + j
    dbd[idx] = (j >= 0 && j < T) ? scale * ds[row * T + j] : 0.f;
  }
}

// ---------------------------------------------------------------- residual / scale / mask glue
template <bool VEC>
__global__ __launch_bounds__(EW_THREADS) void scale_add_fwd_kernel(
    float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ bval,
    const float* __restrict__ colscale, const float* __restrict__ rowscale, const int* __restrict__ len,
    int mask_a, int B, int T, int C) {
  constexpr int V = VEC ? 4 : 1;
  const int CV = C / V;
  const long total = (long)B * T * CV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * V;
    const long bt = i / CV;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const float am = (mask_a && len && t >= len[b]) ? 0.f : 1.f;
    const float rs = rowscale ? rowscale[b] : 1.f;
    const long o = bt * C + c;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float cs = colscale ? colscale[c + e] : 1.f;
      out[o + e] = (a ? a[o + e] * am : 0.f) + cs * rs * bval[o + e];
    }
  }
}

// da = dout*am ; db = dout*cs*rs ; partial dcolscale[c] = sum dout*bval*rs  -> ws[row chunk][C]
// grid = (row chunks, column chunks of 256): every thread owns one column of one row chunk, so loads are
// coalesced across the block and there are rows/32 * C/256 blocks to fill the chip.
__global__ __launch_bounds__(EW_THREADS) void scale_add_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ bval, const float* __restrict__ colscale,
    const float* __restrict__ rowscale, const int* __restrict__ len, int mask_a, float* __restrict__ da,
    float* __restrict__ db, float* __restrict__ ws, int B, int T, int C, int rows_per_block,
    float* __restrict__ dcolscale, unsigned* sync, float* __restrict__ db_amax) {
  __shared__ float amax_red[EW_THREADS / 64];
  float amax = 0.f;          // max |db| of this block: db is the upstream gradient of the branch's last layer (see act_bwd_kernel<true>)
  const long R = (long)B * T;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c < C) {
    const float cs = colscale ? colscale[c] : 1.f;
    float acc = 0.f;
    // rows in groups of four with the loads issued together: a row loop with one load -> use -> store chain per
    // iteration is one memory round trip per row (36 of them per thread at 4608 rows)
    auto one = [&](long r, float g, float bv) {
      const int t = (int)(r % T), b = (int)(r / T);
      const float am = (mask_a && len && t >= len[b]) ? 0.f : 1.f;
      const float rs = rowscale ? rowscale[b] : 1.f;
      if (da) da[r * C + c] = g * am;
      if (db) { const float v = g * cs * rs; db[r * C + c] = v; amax = fmaxf(amax, fabsf(v)); }
      if (ws) acc += g * bv * rs;
    };
    long r = r0;
    for (; r + 4 <= r1; r += 4) {
      float g[4], bv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] = dout[(r + u) * C + c];
      if (ws) {
#pragma unroll
        for (int u = 0; u < 4; ++u) bv[u] = bval[(r + u) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(r + u, g[u], bv[u]);
    }
    for (; r < r1; ++r) one(r, dout[r * C + c], ws ? bval[r * C + c] : 0.f);
    if (ws) vilco_st_agent(ws + (long)blockIdx.x * C + c, acc);
  }
  if (db_amax) {
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0)
      db_amax[blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(amax_red[0], amax_red[1]), fmaxf(amax_red[2], amax_red[3]));
  }
  if (sync) vilco_finish_colsum(ws, dcolscale, nullptr, (int)gridDim.x, C, C, sync, blockIdx.y * gridDim.x + blockIdx.x,
                                gridDim.x * gridDim.y);
}

__global__ __launch_bounds__(EW_THREADS) void axpby_kernel(float* __restrict__ out,
                                                           const float* __restrict__ a,
                                                           const float* __restrict__ b, float alpha,
                                                           float beta, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

// Producer-written operand planes (round 4).  dz goes nowhere but into the two backward products of its layer, as fp16 x2
// planes (pack.h): a plane element can be written the moment its value exists IF the tensor's scale is known -- and a scale
// needs no exact maximum, only a bound: |dz| <= max|dy| * (1 / keep) * max|act'|, with max|dy| folded from the partials the
// producer of dy left (what the pack kernel does with them), max|gelu'| < 1.13.  The scale is the same power-of-two rule
// (bound * s in [2^14, 2^15)); a bound k bits above the true maximum only raises the format's absolute floor from 2^-40 to
// 2^(k-40) of the tensor maximum (k <= 2 here), the 22 significant bits of every element are untouched.
struct PlaneOut {
  _Float16* p0;              // part 0 of element (r, c) at p0[r * C + c] (C % 32 == 0: no column padding); null = no planes
  long plane_stride;         // elements between the two parts
  const float* in_amax;      // partial maxima of |dy|
  int n_in_amax;
  float bound_factor;
  float* inv_scale;          // {1/s, s} for the consumer kernel
  long rows32;               // rows of a plane; rows .. rows32 - 1 are zeroed here (natural layout)
  int seqT;                  // 0: natural rows; T > 0 (round 6): the k=3 convs' zero-padded image, row (b, t) at b * (T + 2) + 1 + t
  long rows_out;             // image: plane rows in all (vilco_tap_plane_rows); the pad rows of every sequence and the slack are zeroed here
};

__device__ __forceinline__ long plane_row(const PlaneOut& po, long r) {
  return po.seqT ? (r / po.seqT) * (po.seqT + 2) + 1 + r % po.seqT : r;
}
// the k-th zero row of the image (k = 0 .. 2 nseq - 1: the pad rows around the sequences; then the slack below the last one), or -1
__device__ __forceinline__ long plane_zero_row(const PlaneOut& po, long rows, long k) {
  if (!po.seqT) return rows + k < po.rows32 ? rows + k : -1;
  const long nseq = rows / po.seqT;
  if (k < 2 * nseq) return (k >> 1) * (po.seqT + 2) + ((k & 1) ? po.seqT + 1 : 0);
  const long rz = nseq * (po.seqT + 2) + (k - 2 * nseq);
  return rz < po.rows_out ? rz : -1;
}

__device__ __forceinline__ float plane_scale_from_bound(const PlaneOut& po, bool writer) {
  __shared__ float red_[EW_THREADS / 64];
  float m = 0.f;
  const int last = po.n_in_amax - 1;
  for (int i = threadIdx.x; i < po.n_in_amax; i += 4 * EW_THREADS) {      // four partials in flight (a clamped repeat changes no maximum)
    const float p0 = po.in_amax[i], p1 = po.in_amax[i + EW_THREADS < last ? i + EW_THREADS : last],
                p2 = po.in_amax[i + 2 * EW_THREADS < last ? i + 2 * EW_THREADS : last],
                p3 = po.in_amax[i + 3 * EW_THREADS < last ? i + 3 * EW_THREADS : last];
    m = fmaxf(fmaxf(m, fmaxf(p0, p1)), fmaxf(p2, p3));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red_[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red_[0], red_[1]), fmaxf(red_[2], red_[3])) * po.bound_factor;
  int e = (int)((__float_as_uint(m) >> 23) & 0xff);
  if (e < 15) e = 15;                                     // tiny / zero tensors: any scale works
  if (e > 250) e = 250;                                   // inf input: the result is garbage either way
  if (writer) { po.inv_scale[0] = __uint_as_float((unsigned)(e - 14) << 23); po.inv_scale[1] = __uint_as_float((unsigned)(268 - e) << 23); }
  return __uint_as_float((unsigned)(268 - e) << 23);
}

// dz = dy * act'(aux) * rowmask ; partial dbias -> ws[row chunk][C]   (same 2-D decomposition)
// PLANES: dz is (also, or only: dz may be null) written as the fp16 x2 operand planes of its consumers
template <bool PLANES>
__global__ __launch_bounds__(EW_THREADS) void act_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ aux, float* __restrict__ dz,
    float* __restrict__ ws, int act, const int* __restrict__ len, int T, long rows, int C,
    int rows_per_block, uint32_t drop_thresh, uint32_t drop_seed, float drop_inv_keep, float* __restrict__ dbias,
    unsigned* sync, float* __restrict__ amax_parts, const uint32_t* __restrict__ seed_word, PlaneOut po,
    const float* __restrict__ row_mask) {
  if (drop_thresh) drop_seed = vilco_step_seed(drop_seed, seed_word);
  __shared__ float amax_red[EW_THREADS / 64];
  float amax = 0.f;          // max |dz| of this block: dz goes straight into an operand pack
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  float fs = 0.f;
  if (PLANES) fs = plane_scale_from_bound(po, (blockIdx.x | blockIdx.y) == 0 && threadIdx.x == 0);
  if (c < C) {
    float acc = 0.f;
    auto one = [&](long r, float g, float ax) {
      if (drop_thresh) g = vilco_drop_hash(drop_seed, (uint64_t)(r * C + c)) >= drop_thresh ? g * drop_inv_keep : 0.f;
      if (len && (int)(r % T) >= len[r / T]) g = 0.f;
      if (row_mask && row_mask[r] == 0.f) g = 0.f;
      if (act == VILCO_ACT_RELU) g = (ax > 0.f) ? g : 0.f;
      else if (act == VILCO_ACT_GELU) g *= gelu_grad_f(ax);
      if (!PLANES || dz) dz[r * C + c] = g;
      if (PLANES) {
        const float xs = g * fs;                        // exact (power of two)
        const _Float16 h0 = (_Float16)xs;
        const long pr = plane_row(po, r);
        po.p0[pr * C + c] = h0;
        po.p0[po.plane_stride + pr * C + c] = (_Float16)(xs - (float)h0);
      }
      acc += g;
      amax = fmaxf(amax, fabsf(g));
    };
    const bool has_aux = act == VILCO_ACT_RELU || act == VILCO_ACT_GELU;
    long r = r0;
    for (; r + 4 <= r1; r += 4) {          // four rows' loads in flight (see scale_add_bwd_kernel)
      float g[4], ax[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] = dy[(r + u) * C + c];
      if (has_aux) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ax[u] = aux[(r + u) * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(r + u, g[u], ax[u]);
    }
    for (; r < r1; ++r) one(r, dy[r * C + c], has_aux ? aux[r * C + c] : 0.f);
    if (ws) vilco_st_agent(ws + (long)blockIdx.x * C + c, acc);
    if (PLANES && blockIdx.x == gridDim.x - 1)          // the planes' zero rows (a k-major reader contracts over them)
      for (long k = 0;; ++k) {
        const long rz = plane_zero_row(po, rows, k);
        if (rz < 0) break;
        po.p0[rz * C + c] = (_Float16)0.f; po.p0[po.plane_stride + rz * C + c] = (_Float16)0.f;
      }
  }
  if (amax_parts) {
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0)
      amax_parts[blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(amax_red[0], amax_red[1]), fmaxf(amax_red[2], amax_red[3]));
  }
  if (sync) vilco_finish_colsum(ws, dbias, nullptr, (int)gridDim.x, C, C, sync, blockIdx.y * gridDim.x + blockIdx.x,
                                gridDim.x * gridDim.y);
}

// The same op, four columns per lane (C % 4 == 0, 16-byte aligned rows): a wave covers 256 columns of one row with one
// 16-byte load per lane, the block's four waves take every fourth row of the block's row chunk, four rows of a wave in flight
// (8 x 16 bytes per lane against the scalar kernel's 8 x 4).  Column sums: per-wave partials, combined through LDS in wave
// order.  Planes go out as 8-byte stores (512 contiguous bytes per wave and row).
template <bool PLANES>
__global__ __launch_bounds__(EW_THREADS) void act_bwd_vec_kernel(
    const float* __restrict__ dy, const float* __restrict__ aux, float* __restrict__ dz,
    float* __restrict__ ws, int act, const int* __restrict__ len, int T, long rows, int C,
    int rows_per_block, uint32_t drop_thresh, uint32_t drop_seed, float drop_inv_keep, float* __restrict__ dbias,
    unsigned* sync, float* __restrict__ amax_parts, const uint32_t* __restrict__ seed_word, PlaneOut po,
    const float* __restrict__ row_mask) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  if (drop_thresh) drop_seed = vilco_step_seed(drop_seed, seed_word);
  __shared__ float amax_red[EW_THREADS / 64];
  __shared__ float4 acc_red[EW_THREADS / 64][64];
  float amax = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  const int c = blockIdx.y * 256 + lane * 4;
  const bool col_ok = c < C;
  float fs = 0.f;
  if (PLANES) fs = plane_scale_from_bound(po, (blockIdx.x | blockIdx.y) == 0 && threadIdx.x == 0);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool has_aux = act == VILCO_ACT_RELU || act == VILCO_ACT_GELU;
  auto one = [&](long r, float4 g4, float4 a4) {
    float g[4] = {g4.x, g4.y, g4.z, g4.w};
    const float ax[4] = {a4.x, a4.y, a4.z, a4.w};
    const bool row_off = (len && (int)(r % T) >= len[r / T]) || (row_mask && row_mask[r] == 0.f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (drop_thresh) g[e] = vilco_drop_hash(drop_seed, (uint64_t)(r * C + c + e)) >= drop_thresh ? g[e] * drop_inv_keep : 0.f;
      if (row_off) g[e] = 0.f;
      if (act == VILCO_ACT_RELU) g[e] = (ax[e] > 0.f) ? g[e] : 0.f;
      else if (act == VILCO_ACT_GELU) g[e] *= gelu_grad_f(ax[e]);
      amax = fmaxf(amax, fabsf(g[e]));
    }
    if (!PLANES || dz) *reinterpret_cast<float4*>(dz + r * C + c) = make_float4(g[0], g[1], g[2], g[3]);
    if (PLANES) {
      h4 h0, h1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xs = g[e] * fs;                      // exact (power of two)
        h0[e] = (_Float16)xs;
        h1[e] = (_Float16)(xs - (float)h0[e]);
      }
      const long pr = plane_row(po, r);
      *reinterpret_cast<h4*>(po.p0 + pr * C + c) = h0;
      *reinterpret_cast<h4*>(po.p0 + po.plane_stride + pr * C + c) = h1;
    }
    acc.x += g[0]; acc.y += g[1]; acc.z += g[2]; acc.w += g[3];
  };
  if (col_ok) {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    long r = r0 + wave;
    for (; r + 12 < r1; r += 16) {          // four rows of this wave in flight
      float4 g[4], ax[4] = {z4, z4, z4, z4};
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] = *reinterpret_cast<const float4*>(dy + (r + 4 * u) * C + c);
      if (has_aux) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ax[u] = *reinterpret_cast<const float4*>(aux + (r + 4 * u) * C + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(r + 4 * u, g[u], ax[u]);
    }
    for (; r < r1; r += 4)
      one(r, *reinterpret_cast<const float4*>(dy + r * C + c), has_aux ? *reinterpret_cast<const float4*>(aux + r * C + c) : z4);
    if (PLANES && blockIdx.x == gridDim.x - 1) {          // the planes' zero rows (a k-major reader contracts over them)
      const h4 hz = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      for (long k = wave;; k += 4) {
        const long rz = plane_zero_row(po, rows, k);
        if (rz < 0) break;
        *reinterpret_cast<h4*>(po.p0 + rz * C + c) = hz;
        *reinterpret_cast<h4*>(po.p0 + po.plane_stride + rz * C + c) = hz;
      }
    }
  }
  if (ws) {                                  // column sums of the block: the four waves' partials in wave order
    acc_red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && col_ok) {
      float4 t = acc_red[0][lane];
#pragma unroll
      for (int w = 1; w < EW_THREADS / 64; ++w) { const float4 u = acc_red[w][lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
      float* o = ws + (long)blockIdx.x * C + c;
      vilco_st_agent(o, t.x); vilco_st_agent(o + 1, t.y); vilco_st_agent(o + 2, t.z); vilco_st_agent(o + 3, t.w);
    }
  }
  if (amax_parts) {
    amax = wave_max(amax);
    if (lane == 0) amax_red[wave] = amax;
    __syncthreads();
    if (threadIdx.x == 0)
      amax_parts[blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(amax_red[0], amax_red[1]), fmaxf(amax_red[2], amax_red[3]));
  }
  if (sync) vilco_finish_colsum(ws, dbias, nullptr, (int)gridDim.x, C, C, sync, blockIdx.y * gridDim.x + blockIdx.x,
                                gridDim.x * gridDim.y);
}

__global__ __launch_bounds__(EW_THREADS) void colsum_partial_kernel(const float* __restrict__ x,
                                                                    float* __restrict__ ws, long rows,
                                                                    int C, int rows_per_block, float* __restrict__ out,
                                                                    unsigned* sync) {
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c < C) {
    float acc = 0.f;
    long r = r0;
    for (; r + 8 <= r1; r += 8) {          // eight loads in flight; the sum keeps the row order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(r + u) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; r < r1; ++r) acc += x[r * C + c];
    vilco_st_agent(ws + (long)blockIdx.x * C + c, acc);
  }
  if (sync) vilco_finish_colsum(ws, out, nullptr, (int)gridDim.x, C, C, sync, blockIdx.y * gridDim.x + blockIdx.x,
                                gridDim.x * gridDim.y);
}

__global__ __launch_bounds__(EW_THREADS) void mask_rows_kernel(float* __restrict__ x,
                                                               const int* __restrict__ len, int B, int T,
                                                               int C) {
  const long total = (long)B * T * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long bt = i / C;
    if ((int)(bt % T) >= len[bt / T]) x[i] = 0.f;
  }
}

__global__ __launch_bounds__(EW_THREADS) void add_pe_kernel(float* __restrict__ out,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ pe,
                                                            const int* __restrict__ len, int B, int T,
                                                            int C) {
  const long total = (long)B * T * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long bt = i / C;
    const int t = (int)(bt % T), b = (int)(bt / T);
    out[i] = x[i] + ((t < len[b]) ? pe[(long)t * C + c] : 0.f);
  }
}

// in[z][R][S] -> out[z][S][R]; 32x32 tiles through LDS (padded), coalesced on both sides
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ in,
                                                          float* __restrict__ out, int R, int S) {
  __shared__ float tile[32][33];
  const int z = blockIdx.z;
  const float* src = in + (long)z * R * S;
  float* dst = out + (long)z * R * S;
  const int s0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const int r = r0 + ty + k, sc = s0 + tx;
    if (r < R && sc < S) tile[ty + k][tx] = src[(long)r * S + sc];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const int sc = s0 + ty + k, r = r0 + tx;
    if (r < R && sc < S) dst[(long)sc * R + r] = tile[tx][ty + k];
  }
}

__global__ __launch_bounds__(EW_THREADS) void permute3_kernel(const float* __restrict__ in,
                                                              float* __restrict__ out, int d0, int d1,
                                                              int d2, long off, long s0, long s1, long s2) {
  const long total = (long)d0 * d1 * d2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % d2);
    const long ij = i / d2;
    const int j = (int)(ij % d1), ii = (int)(ij / d1);
    out[i] = in[off + ii * s0 + j * s1 + k * s2];
  }
}

int col_blocks(long rows) {
  long b = (rows + 15) / 16;
  if (b > 128) b = 128;
  if (b < 1) b = 1;
  return (int)b;
}

// row blocks of the 2-D (row chunk, 256-column group) reductions: as col_blocks, but never more than
// VILCO_SYNC_MAX_BLOCKS blocks in total, so that the in-launch finish (grid barrier) is safe
int col_blocks_sync(long rows, int C) {
  const int gy = (C + EW_THREADS - 1) / EW_THREADS;
  int nb = col_blocks(rows);
  if ((long)nb * gy > VILCO_SYNC_MAX_BLOCKS) nb = VILCO_SYNC_MAX_BLOCKS / gy;
  return nb < 1 ? 1 : nb;
}

}  // namespace

extern "C" int vilco_softmax_fwd(float* s, const int32_t* kv_len, int32_t B, int32_t H, int32_t Tq,
                                 int32_t Tk, int32_t mode, void* stream) {
  if (!s || B < 0 || H < 0 || Tq < 0 || Tk < 0 || mode < 0 || mode > 2) return VILCO_ERR_BADARG;
  if (mode != 2 && !kv_len) return VILCO_ERR_BADARG;
  const long nrows = (long)B * H * Tq;
  if (nrows == 0 || Tk == 0) return VILCO_OK;
  long blocks = (nrows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((int)blocks), dim3(EW_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), s, kv_len, B, H, Tq, Tk, mode);
  return vilco_launch_status();
}

extern "C" int vilco_softmax_bwd(float* dp, const float* p, int32_t B, int32_t H, int32_t Tq,
                                 int32_t Tk, void* stream) {
  if (!dp || !p || B < 0 || H < 0 || Tq < 0 || Tk < 0) return VILCO_ERR_BADARG;
  const long nrows = (long)B * H * Tq;
  if (nrows == 0 || Tk == 0) return VILCO_OK;
  long blocks = (nrows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((int)blocks), dim3(EW_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), dp, p, nrows, Tk);
  return vilco_launch_status();
}

extern "C" int vilco_relshift_add(float* s, const float* bd, float scale, int32_t B, int32_t H,
                                  int32_t T, void* stream) {
  if (!s || !bd || B < 0 || H < 0 || T < 0) return VILCO_ERR_BADARG;
  const long nrows = (long)B * H * T;
  if (nrows == 0) return VILCO_OK;
  hipLaunchKernelGGL(relshift_add_kernel, dim3(ew_grid(nrows * T)), dim3(EW_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), s, bd, scale, nrows, T);
  return vilco_launch_status();
}

extern "C" int vilco_relshift_bwd(const float* ds, float* dbd, float scale, int32_t B, int32_t H,
                                  int32_t T, void* stream) {
  if (!ds || !dbd || B < 0 || H < 0 || T < 0) return VILCO_ERR_BADARG;
  const long nrows = (long)B * H * T;
  if (nrows == 0) return VILCO_OK;
  hipLaunchKernelGGL(relshift_bwd_kernel, dim3(ew_grid(nrows * 2 * T)), dim3(EW_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), ds, dbd, scale, nrows, T);
  return vilco_launch_status();
}

extern "C" int vilco_scale_add_fwd(float* out, const float* a, const float* bval,
                                   const float* colscale, const float* rowscale, const int32_t* len,
                                   int32_t mask_a, int32_t B, int32_t T, int32_t C, void* stream) {
  if (!out || !bval || B < 0 || T < 0 || C <= 0) return VILCO_ERR_BADARG;
  if ((long)B * T == 0) return VILCO_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if ((C % 4) == 0)
    hipLaunchKernelGGL((scale_add_fwd_kernel<true>), dim3(ew_grid((long)B * T * C / 4)), dim3(EW_THREADS), 0,
                       s, out, a, bval, colscale, rowscale, len, mask_a, B, T, C);
  else
    hipLaunchKernelGGL((scale_add_fwd_kernel<false>), dim3(ew_grid((long)B * T * C)), dim3(EW_THREADS), 0, s,
                       out, a, bval, colscale, rowscale, len, mask_a, B, T, C);
  return vilco_launch_status();
}

extern "C" size_t vilco_colsum_workspace(int64_t rows, int32_t C) {
  return (size_t)col_blocks(rows) * (size_t)C * sizeof(float);
}

extern "C" int vilco_scale_add_bwd(const float* dout, const float* bval, const float* colscale,
                                   const float* rowscale, const int32_t* len, int32_t mask_a,
                                   float* da, float* db, float* dcolscale, int32_t B, int32_t T,
                                   int32_t C, void* workspace, size_t workspace_bytes, void* stream) {
  return vilco_scale_add_bwd_amax(dout, bval, colscale, rowscale, len, mask_a, da, db, dcolscale, B, T, C, workspace, workspace_bytes,
                                  nullptr, nullptr, stream);
}

extern "C" int vilco_scale_add_bwd_amax(const float* dout, const float* bval, const float* colscale,
                                        const float* rowscale, const int32_t* len, int32_t mask_a,
                                        float* da, float* db, float* dcolscale, int32_t B, int32_t T,
                                        int32_t C, void* workspace, size_t workspace_bytes, float* db_amax_parts,
                                        int32_t* n_parts, void* stream) {
  if (n_parts) *n_parts = 0;
  if (!dout || B < 0 || T < 0 || C <= 0) return VILCO_ERR_BADARG;
  if (dcolscale && !bval) return VILCO_ERR_BADARG;
  const long rows = (long)B * T;
  if (rows == 0) return VILCO_OK;
  if (dcolscale && (!workspace || workspace_bytes < vilco_colsum_workspace(rows, C))) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nb = col_blocks_sync(rows, C);
  const int rpb = (int)((rows + nb - 1) / nb);
  float* ws = dcolscale ? reinterpret_cast<float*>(workspace) : nullptr;
  unsigned* sync = (dcolscale && C <= 256 * VILCO_SYNC_MAX_BLOCKS) ? vilco_sync_counter(s, VILCO_SITE_COLSUM) : nullptr;
  const bool emit = db && db_amax_parts && n_parts;
  if (emit) *n_parts = nb * ((C + EW_THREADS - 1) / EW_THREADS);
  hipLaunchKernelGGL(scale_add_bwd_kernel, dim3(nb, (C + EW_THREADS - 1) / EW_THREADS), dim3(EW_THREADS), 0, s, dout, bval, colscale, rowscale,
                     len, mask_a, da, db, ws, B, T, C, rpb, dcolscale, sync, emit ? db_amax_parts : nullptr);
  if (dcolscale && !sync) vilco_reduce_rows(ws, dcolscale, nullptr, nb, C, C, s);
  return vilco_launch_status();
}

// y = keep ? x / (1-p) : 0   (x null: the mask factors themselves, for tests);  in place allowed
__global__ __launch_bounds__(EW_THREADS) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                             uint32_t thresh, float inv_keep, uint32_t seed, uint64_t offset,
                                                             const uint32_t* __restrict__ seed_word) {
  seed = vilco_step_seed(seed, seed_word);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const bool keep = vilco_drop_hash(seed, offset + (uint64_t)i) >= thresh;
    y[i] = keep ? (x ? x[i] : 1.f) * inv_keep : 0.f;
  }
}

extern "C" int vilco_dropout(const float* x, float* y, int64_t n, float p, uint32_t seed, uint64_t offset, void* stream) {
  if (!y || n < 0 || !(p >= 0.f) || p >= 1.f) return VILCO_ERR_BADARG;
  if (n == 0) return VILCO_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0, reinterpret_cast<hipStream_t>(stream), x, y,
                     (long)n, vilco_drop_threshold_host(p), 1.f / (1.f - p), seed, offset, vilco_seed_word_dev());
  return vilco_launch_status();
}

// the mask factors (0 or 1/(1-p)) of the attention-probability dropout of vilco_attn_* (common.h: vilco_attn_drop_*), for tests
__global__ __launch_bounds__(EW_THREADS) void attn_dropout_mask_kernel(float* __restrict__ y, long rows, int cols, uint32_t thresh,
                                                                       float inv_keep, uint32_t seed, const uint32_t* __restrict__ seed_word) {
  seed = vilco_step_seed(seed, seed_word);
  const long n = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols;
    const bool keep = vilco_attn_drop_keep(vilco_attn_drop_row(seed, (uint64_t)r), (uint32_t)(i - r * cols), thresh);
    y[i] = keep ? inv_keep : 0.f;
  }
}

extern "C" int vilco_attn_dropout_mask(float* y, int64_t rows, int32_t cols, float p, uint32_t seed, void* stream) {
  if (!y || rows < 0 || cols < 0 || !(p >= 0.f) || p >= 1.f) return VILCO_ERR_BADARG;
  if (rows == 0 || cols == 0) return VILCO_OK;
  hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(ew_grid(rows * cols)), dim3(EW_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                     y, (long)rows, (int)cols, vilco_drop_threshold_host(p), 1.f / (1.f - p), seed, vilco_seed_word_dev());
  return vilco_launch_status();
}

extern "C" int vilco_axpby(float* out, const float* a, const float* b, float alpha, float beta,
                           int64_t n, void* stream) {
  if (!out || !a || n < 0) return VILCO_ERR_BADARG;
  if (n == 0) return VILCO_OK;
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), out, a, b, alpha, beta, (long)n);
  return vilco_launch_status();
}

extern "C" int vilco_act_bwd(const float* dy, const float* aux, float* dz, float* dbias, int32_t act,
                             const int32_t* len, int32_t T, int64_t rows, int32_t C, float drop_p,
                             uint32_t drop_seed, void* workspace, size_t workspace_bytes, void* stream) {
  return vilco_act_bwd_amax(dy, aux, dz, dbias, act, len, T, rows, C, drop_p, drop_seed, workspace, workspace_bytes, nullptr,
                            nullptr, stream);
}

extern "C" int vilco_act_bwd_amax(const float* dy, const float* aux, float* dz, float* dbias, int32_t act,
                                  const int32_t* len, int32_t T, int64_t rows, int32_t C, float drop_p,
                                  uint32_t drop_seed, void* workspace, size_t workspace_bytes, float* amax_parts,
                                  int32_t* n_parts, void* stream) {
  return vilco_act_bwd_planes(dy, aux, dz, dbias, act, len, T, rows, C, drop_p, drop_seed, workspace, workspace_bytes, amax_parts,
                              n_parts, nullptr, 0, nullptr, 0, nullptr, stream);
}

extern "C" int vilco_act_bwd_planes(const float* dy, const float* aux, float* dz, float* dbias, int32_t act,
                                    const int32_t* len, int32_t T, int64_t rows, int32_t C, float drop_p,
                                    uint32_t drop_seed, void* workspace, size_t workspace_bytes, float* amax_parts,
                                    int32_t* n_parts, const float* dy_amax, int32_t n_dy_amax, void* planes, size_t planes_bytes,
                                    const float* row_mask, void* stream) {
  return vilco_act_bwd_planes_seq(dy, aux, dz, dbias, act, len, T, rows, C, drop_p, drop_seed, workspace, workspace_bytes, amax_parts,
                                  n_parts, dy_amax, n_dy_amax, planes, planes_bytes, 0, row_mask, stream);
}

extern "C" size_t vilco_act_bwd_planes_bytes(int64_t rows, int32_t C, int32_t seq_len) {
  if (rows < 0 || C <= 0 || seq_len < 0) return 0;
  if (seq_len > 0) return (size_t)(VILCO_PACK_HDR + (vilco_tap_plane_rows(rows / seq_len, seq_len) * C + 7) / 8 * 8 * 4);
  const long rows32 = (rows + 31) / 32 * 32;
  return (size_t)(VILCO_PACK_HDR + (rows32 > 0 ? rows32 : 32) * (long)C * 4);
}

extern "C" int vilco_act_bwd_planes_seq(const float* dy, const float* aux, float* dz, float* dbias, int32_t act,
                                        const int32_t* len, int32_t T, int64_t rows, int32_t C, float drop_p,
                                        uint32_t drop_seed, void* workspace, size_t workspace_bytes, float* amax_parts,
                                        int32_t* n_parts, const float* dy_amax, int32_t n_dy_amax, void* planes,
                                        size_t planes_bytes, int32_t seq_len, const float* row_mask, void* stream) {
  if (n_parts) *n_parts = 0;
  if (!(drop_p >= 0.f) || drop_p >= 1.f) return VILCO_ERR_BADARG;
  if (!dy || (!dz && !planes) || rows < 0 || C <= 0 || act < 0 || act > 2 || seq_len < 0) return VILCO_ERR_BADARG;
  PlaneOut po = {nullptr, 0, nullptr, 0, 1.f, nullptr, 0, 0, 0};
  if (planes && seq_len > 0) {
    // the k=3 convs' zero-padded per-sequence image (vilco_pack_item.seq_len): C % 8 == 0, whole sequences
    if (!dy_amax || n_dy_amax <= 0 || (C % 8) != 0 || (rows % seq_len) != 0 || !vilco_aligned(planes, 256)) return VILCO_ERR_BADARG;
    if (planes_bytes < vilco_act_bwd_planes_bytes(rows, C, seq_len)) return VILCO_ERR_WORKSPACE;
    unsigned char* u = reinterpret_cast<unsigned char*>(planes);
    po.p0 = reinterpret_cast<_Float16*>(u + VILCO_PACK_HDR);
    po.rows_out = vilco_tap_plane_rows(rows / seq_len, seq_len);
    po.plane_stride = (po.rows_out * C + 7) / 8 * 8;
    po.seqT = seq_len;
    po.in_amax = dy_amax; po.n_in_amax = n_dy_amax;
    po.bound_factor = (1.f / (1.f - drop_p)) * (act == VILCO_ACT_GELU ? 1.13f : 1.f);
    po.inv_scale = reinterpret_cast<float*>(u) + VILCO_AMAX_MAX_BLOCKS;
    po.rows32 = po.rows_out;
  } else if (planes) {
    // the planes of dz in vilco_pack's layout (precision 3, [rows][C]); needs the partial maxima of |dy| for the scale bound
    if (!dy_amax || n_dy_amax <= 0 || (C % 32) != 0 || !vilco_aligned(planes, 256)) return VILCO_ERR_BADARG;
    const long rows32 = (rows + 31) / 32 * 32;
    if (planes_bytes < (size_t)(VILCO_PACK_HDR + (rows32 > 0 ? rows32 : 32) * (long)C * 4)) return VILCO_ERR_WORKSPACE;
    unsigned char* u = reinterpret_cast<unsigned char*>(planes);
    po.p0 = reinterpret_cast<_Float16*>(u + VILCO_PACK_HDR);
    po.plane_stride = (rows32 > 0 ? rows32 : 32) * (long)C;
    po.in_amax = dy_amax; po.n_in_amax = n_dy_amax;
    po.bound_factor = (1.f / (1.f - drop_p)) * (act == VILCO_ACT_GELU ? 1.13f : 1.f);
    po.inv_scale = reinterpret_cast<float*>(u) + VILCO_AMAX_MAX_BLOCKS;
    po.rows32 = rows32;
  }
  if (act != VILCO_ACT_NONE && !aux) return VILCO_ERR_BADARG;
  if (len && T <= 0) return VILCO_ERR_BADARG;
  if (rows == 0) return VILCO_OK;
  if (dbias && (!workspace || workspace_bytes < vilco_colsum_workspace(rows, C))) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nb = col_blocks_sync(rows, C);
  const int rpb = (int)((rows + nb - 1) / nb);
  float* ws = dbias ? reinterpret_cast<float*>(workspace) : nullptr;
  unsigned* sync = (dbias && C <= 256 * VILCO_SYNC_MAX_BLOCKS) ? vilco_sync_counter(s, VILCO_SITE_COLSUM) : nullptr;
  const bool emit = amax_parts && n_parts;
  if (emit) *n_parts = nb * ((C + EW_THREADS - 1) / EW_THREADS);
  static const bool vec_on = [] { const char* e = getenv("VILCO_ACT_BWD_VEC"); return !(e && e[0] == '0'); }();
  const bool vec = vec_on && (C % 4) == 0 && vilco_aligned(dy, 16) && vilco_aligned(aux, 16) && vilco_aligned(dz, 16);
  const dim3 grid(nb, (C + EW_THREADS - 1) / EW_THREADS);
#define ACT_BWD_LAUNCH(K)                                                                                                        \
  hipLaunchKernelGGL(K, grid, dim3(EW_THREADS), 0, s, dy, aux, dz, ws, act, len, T, (long)rows, C, rpb,                          \
                     vilco_drop_threshold_host(drop_p), drop_seed, 1.f / (1.f - drop_p), dbias, sync,                            \
                     emit ? amax_parts : nullptr, vilco_seed_word_dev(), po, row_mask)
  if (planes) { if (vec) ACT_BWD_LAUNCH(act_bwd_vec_kernel<true>); else ACT_BWD_LAUNCH(act_bwd_kernel<true>); }
  else { if (vec) ACT_BWD_LAUNCH(act_bwd_vec_kernel<false>); else ACT_BWD_LAUNCH(act_bwd_kernel<false>); }
#undef ACT_BWD_LAUNCH
  if (dbias && !sync) vilco_reduce_rows(ws, dbias, nullptr, nb, C, C, s);
  return vilco_launch_status();
}

extern "C" int vilco_colsum(const float* x, float* out, int64_t rows, int32_t C, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (!x || !out || rows < 0 || C <= 0) return VILCO_ERR_BADARG;
  if (!workspace || workspace_bytes < vilco_colsum_workspace(rows, C)) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (rows == 0) return hipMemsetAsync(out, 0, sizeof(float) * C, s) == hipSuccess ? VILCO_OK : VILCO_ERR_LAUNCH;
  const int nb = col_blocks_sync(rows, C);
  const int rpb = (int)((rows + nb - 1) / nb);
  float* ws = reinterpret_cast<float*>(workspace);
  unsigned* sync = C <= 256 * VILCO_SYNC_MAX_BLOCKS ? vilco_sync_counter(s, VILCO_SITE_COLSUM) : nullptr;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb, (C + EW_THREADS - 1) / EW_THREADS), dim3(EW_THREADS), 0, s, x, ws, (long)rows, C, rpb,
                     out, sync);
  if (!sync) vilco_reduce_rows(ws, out, nullptr, nb, C, C, s);
  return vilco_launch_status();
}

extern "C" int vilco_mask_rows(float* x, const int32_t* len, int32_t B, int32_t T, int32_t C,
                               void* stream) {
  if (!x || !len || B < 0 || T < 0 || C <= 0) return VILCO_ERR_BADARG;
  if ((long)B * T == 0) return VILCO_OK;
  hipLaunchKernelGGL(mask_rows_kernel, dim3(ew_grid((long)B * T * C)), dim3(EW_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), x, len, B, T, C);
  return vilco_launch_status();
}

extern "C" int vilco_add_pe(float* out, const float* x, const float* pe, const int32_t* len, int32_t B,
                            int32_t T, int32_t C, void* stream) {
  if (!out || !x || !pe || !len || B < 0 || T < 0 || C <= 0) return VILCO_ERR_BADARG;
  if ((long)B * T == 0) return VILCO_OK;
  hipLaunchKernelGGL(add_pe_kernel, dim3(ew_grid((long)B * T * C)), dim3(EW_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), out, x, pe, len, B, T, C);
  return vilco_launch_status();
}

extern "C" int vilco_transpose2d(const float* in, float* out, int32_t batch, int32_t R, int32_t S,
                                 void* stream) {
  if (!in || !out || batch < 0 || R < 0 || S < 0) return VILCO_ERR_BADARG;
  if ((long)batch * R * S == 0) return VILCO_OK;
  if (batch > 65535) return VILCO_ERR_UNSUPPORTED;
  dim3 grid((S + 31) / 32, (R + 31) / 32, batch);
  hipLaunchKernelGGL(transpose2d_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out,
                     R, S);
  return vilco_launch_status();
}

extern "C" int vilco_permute3(const float* in, float* out, int32_t d0, int32_t d1, int32_t d2,
                              int64_t off, int64_t s0, int64_t s1, int64_t s2, void* stream) {
  if (!in || !out || d0 < 0 || d1 < 0 || d2 < 0) return VILCO_ERR_BADARG;
  const long total = (long)d0 * d1 * d2;
  if (total == 0) return VILCO_OK;
  hipLaunchKernelGGL(permute3_kernel, dim3(ew_grid(total)), dim3(EW_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), in, out, d0, d1, d2, (long)off, (long)s0, (long)s1,
                     (long)s2);
  return vilco_launch_status();
}
