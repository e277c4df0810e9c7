// Fused masked multi-head attention for gfx950 (flash-style: scores never leave the CU).
//
//   o[b,i,h,:] = softmax_j( scale * q[b,i,h,:].k[b,j,h,:] + bias[b,h,i,j] + mask ) v[b,j,h,:]
//
// q/k/v/o are token-major fp32 [B,T,C], heads are channel slices of width hd (blocks.py:383-400 moves heads
// to the batch dim with view/transpose; here they are addressed in place).  mask modes as in vilco_softmax_fwd:
// 0 = keys j >= kv_len[b] excluded, 1 = XLNet (-1e30 unless j == i), 2 = none.  bias is optional (XLNet's
// relative-position term, already scaled).  The forward also writes lse[b,h,i] for the backward.
//
// Work decomposition: one 256-thread workgroup per (b, h, 64-query tile); each wave owns 16 queries and
// walks the keys in tiles of 64.  All products run on v_mfma_f32_16x16x32_bf16 with the same split-bf16
// precision modes as the GEMM (NP = 1, 2, 3 parts -> 1, 3, 6 MFMAs; NP = 3 is fp32-equivalent).  The kernels
// work in the "transposed" orientation S^T = K Q^T, O^T = V^T P^T: the MFMA C layout then puts ONE query on a
// lane (col = lane & 15) and 4 consecutive keys / channels in its registers, so the softmax statistics are
// lane-local (two shuffles per reduction), P goes to LDS as packed 8-byte stores, and O leaves as float4.
// K / V tiles are staged global -> registers -> (split) -> LDS; V is transposed in the register stage
// ([d][key] image) because the PV product contracts over keys.  LDS rows are XOR-swizzled per row width so
// every ds_read_b128 fragment read is bank-conflict free.
//
// Backward = two kernels that recompute P from (q, k, lse): attn_bwd_dq (same orientation, one workgroup per
// query tile: dS -> dQ, optional dBias) and attn_bwd_dkdv (one workgroup per key tile, S = Q K^T orientation so
// P^T / dS^T are packed stores: dV^T = dO^T P, dK^T = Q^T dS).  No atomics: results are bitwise reproducible.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int ATT_THREADS = 256;

// chunk swizzle for a row of W bf16 (W/8 chunks of 16 B): conflict-free ds_read_b128 fragment reads
template <int W>
__device__ __forceinline__ int swzc(int row, int chunk) {
  if (W == 32) return chunk ^ ((4 - ((row >> 2) & 3)) & 3);
  if (W == 64) return chunk ^ ((row >> 1) & 7);
  return chunk ^ (row & 15);   // W == 128
}
template <int W>
__device__ __forceinline__ int toff(int row, int chunk) { return row * W + (swzc<W>(row, chunk) << 3); }

template <int NP>
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&part)[3]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)v[e];
    part[0][e] = h;
    if (NP >= 2) {
      const float r1 = v[e] - (float)h;
      const __bf16 m = (__bf16)r1;
      part[1][e] = m;
      if (NP >= 3) part[2][e] = (__bf16)(r1 - (float)m);
    }
  }
}

template <int NP>
__device__ __forceinline__ void split4(const float (&v)[4], bf16x4 (&part)[3]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h = (__bf16)v[e];
    part[0][e] = h;
    if (NP >= 2) {
      const float r1 = v[e] - (float)h;
      const __bf16 m = (__bf16)r1;
      part[1][e] = m;
      if (NP >= 3) part[2][e] = (__bf16)(r1 - (float)m);
    }
  }
}

// natural tile: R rows x W (>= hd, zero padded) from fp32 rows g[(row0+r)*ld + d]; rows >= rows_valid are zero
template <int W, int NP>
__device__ __forceinline__ void load_tile_nat(__bf16* lds, int R, const float* __restrict__ g, long ld, int row0,
                                              int rows_valid, int hd, int tid) {
  constexpr int CPR = W / 8;
  const int chunks = R * CPR;
  for (int id = tid; id < chunks; id += ATT_THREADS) {
    const int row = id / CPR, c = id % CPR;
    const int d0 = c * 8;
    const int grow = row0 + row;
    float v[8];
    if (grow < rows_valid && d0 + 8 <= hd) {
      const float4 a = *reinterpret_cast<const float4*>(g + (long)grow * ld + d0);
      const float4 b = *reinterpret_cast<const float4*>(g + (long)grow * ld + d0 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (grow < rows_valid && d0 + e < hd) ? g[(long)grow * ld + d0 + e] : 0.f;
    }
    bf16x8 part[3];
    split8<NP>(v, part);
#pragma unroll
    for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8*>(lds + q * R * W + toff<W>(row, c)) = part[q];
  }
}

// transposed tile: HDP rows (d) x WK (keys) from fp32 rows g[(key0+k)*ld + d]; patches of 8 keys x 2 d
template <int WK, int NP>
__device__ __forceinline__ void load_tile_tr(__bf16* lds, int HDP, const float* __restrict__ g, long ld, int key0,
                                             int keys_valid, int hd, int tid) {
  const int DP = HDP / 2;
  const int patches = (WK / 8) * DP;
  for (int id = tid; id < patches; id += ATT_THREADS) {
    const int dp = id % DP, kc = id / DP;
    const int d = dp * 2;
    float v0[8], v1[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int key = key0 + kc * 8 + kk;
      float2 t = make_float2(0.f, 0.f);
      if (key < keys_valid) {
        if (d + 2 <= hd) t = *reinterpret_cast<const float2*>(g + (long)key * ld + d);
        else if (d < hd) t.x = g[(long)key * ld + d];
      }
      v0[kk] = t.x;
      v1[kk] = t.y;
    }
    bf16x8 p0[3], p1[3];
    split8<NP>(v0, p0);
    split8<NP>(v1, p1);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      *reinterpret_cast<bf16x8*>(lds + q * HDP * WK + toff<WK>(d, kc)) = p0[q];
      *reinterpret_cast<bf16x8*>(lds + q * HDP * WK + toff<WK>(d + 1, kc)) = p1[q];
    }
  }
}

template <int W>
__device__ __forceinline__ bf16x8 frag(const __bf16* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + toff<W>(row, chunk));
}

// acc += sum over part pairs of A_parts x B_parts (smallest terms first)
template <int NP>
__device__ __forceinline__ f32x4 mfma_parts(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c) {
  if (NP == 3) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
  }
  if (NP >= 2) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
}

struct AttnArgs {
  const float* q; const float* k; const float* v; const float* bias;
  float* o; float* lse;
  const int* kv_len;
  int B, H, Tq, Tk, hd, C;
  float scale;
  int mode;
  // backward
  const float* dout; const float* delta;
  float* dq; float* dk; float* dv; float* dbias;
};

// registers holding the B-operand fragments of this wave's 16 query rows (all k-steps, all parts)
template <int HDP, int NP>
struct QFrag { bf16x8 f[HDP / 32][3]; };

template <int HDP, int NP>
__device__ __forceinline__ void load_qfrag(QFrag<HDP, NP>& qf, const float* __restrict__ g, long ld, int qrow,
                                           int Tq, int hd, int lane) {
  // lane supplies row (lane & 15), k = 8*(lane >> 4) + j of each 32-wide k-step
  const int r = qrow + (lane & 15);
#pragma unroll
  for (int ks = 0; ks < HDP / 32; ++ks) {
    const int d0 = ks * 32 + (lane >> 4) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (r < Tq && d0 + e < hd) ? g[(long)r * ld + d0 + e] : 0.f;
    bf16x8 part[3];
    split8<NP>(v, part);
#pragma unroll
    for (int q = 0; q < NP; ++q) qf.f[ks][q] = part[q];
  }
}

// score of (query i, key j) after scale: add bias, apply mask
__device__ __forceinline__ float mask_score(float s, int i, int j, int len, int Tk, int mode) {
  if (j >= Tk) return -INFINITY;
  if (mode == 0) return j < len ? s : -INFINITY;
  if (mode == 1) return (j >= len && j != i) ? s - 1e30f : s;
  return s;
}

// ------------------------------------------------------------------------------------------ forward
template <int HDP, int NP>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(AttnArgs a) {
  constexpr int BKV = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sK = reinterpret_cast<__bf16*>(smem_raw);          // [NP][64 keys][HDP]
  __bf16* sVt = sK + NP * BKV * HDP;                          // [NP][HDP d][64 keys]
  __bf16* sP = sVt + NP * HDP * BKV;                          // [4 waves][NP][16 q][64 keys]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const long ld = a.C;
  const float* qg = a.q + (long)b * a.Tq * ld + h * a.hd;
  const float* kg = a.k + (long)b * a.Tk * ld + h * a.hd;
  const float* vg = a.v + (long)b * a.Tk * ld + h * a.hd;
  const int q0 = qt * 64 + wave * 16;
  const int qi = q0 + (lane & 15);                 // this lane's query
  const int len = a.mode == 2 ? a.Tk : a.kv_len[b];
  const float* bias = a.bias ? a.bias + ((long)(b * a.H + h) * a.Tq) * a.Tk : nullptr;

  QFrag<HDP, NP> qf;
  load_qfrag<HDP, NP>(qf, qg, ld, q0, a.Tq, a.hd, lane);

  f32x4 oacc[HDP / 16];
#pragma unroll
  for (int i = 0; i < HDP / 16; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  int kend = a.Tk;
  if (a.mode == 0 && len < kend) kend = len;      // tiles entirely beyond kv_len contribute nothing
  const int ntiles = (kend + BKV - 1) / BKV;
  __bf16* myP = sP + wave * NP * 16 * BKV;

  for (int t = 0; t < ntiles; ++t) {
    const int k0 = t * BKV;
    __syncthreads();                                // previous tile fully consumed
    load_tile_nat<HDP, NP>(sK, BKV, kg, ld, k0, a.Tk, a.hd, tid);
    load_tile_tr<BKV, NP>(sVt, HDP, vg, ld, k0, a.Tk, a.hd, tid);
    __syncthreads();

    // S^T[key][q] = K Q^T : 4 m-tiles of 16 keys, this wave's 16 queries
    f32x4 s[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HDP / 32; ++ks) {
        bf16x8 ka[3];
#pragma unroll
        for (int q = 0; q < NP; ++q) ka[q] = frag<HDP>(sK + q * BKV * HDP, mi * 16 + (lane & 15), ks * 4 + (lane >> 4));
        c = mfma_parts<NP>(ka, qf.f[ks], c);
      }
      s[mi] = c;
    }
    // lane holds query qi, keys k0 + mi*16 + 4*(lane>>4) + r
    float tmax = -INFINITY;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int jb = k0 + mi * 16 + (lane >> 4) * 4;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias && qi < a.Tq) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (jb + r < a.Tk) bv[r] = bias[(long)qi * a.Tk + jb + r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = mask_score(s[mi][r] * a.scale + bv[r], qi, jb + r, len, a.Tk, a.mode);
        s[mi][r] = x;
        tmax = fmaxf(tmax, x);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = (m_new == -INFINITY) ? 1.f : expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = (m_new == -INFINITY) ? 0.f : expf(s[mi][r] - m_new);
        psum += p[r];
      }
      bf16x4 pp[3];
      split4<NP>(p, pp);
      // P[q][keys mi*16 + 4*(lane>>4) .. +3]: chunk = 2*mi + (lane>>5), half = (lane>>4)&1
      const int off = toff<BKV>(lane & 15, 2 * mi + (lane >> 5)) + ((lane >> 4) & 1) * 4;
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x4*>(myP + q * 16 * BKV + off) = pp[q];
    }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < HDP / 16; ++i) {
      oacc[i][0] *= alpha; oacc[i][1] *= alpha; oacc[i][2] *= alpha; oacc[i][3] *= alpha;
    }
    __syncthreads();                                // P visible (only this wave reads it; barrier keeps it simple)

    // O^T[d][q] += V^T P^T : m-tiles over d, k-steps over the 64 keys
#pragma unroll
    for (int ks = 0; ks < BKV / 32; ++ks) {
      bf16x8 pb[3];
#pragma unroll
      for (int q = 0; q < NP; ++q) pb[q] = frag<BKV>(myP + q * 16 * BKV, lane & 15, ks * 4 + (lane >> 4));
#pragma unroll
      for (int di = 0; di < HDP / 16; ++di) {
        bf16x8 va[3];
#pragma unroll
        for (int q = 0; q < NP; ++q) va[q] = frag<BKV>(sVt + q * HDP * BKV, di * 16 + (lane & 15), ks * 4 + (lane >> 4));
        oacc[di] = mfma_parts<NP>(va, pb, oacc[di]);
      }
    }
  }

  // finish: lane holds query qi, channels di*16 + 4*(lane>>4) + r
  if (qi < a.Tq) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* og = a.o + ((long)b * a.Tq + qi) * ld + h * a.hd;
#pragma unroll
    for (int di = 0; di < HDP / 16; ++di) {
      const int d = di * 16 + (lane >> 4) * 4;
      if (d + 4 <= a.hd) {
        *reinterpret_cast<float4*>(og + d) = make_float4(oacc[di][0] * inv, oacc[di][1] * inv, oacc[di][2] * inv, oacc[di][3] * inv);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (d + r < a.hd) og[d + r] = oacc[di][r] * inv;
      }
    }
    if (a.lse && (lane >> 4) == 0) a.lse[((long)b * a.H + h) * a.Tq + qi] = m_run + logf(l_run);
  }
}

// delta[b,h,i] = sum_d dout[b,i,h,d] * o[b,i,h,d]
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ dout, const float* __restrict__ o,
                                                         float* __restrict__ delta, int B, int H, int T, int hd, int C) {
  const long total = (long)B * H * T;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const long bh = i / T;
    const int h = (int)(bh % H), b = (int)(bh / H);
    const float* p = dout + ((long)b * T + t) * C + h * hd;
    const float* r = o + ((long)b * T + t) * C + h * hd;
    float s = 0.f;
    for (int d = 0; d < hd; ++d) s += p[d] * r[d];
    delta[i] = s;
  }
}

// ------------------------------------------------------------------------------------------ backward: dQ (+ dBias)
template <int HDP, int NP>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dq_kernel(AttnArgs a) {
  constexpr int BKV = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sK = reinterpret_cast<__bf16*>(smem_raw);          // [NP][64 keys][HDP]
  __bf16* sV = sK + NP * BKV * HDP;                           // [NP][64 keys][HDP]
  __bf16* sKt = sV + NP * BKV * HDP;                          // [NP][HDP d][64 keys]
  __bf16* sS = sKt + NP * HDP * BKV;                          // [4 waves][NP][16 q][64 keys]  (dS)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const long ld = a.C;
  const float* qg = a.q + (long)b * a.Tq * ld + h * a.hd;
  const float* kg = a.k + (long)b * a.Tk * ld + h * a.hd;
  const float* vg = a.v + (long)b * a.Tk * ld + h * a.hd;
  const float* dog = a.dout + (long)b * a.Tq * ld + h * a.hd;
  const int q0 = qt * 64 + wave * 16;
  const int qi = q0 + (lane & 15);
  const int len = a.mode == 2 ? a.Tk : a.kv_len[b];
  const long row_bh = ((long)b * a.H + h) * a.Tq;
  const float* bias = a.bias ? a.bias + row_bh * a.Tk : nullptr;
  float* dbias = a.dbias ? a.dbias + row_bh * a.Tk : nullptr;

  QFrag<HDP, NP> qf, dof;
  load_qfrag<HDP, NP>(qf, qg, ld, q0, a.Tq, a.hd, lane);
  load_qfrag<HDP, NP>(dof, dog, ld, q0, a.Tq, a.hd, lane);
  const float lse = qi < a.Tq ? a.lse[row_bh + qi] : 0.f;
  const float dlt = qi < a.Tq ? a.delta[row_bh + qi] : 0.f;

  f32x4 dqacc[HDP / 16];
#pragma unroll
  for (int i = 0; i < HDP / 16; ++i) dqacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kend = a.Tk;
  if (a.mode == 0 && len < kend) kend = len;
  const int ntiles = (kend + BKV - 1) / BKV;
  __bf16* myS = sS + wave * NP * 16 * BKV;

  for (int t = 0; t < ntiles; ++t) {
    const int k0 = t * BKV;
    __syncthreads();
    load_tile_nat<HDP, NP>(sK, BKV, kg, ld, k0, a.Tk, a.hd, tid);
    load_tile_nat<HDP, NP>(sV, BKV, vg, ld, k0, a.Tk, a.hd, tid);
    load_tile_tr<BKV, NP>(sKt, HDP, kg, ld, k0, a.Tk, a.hd, tid);
    __syncthreads();

#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HDP / 32; ++ks) {
        bf16x8 ka[3], va[3];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          ka[q] = frag<HDP>(sK + q * BKV * HDP, mi * 16 + (lane & 15), ks * 4 + (lane >> 4));
          va[q] = frag<HDP>(sV + q * BKV * HDP, mi * 16 + (lane & 15), ks * 4 + (lane >> 4));
        }
        s = mfma_parts<NP>(ka, qf.f[ks], s);       // S^T  = K Q^T
        dp = mfma_parts<NP>(va, dof.f[ks], dp);    // dP^T = V dO^T
      }
      const int jb = k0 + mi * 16 + (lane >> 4) * 4;
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float bv = 0.f;
        if (bias && qi < a.Tq && jb + r < a.Tk) bv = bias[(long)qi * a.Tk + jb + r];
        const float x = mask_score(s[r] * a.scale + bv, qi, jb + r, len, a.Tk, a.mode);
        const float p = (x == -INFINITY) ? 0.f : expf(x - lse);
        ds[r] = p * (dp[r] - dlt);
      }
      if (dbias && qi < a.Tq) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (jb + r < a.Tk) dbias[(long)qi * a.Tk + jb + r] = ds[r];
      }
      bf16x4 pp[3];
      split4<NP>(ds, pp);
      const int off = toff<BKV>(lane & 15, 2 * mi + (lane >> 5)) + ((lane >> 4) & 1) * 4;
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x4*>(myS + q * 16 * BKV + off) = pp[q];
    }
    __syncthreads();

    // dQ^T[d][q] += K^T dS^T
#pragma unroll
    for (int ks = 0; ks < BKV / 32; ++ks) {
      bf16x8 sb[3];
#pragma unroll
      for (int q = 0; q < NP; ++q) sb[q] = frag<BKV>(myS + q * 16 * BKV, lane & 15, ks * 4 + (lane >> 4));
#pragma unroll
      for (int di = 0; di < HDP / 16; ++di) {
        bf16x8 ka[3];
#pragma unroll
        for (int q = 0; q < NP; ++q) ka[q] = frag<BKV>(sKt + q * HDP * BKV, di * 16 + (lane & 15), ks * 4 + (lane >> 4));
        dqacc[di] = mfma_parts<NP>(ka, sb, dqacc[di]);
      }
    }
  }
  // rows of dbias beyond the visited key tiles (mode 0, keys >= kv_len) are zero
  if (dbias && qi < a.Tq) {
    for (int j = ntiles * BKV + (lane >> 4); j < a.Tk; j += 4) dbias[(long)qi * a.Tk + j] = 0.f;
  }
  if (qi < a.Tq) {
    float* g = a.dq + ((long)b * a.Tq + qi) * ld + h * a.hd;
#pragma unroll
    for (int di = 0; di < HDP / 16; ++di) {
      const int d = di * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) if (d + r < a.hd) g[d + r] = dqacc[di][r] * a.scale;
    }
  }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// one workgroup per (b, h, 64-key tile); inner loop over 32-query tiles.  S = Q K^T orientation: the C layout
// gives each lane one key (col) and 4 consecutive queries (rows), so P^T / dS^T go to LDS as packed stores.
template <int HDP, int NP>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dkdv_kernel(AttnArgs a) {
  constexpr int BKV = 64, BQ = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sK = reinterpret_cast<__bf16*>(smem_raw);          // [NP][64 keys][HDP]
  __bf16* sV = sK + NP * BKV * HDP;                           // [NP][64 keys][HDP]
  __bf16* sQ = sV + NP * BKV * HDP;                           // [NP][32 q][HDP]
  __bf16* sdO = sQ + NP * BQ * HDP;                           // [NP][32 q][HDP]
  __bf16* sQt = sdO + NP * BQ * HDP;                          // [NP][HDP d][32 q]
  __bf16* sdOt = sQt + NP * HDP * BQ;                         // [NP][HDP d][32 q]
  __bf16* sPt = sdOt + NP * HDP * BQ;                         // [NP][64 keys][32 q]
  __bf16* sdSt = sPt + NP * BKV * BQ;                         // [NP][64 keys][32 q]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const long ld = a.C;
  const float* qg = a.q + (long)b * a.Tq * ld + h * a.hd;
  const float* kg = a.k + (long)b * a.Tk * ld + h * a.hd;
  const float* vg = a.v + (long)b * a.Tk * ld + h * a.hd;
  const float* dog = a.dout + (long)b * a.Tq * ld + h * a.hd;
  const int k0 = kt * BKV;
  const int len = a.mode == 2 ? a.Tk : a.kv_len[b];
  const long row_bh = ((long)b * a.H + h) * a.Tq;
  const float* bias = a.bias ? a.bias + row_bh * a.Tk : nullptr;

  load_tile_nat<HDP, NP>(sK, BKV, kg, ld, k0, a.Tk, a.hd, tid);
  load_tile_nat<HDP, NP>(sV, BKV, vg, ld, k0, a.Tk, a.hd, tid);

  // S / dP tiles: wave -> (m-tile of 16 queries = wave & 1, key half = wave >> 1 : 2 n-tiles of 16 keys)
  const int mq = wave & 1, kh = wave >> 1;
  // dV^T / dK^T accumulators: wave owns keys [wave*16, wave*16+16) (n-tile), all d (HDP/16 m-tiles)
  f32x4 dvacc[HDP / 16], dkacc[HDP / 16];
#pragma unroll
  for (int i = 0; i < HDP / 16; ++i) { dvacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dkacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const bool tile_dead = (a.mode == 0 && k0 >= len);      // every key of this tile is masked: grads are zero
  const int nq = tile_dead ? 0 : (a.Tq + BQ - 1) / BQ;
  for (int t = 0; t < nq; ++t) {
    const int q0 = t * BQ;
    __syncthreads();
    load_tile_nat<HDP, NP>(sQ, BQ, qg, ld, q0, a.Tq, a.hd, tid);
    load_tile_nat<HDP, NP>(sdO, BQ, dog, ld, q0, a.Tq, a.hd, tid);
    load_tile_tr<BQ, NP>(sQt, HDP, qg, ld, q0, a.Tq, a.hd, tid);
    load_tile_tr<BQ, NP>(sdOt, HDP, dog, ld, q0, a.Tq, a.hd, tid);
    __syncthreads();

    // lane: key col = lane & 15 (per n-tile), query rows 4*(lane>>4) + r of m-tile mq
    float lse4[4], dl4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = q0 + mq * 16 + (lane >> 4) * 4 + r;
      lse4[r] = qi < a.Tq ? a.lse[row_bh + qi] : 0.f;
      dl4[r] = qi < a.Tq ? a.delta[row_bh + qi] : 0.f;
    }
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
      const int ncol = kh * 32 + nj * 16;                  // key offset inside the tile
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HDP / 32; ++ks) {
        bf16x8 qa[3], kb[3], da[3], vb[3];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          qa[q] = frag<HDP>(sQ + q * BQ * HDP, mq * 16 + (lane & 15), ks * 4 + (lane >> 4));
          da[q] = frag<HDP>(sdO + q * BQ * HDP, mq * 16 + (lane & 15), ks * 4 + (lane >> 4));
          kb[q] = frag<HDP>(sK + q * BKV * HDP, ncol + (lane & 15), ks * 4 + (lane >> 4));
          vb[q] = frag<HDP>(sV + q * BKV * HDP, ncol + (lane & 15), ks * 4 + (lane >> 4));
        }
        s = mfma_parts<NP>(qa, kb, s);        // S  = Q K^T
        dp = mfma_parts<NP>(da, vb, dp);      // dP = dO V^T
      }
      const int j = k0 + ncol + (lane & 15);
      float p[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q0 + mq * 16 + (lane >> 4) * 4 + r;
        float bv = 0.f;
        if (bias && qi < a.Tq && j < a.Tk) bv = bias[(long)qi * a.Tk + j];
        const float x = mask_score(s[r] * a.scale + bv, qi, j, len, a.Tk, a.mode);
        p[r] = (qi < a.Tq && x != -INFINITY) ? expf(x - lse4[r]) : 0.f;
        ds[r] = p[r] * (dp[r] - dl4[r]);
      }
      bf16x4 pp[3], dd[3];
      split4<NP>(p, pp);
      split4<NP>(ds, dd);
      // P^T[key][q .. q+3]: row = ncol + (lane&15), 4 queries at mq*16 + 4*(lane>>4): chunk = mq*2 + (lane>>5)
      const int off = toff<BQ>(ncol + (lane & 15), mq * 2 + (lane >> 5)) + ((lane >> 4) & 1) * 4;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        *reinterpret_cast<bf16x4*>(sPt + q * BKV * BQ + off) = pp[q];
        *reinterpret_cast<bf16x4*>(sdSt + q * BKV * BQ + off) = dd[q];
      }
    }
    __syncthreads();

    // dV^T[d][key] += dO^T[d][q] P^T[key][q]^T ; dK^T[d][key] += Q^T[d][q] dS^T[key][q]^T   (K = 32 queries)
    bf16x8 pb[3], sb[3];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      pb[q] = frag<BQ>(sPt + q * BKV * BQ, wave * 16 + (lane & 15), lane >> 4);
      sb[q] = frag<BQ>(sdSt + q * BKV * BQ, wave * 16 + (lane & 15), lane >> 4);
    }
#pragma unroll
    for (int di = 0; di < HDP / 16; ++di) {
      bf16x8 oa[3], qa[3];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        oa[q] = frag<BQ>(sdOt + q * HDP * BQ, di * 16 + (lane & 15), lane >> 4);
        qa[q] = frag<BQ>(sQt + q * HDP * BQ, di * 16 + (lane & 15), lane >> 4);
      }
      dvacc[di] = mfma_parts<NP>(oa, pb, dvacc[di]);
      dkacc[di] = mfma_parts<NP>(qa, sb, dkacc[di]);
    }
  }

  // lane: key = k0 + wave*16 + (lane & 15), channels di*16 + 4*(lane>>4) + r
  const int key = k0 + wave * 16 + (lane & 15);
  if (key < a.Tk) {
    float* gk = a.dk + ((long)b * a.Tk + key) * ld + h * a.hd;
    float* gv = a.dv + ((long)b * a.Tk + key) * ld + h * a.hd;
#pragma unroll
    for (int di = 0; di < HDP / 16; ++di) {
      const int d = di * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (d + r < a.hd) {
          gk[d + r] = dkacc[di][r] * a.scale;
          gv[d + r] = dvacc[di][r];
        }
      }
    }
  }
}

template <int HDP, int NP>
size_t fwd_lds() { return (size_t)NP * (64 * HDP + HDP * 64 + 4 * 16 * 64) * sizeof(__bf16); }
template <int HDP, int NP>
size_t dq_lds() { return (size_t)NP * (3 * 64 * HDP + 4 * 16 * 64) * sizeof(__bf16); }
template <int HDP, int NP>
size_t dkdv_lds() { return (size_t)NP * (2 * 64 * HDP + 4 * 32 * HDP + 2 * 64 * 32) * sizeof(__bf16); }

template <typename K>
void set_lds(K kernel, size_t bytes) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  (void)hipGetLastError();
}

template <int HDP, int NP>
int launch_fwd(const AttnArgs& a, hipStream_t s) {
  static const bool once = [] {
    set_lds(&attn_fwd_kernel<HDP, NP>, fwd_lds<HDP, NP>());
    return true;
  }();
  (void)once;
  dim3 grid((a.Tq + 63) / 64, a.H, a.B);
  const size_t lds = fwd_lds<HDP, NP>();
  hipLaunchKernelGGL((attn_fwd_kernel<HDP, NP>), grid, dim3(ATT_THREADS), lds, s, a);
  return vilco_launch_status();
}

template <int HDP, int NP>
int launch_bwd(const AttnArgs& a, hipStream_t s) {
  static const bool once = [] {
    set_lds(&attn_bwd_dq_kernel<HDP, NP>, dq_lds<HDP, NP>());
    set_lds(&attn_bwd_dkdv_kernel<HDP, NP>, dkdv_lds<HDP, NP>());
    return true;
  }();
  (void)once;
  dim3 gq((a.Tq + 63) / 64, a.H, a.B), gk((a.Tk + 63) / 64, a.H, a.B);
  const size_t lq = dq_lds<HDP, NP>(), lk = dkdv_lds<HDP, NP>();
  hipLaunchKernelGGL((attn_bwd_dq_kernel<HDP, NP>), gq, dim3(ATT_THREADS), lq, s, a);
  hipLaunchKernelGGL((attn_bwd_dkdv_kernel<HDP, NP>), gk, dim3(ATT_THREADS), lk, s, a);
  return vilco_launch_status();
}

template <int HDP>
int dispatch(const AttnArgs& a, int precision, bool bwd, hipStream_t s) {
  if (precision == 1) return bwd ? launch_bwd<HDP, 1>(a, s) : launch_fwd<HDP, 1>(a, s);
  if (precision == 0) return bwd ? launch_bwd<HDP, 2>(a, s) : launch_fwd<HDP, 2>(a, s);
  return bwd ? launch_bwd<HDP, 3>(a, s) : launch_fwd<HDP, 3>(a, s);
}

int check_common(int B, int H, int Tq, int Tk, int hd, int mode, int precision) {
  if (B < 0 || H <= 0 || Tq < 0 || Tk < 0 || hd <= 0) return VILCO_ERR_BADARG;
  if (mode < 0 || mode > 2 || precision < 0 || precision > 2) return VILCO_ERR_BADARG;
  if (hd > 64) return VILCO_ERR_UNSUPPORTED;          // head dims up to 64 (P: 64, tests: 8, 16)
  return VILCO_OK;
}

}  // namespace

extern "C" int vilco_attn_supported(int32_t hd) { return hd > 0 && hd <= 64; }

extern "C" int vilco_attn_fwd(const float* q, const float* k, const float* v, const float* bias,
                              const int32_t* kv_len, float* o, float* lse, int32_t B, int32_t H, int32_t Tq,
                              int32_t Tk, int32_t hd, float scale, int32_t mode, int32_t precision, void* stream) {
  int rc = check_common(B, H, Tq, Tk, hd, mode, precision);
  if (rc != VILCO_OK) return rc;
  if (!q || !k || !v || !o || !lse) return VILCO_ERR_BADARG;
  if (mode != 2 && !kv_len) return VILCO_ERR_BADARG;
  if (B == 0 || Tq == 0) return VILCO_OK;
  if (Tk == 0) return VILCO_ERR_BADARG;
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.bias = bias; a.o = o; a.lse = lse; a.kv_len = kv_len;
  a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.hd = hd; a.C = H * hd; a.scale = scale; a.mode = mode;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return hd <= 32 ? dispatch<32>(a, precision, false, s) : dispatch<64>(a, precision, false, s);
}

extern "C" size_t vilco_attn_bwd_workspace(int32_t B, int32_t H, int32_t Tq) {
  return (size_t)B * H * (Tq > 0 ? Tq : 1) * sizeof(float);
}

extern "C" int vilco_attn_bwd(const float* q, const float* k, const float* v, const float* bias,
                              const int32_t* kv_len, const float* o, const float* lse, const float* dout,
                              float* dq, float* dk, float* dv, float* dbias, int32_t B, int32_t H, int32_t Tq,
                              int32_t Tk, int32_t hd, float scale, int32_t mode, int32_t precision,
                              void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_common(B, H, Tq, Tk, hd, mode, precision);
  if (rc != VILCO_OK) return rc;
  if (!q || !k || !v || !o || !lse || !dout || !dq || !dk || !dv) return VILCO_ERR_BADARG;
  if (mode != 2 && !kv_len) return VILCO_ERR_BADARG;
  if (B == 0 || Tq == 0) return VILCO_OK;
  if (Tk == 0) return VILCO_ERR_BADARG;
  if (!workspace || workspace_bytes < vilco_attn_bwd_workspace(B, H, Tq)) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* delta = reinterpret_cast<float*>(workspace);
  const long rows = (long)B * H * Tq;
  long blocks = (rows + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((int)blocks), dim3(256), 0, s, dout, o, delta, B, H, Tq, hd, H * hd);
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.bias = bias; a.lse = const_cast<float*>(lse); a.kv_len = kv_len;
  a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.hd = hd; a.C = H * hd; a.scale = scale; a.mode = mode;
  a.dout = dout; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv; a.dbias = dbias;
  return hd <= 32 ? dispatch<32>(a, precision, true, s) : dispatch<64>(a, precision, true, s);
}
