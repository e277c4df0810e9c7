// MFMA GEMM for gfx950: fp32 operands in HBM, bf16 MFMA (16x16x32) with fp32 accumulate.
//
//   C[m][n] = epilogue(alpha * sum_k A(m,k) * B(k,n))
//
// Tile 128x128x32, 256 threads = 4 waves (2x2), each wave owns a 64x64 sub-tile = 4x4 MFMA tiles.
// Operands are staged global -> registers -> (fp32 -> bf16 hi[/lo]) -> LDS, double buffered, one
// barrier per K-step: the global loads of step k+1 are in flight while step k's MFMAs run.
// LDS tiles are always [128 rows][32 k] bf16 with a 16-byte-chunk XOR swizzle that makes the
// ds_read_b128 fragment reads bank-conflict free (see swz()).  Operands whose contiguous dim is
// NOT k ("transposed" operands: activations in dW = dY^T X, weights in dX = dY W) are transposed
// inside the register stage (8(k) x 2(row) patches -> one 16-byte LDS store per row).
//
// precision 0 ("split"): x = hi + lo with hi = bf16(x), lo = bf16(x - hi); a*b ~= ah*bh + ah*bl +
// al*bh (3 MFMAs), relative error ~2^-17.  precision 1: single bf16 pass (2^-9).  precision 2
// ("split3"): three bf16 parts, 6 MFMAs, ~2^-25: numerically an fp32 GEMM at 1/6 of the bf16 MFMA
// rate (still 2.6x the fp32-MFMA peak of gfx950).
//
// Reference arithmetic replaced: see include/vilco_hip.h (vilco_gemm).
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NTHREADS = 256;
constexpr int TILE_ELEMS = 128 * 32;  // bf16 elements per operand tile

// Swizzle: rows are 64 B (4 chunks of 16 B); chunk' = chunk ^ f(row>>2).  With f = {0,3,2,1}
// every ds_read_b128 lane group ({0-3,12-15,20-27}, ...) touches 16 distinct 16-B slots of the
// 256-B bank row (derivation in DESIGN.md "GEMM LDS layout").
__device__ __forceinline__ int swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 32 + ((chunk ^ swz(row)) << 3);
}

struct Operand {
  const float* p;  // batch-offset base pointer
  long ld;
  int rows;  // extent of the non-k dim (M for A, N for B)
  int tap;   // 1: tapped (overlapped-row conv) operand, tapC % 8 == 0;  2: tapped, any tapC (A only)
};

// x = p0 + p1 + p2 (+ ~2^-27 x): p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1); the
// subtractions are exact in fp32.
template <int NP>
__device__ __forceinline__ void splitN(const float (&v)[8], bf16x8 (&part)[3]) {
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

// ---- operand whose contiguous dim is k: element (r,k) at p[r*ld + k] --------------------------
template <bool VEC>
__device__ __forceinline__ void gload_kc(float (&r)[16], const Operand& op, int row0, int k0, int K,
                                         int tapC, int tapT, int tid) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int id = tid + h * NTHREADS;
    const int row = id >> 2, c = id & 3;
    const int grow = row0 + row, gk = k0 + c * 8;
    bool ok = (grow < op.rows) && (gk < K);
    long off = (long)grow * op.ld + gk;
    if (op.tap == 1) {
      // contiguous dim spans taps {t-1, t, t+1}; zero the taps that fall outside the sequence
      const int tap = gk / tapC;
      const int t = grow % tapT;
      if ((tap == 0 && t == 0) || (tap == 2 && t == tapT - 1)) ok = false;
      off -= tapC;
    } else if (op.tap == 2) {
      off -= tapC;
    }
    float* d = &r[h * 8];
    if (op.tap == 2) {
      // fine-grained taps (tapC not a multiple of 8, e.g. the 2- and 22-channel head outputs in
      // their dX pass): the 8-wide chunk can straddle taps, so test every element
      const int t = grow % tapT;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = gk + e;
        const int tap = k / tapC;
        const bool oke = (grow < op.rows) && (k < K) && !((tap == 0 && t == 0) || (tap == 2 && t == tapT - 1));
        d[e] = oke ? op.p[off + e] : 0.f;
      }
    } else if (ok && VEC && gk + 8 <= K) {
      const float4 v0 = *reinterpret_cast<const float4*>(op.p + off);
      const float4 v1 = *reinterpret_cast<const float4*>(op.p + off + 4);
      d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w;
      d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = (ok && gk + e < K) ? op.p[off + e] : 0.f;
    }
  }
}

template <int NP>
__device__ __forceinline__ void lstore_kc(const float (&r)[16], __bf16* tile, int tid) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int id = tid + h * NTHREADS;
    const int row = id >> 2, c = id & 3;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = r[h * 8 + e];
    bf16x8 part[3];
    splitN<NP>(v, part);
    const int o = lds_off(row, c);
#pragma unroll
    for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8*>(tile + q * TILE_ELEMS + o) = part[q];
  }
}

// ---- operand whose contiguous dim is its row index: element (r,k) at p[k*ld + r] --------------
// thread -> patch of 8 k x 2 rows: kg = tid>>6 (k chunk), rg = tid&63 (row pair)
template <bool VEC>
__device__ __forceinline__ void gload_tr(float (&r)[16], const Operand& op, int row0, int k0, int K,
                                         int tapC, int tapT, int tid) {
  const int kg = tid >> 6, rg = tid & 63;
  const int gr = row0 + rg * 2;
  int tap = 0;
  long coff = gr;
  if (op.tap) {
    tap = gr / tapC;  // both rows of the pair are in one tap (tapC even)
    coff -= tapC;
  }
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const int gk = k0 + kg * 8 + kk;
    bool ok = gk < K;
    if (op.tap) {
      const int t = gk % tapT;
      if ((tap == 0 && t == 0) || (tap == 2 && t == tapT - 1)) ok = false;
    }
    const long off = (long)gk * op.ld + coff;
    if (ok && VEC && gr + 2 <= op.rows) {
      const float2 v = *reinterpret_cast<const float2*>(op.p + off);
      r[kk * 2] = v.x;
      r[kk * 2 + 1] = v.y;
    } else {
      r[kk * 2] = (ok && gr < op.rows) ? op.p[off] : 0.f;
      r[kk * 2 + 1] = (ok && gr + 1 < op.rows) ? op.p[off + 1] : 0.f;
    }
  }
}

template <int NP>
__device__ __forceinline__ void lstore_tr(const float (&r)[16], __bf16* tile, int tid) {
  const int kg = tid >> 6, rg = tid & 63;
#pragma unroll
  for (int nn = 0; nn < 2; ++nn) {
    float v[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) v[kk] = r[kk * 2 + nn];
    bf16x8 part[3];
    splitN<NP>(v, part);
    const int o = lds_off(rg * 2 + nn, kg);
#pragma unroll
    for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8*>(tile + q * TILE_ELEMS + o) = part[q];
  }
}

struct Epi {
  float alpha, beta;
  const float* bias;
  float* preact;
  int act;
  const int* row_len;
  int rowT;
  const float* colscale;
  const float* residual;
  int res_masked;
};

struct Args {
  Operand a, b;
  float* c;
  long ldc;
  int M, N, K;
  int batch_inner;
  long sAo, sAi, sBo, sBi, sCo, sCi;
  int tapC, tapT;
  int tiles_n, ntiles;
  Epi e;
};

template <bool A_KC, bool B_KC, int NP, bool VEC>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);
  // layout: [stage 2][operand 2][part NP][TILE_ELEMS]
  auto tile_ptr = [&](int stage, int opnd, int part) {
    return smem + ((stage * 2 + opnd) * NP + part) * TILE_ELEMS;
  };

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
  // contiguous run of tiles (neighbours share the A row panel in its L2).  Bijective for any count.
  int bid = blockIdx.x;
  {
    const int nwg = g.ntiles, q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int z = blockIdx.z, zo = z / g.batch_inner, zi = z % g.batch_inner;
  Operand oa = g.a, ob = g.b;
  oa.p += zo * g.sAo + zi * g.sAi;
  ob.p += zo * g.sBo + zi * g.sBi;
  const long coff = zo * g.sCo + zi * g.sCi;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float ra[16], rb[16];
  const int nk = (g.K + BK - 1) / BK;

  auto gload = [&](int kt) {
    if (A_KC) gload_kc<VEC>(ra, oa, m0, kt * BK, g.K, g.tapC, g.tapT, tid);
    else      gload_tr<VEC>(ra, oa, m0, kt * BK, g.K, g.tapC, g.tapT, tid);
    if (B_KC) gload_kc<VEC>(rb, ob, n0, kt * BK, g.K, g.tapC, g.tapT, tid);
    else      gload_tr<VEC>(rb, ob, n0, kt * BK, g.K, g.tapC, g.tapT, tid);
  };
  auto lstore = [&](int stage) {
    if (A_KC) lstore_kc<NP>(ra, tile_ptr(stage, 0, 0), tid);
    else      lstore_tr<NP>(ra, tile_ptr(stage, 0, 0), tid);
    if (B_KC) lstore_kc<NP>(rb, tile_ptr(stage, 1, 0), tid);
    else      lstore_tr<NP>(rb, tile_ptr(stage, 1, 0), tid);
  };

  gload(0);
  lstore(0);
  __syncthreads();

  const int frow = lane & 15, fchunk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);

    // fragments of every part; products kept (smallest first): NP=2: lh hl hh; NP=3: + mm, hl/lh with
    // the third part.  Dropped terms are <= 2^-17 (NP=2) / 2^-26 (NP=3) relative.
    bf16x8 fa[NP][4], fb[NP][4];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const __bf16* at = tile_ptr(st, 0, q);
      const __bf16* bt = tile_ptr(st, 1, q);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[q][i] = *reinterpret_cast<const bf16x8*>(at + lds_off(wm * 64 + i * 16 + frow, fchunk));
        fb[q][i] = *reinterpret_cast<const bf16x8*>(bt + lds_off(wn * 64 + i * 16 + frow, fchunk));
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 c = acc[i][j];
        if (NP == 3) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2][i], fb[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[2][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[1][j], c, 0, 0, 0);
        }
        if (NP >= 2) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[0][j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[1][j], c, 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[0][j], c, 0, 0, 0);
      }

    if (kt + 1 < nk) lstore(st ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout of mfma_f32_16x16x32: col = lane&15, row = (lane>>4)*4 + reg
  const Epi& e = g.e;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + rr;
      if (m >= g.M) continue;
      bool valid = true;
      if (e.row_len) valid = (m % e.rowT) < e.row_len[m / e.rowT];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + (lane & 15);
        if (n >= g.N) continue;
        const long idx = coff + (long)m * g.ldc + n;
        float v = e.alpha * acc[i][j][rr];
        if (e.bias) v += e.bias[n];
        if (e.preact) e.preact[idx] = v;
        if (e.act == VILCO_ACT_RELU) v = fmaxf(v, 0.f);
        else if (e.act == VILCO_ACT_GELU) v = gelu_f(v);
        if (!valid) v = 0.f;
        if (e.colscale) v *= e.colscale[n];
        if (e.residual && (valid || !e.res_masked)) v += e.residual[idx];
        if (e.beta != 0.f) v += e.beta * g.c[idx];
        g.c[idx] = v;
      }
    }
  }
}

template <bool A_KC, bool B_KC, int NP>
int launch_np(const Args& a, bool vec, dim3 grid, hipStream_t s) {
  const size_t lds = (size_t)2 * 2 * NP * TILE_ELEMS * sizeof(__bf16);   // 32 / 64 / 96 KiB
  static const bool attr_once = [] {
    const int cap = 2 * 2 * NP * TILE_ELEMS * (int)sizeof(__bf16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<A_KC, B_KC, NP, true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<A_KC, B_KC, NP, false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipGetLastError();
    return true;
  }();
  (void)attr_once;
  if (vec) hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, NP, true>), grid, dim3(NTHREADS), lds, s, a);
  else     hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, NP, false>), grid, dim3(NTHREADS), lds, s, a);
  return vilco_launch_status();
}

template <bool A_KC, bool B_KC>
int launch(const Args& a, int precision, bool vec, dim3 grid, hipStream_t s) {
  // precision 0: split-bf16 (2 parts, 3 MFMAs); 1: plain bf16; 2: 3 parts, 6 MFMAs (fp32-equivalent)
  if (precision == 0) return launch_np<A_KC, B_KC, 2>(a, vec, grid, s);
  if (precision == 1) return launch_np<A_KC, B_KC, 1>(a, vec, grid, s);
  return launch_np<A_KC, B_KC, 3>(a, vec, grid, s);
}

}  // namespace

extern "C" int vilco_gemm(const vilco_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->C) return VILCO_ERR_BADARG;
  if (d->M < 0 || d->N < 0 || d->K < 0 || d->batch_outer < 1 || d->batch_inner < 1) return VILCO_ERR_BADARG;
  if (d->M == 0 || d->N == 0) return VILCO_OK;
  if (d->precision < 0 || d->precision > 2) return VILCO_ERR_BADARG;
  if (d->act < 0 || d->act > 2) return VILCO_ERR_BADARG;
  if (d->row_len && d->rowT <= 0) return VILCO_ERR_BADARG;
  if (d->a_kcontig == 0 && d->b_kcontig == 1) return VILCO_ERR_UNSUPPORTED;  // "TT" is never needed
  if (d->tap_operand != VILCO_TAP_NONE) {
    if (d->tapC <= 0 || d->tapT <= 0) return VILCO_ERR_BADARG;
    if (d->tap_operand == VILCO_TAP_B && (d->tapC % 8) != 0) return VILCO_ERR_UNSUPPORTED;
    // tapped operand: its contiguous dim must be the 3*tapC tap span
    if (d->tap_operand == VILCO_TAP_A && !(d->a_kcontig == 1 && d->K == 3 * d->tapC)) return VILCO_ERR_BADARG;
    if (d->tap_operand == VILCO_TAP_B && !(d->b_kcontig == 0 && d->N == 3 * d->tapC)) return VILCO_ERR_BADARG;
    if (d->tap_operand != VILCO_TAP_A && d->tap_operand != VILCO_TAP_B) return VILCO_ERR_BADARG;
  }

  Args a;
  a.a = Operand{d->A, (long)d->lda, d->M, d->tap_operand == VILCO_TAP_A ? ((d->tapC % 8) == 0 ? 1 : 2) : 0};
  a.b = Operand{d->B, (long)d->ldb, d->N, d->tap_operand == VILCO_TAP_B ? 1 : 0};
  a.c = d->C;
  a.ldc = d->ldc;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.batch_inner = d->batch_inner;
  a.sAo = d->sAo; a.sAi = d->sAi; a.sBo = d->sBo; a.sBi = d->sBi; a.sCo = d->sCo; a.sCi = d->sCi;
  a.tapC = d->tapC > 0 ? d->tapC : 1;
  a.tapT = d->tapT > 0 ? d->tapT : 1;
  a.tiles_n = (d->N + BN - 1) / BN;
  a.ntiles = a.tiles_n * ((d->M + BM - 1) / BM);
  a.e = Epi{d->alpha, d->beta, d->bias, d->preact, d->act, d->row_len, d->rowT, d->colscale,
            d->residual, d->res_masked};

  // vector path: k-contiguous operands need 16-B aligned rows, transposed ones 8-B aligned pairs
  auto vec_ok = [&](const float* p, long ld, bool kc, long so, long si) {
    const long q = kc ? 4 : 2;
    return vilco_aligned(p, kc ? 16 : 8) && (ld % q == 0) && (so % q == 0) && (si % q == 0);
  };
  bool vec = vec_ok(d->A, d->lda, d->a_kcontig, d->sAo, d->sAi) &&
             vec_ok(d->B, d->ldb, d->b_kcontig, d->sBo, d->sBi);
  if (d->a_kcontig && (d->K % 4) != 0) vec = false;   // row tails are handled scalar, starts must align
  if (d->tap_operand != VILCO_TAP_NONE && (d->tapC % 4) != 0) vec = false;

  dim3 grid(a.ntiles, 1, d->batch_outer * d->batch_inner);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (d->a_kcontig && d->b_kcontig) return launch<true, true>(a, d->precision, vec, grid, s);
  if (d->a_kcontig && !d->b_kcontig) return launch<true, false>(a, d->precision, vec, grid, s);
  return launch<false, false>(a, d->precision, vec, grid, s);
}
