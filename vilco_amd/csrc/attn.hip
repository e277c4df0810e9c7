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
// q/k/v (and dO in backward) are first split ONCE into bf16 planes by the pack kernels of pack.h -- natural
// [token][d] images and transposed [d][token] images (the PV / dQ / dK / dV products contract over tokens) --
// so the tile loops only copy 16-byte chunks global -> registers -> LDS (next tile prefetched under the current
// tile's MFMAs).  LDS rows are XOR-swizzled per row width so every ds_read_b128 fragment read is conflict free,
// and every fragment address is one per-lane base plus compile-time constants.
//
// Backward = two kernels that recompute P from (q, k, lse): attn_bwd_dq (same orientation, one workgroup per
// query tile: dS -> dQ, optional dBias) and attn_bwd_dkdv (one workgroup per key tile, S = Q K^T orientation so
// P^T / dS^T are packed stores: dV^T = dO^T P, dK^T = Q^T dS).  No atomics: results are bitwise reproducible.
#include <type_traits>
#include "common.h"
#define VILCO_TU "attn"
#include "pack.h"

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

namespace {

constexpr int ATT_THREADS = 256;

// chunk swizzle for a row of W bf16 (W/8 chunks of 16 B): conflict-free ds_read_b128 fragment reads
template <int W>
__device__ __forceinline__ int swzc(int row, int chunk) {
  if (W == 32) return chunk ^ ((4 - ((row >> 2) & 3)) & 3);
  if (W == 64) return chunk ^ ((row >> 1) & 7);
  // W == 160 (head dims 129..160, config W's 144): five 32-element groups, each swizzled like a W == 32 row.  The row
  // stride is 80 dwords = 16 mod 64 banks -- the W == 32 tile's stride -- so a fragment read (16 rows x one group) meets
  // exactly the W == 32 bank pattern, shifted by a constant.
  if (W == 160) return (chunk & ~3) | ((chunk & 3) ^ ((4 - ((row >> 2) & 3)) & 3));
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

// fp16 x2 parts of values the caller has already scaled into fp16 range (pack.h fmt 1): h0 = fp16(v) by the packed
// convert (round to nearest even), h1 = fp16(v - h0) by ONE mixed-precision FMA per value, which reads the fp16 half in
// place, subtracts in fp32 and rounds (instead of convert-back, subtract, convert)
__device__ __forceinline__ void split4h(const float (&v)[4], bf16x4 (&part)[3]) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 h0, h1;
#pragma unroll
  for (int e = 0; e < 4; e += 2) {
    const f16x2 hp = {(_Float16)v[e], (_Float16)v[e + 1]};
    const uint32_t hpu = __builtin_bit_cast(uint32_t, hp);
    uint32_t lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hpu), "v"(v[e]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hpu), "v"(v[e + 1]));
    h0[e >> 1] = hpu;
    h1[e >> 1] = lo;
  }
  part[0] = __builtin_bit_cast(bf16x4, h0);
  part[1] = __builtin_bit_cast(bf16x4, h1);
}

template <int NP, bool F16>
__device__ __forceinline__ void split4s(const float (&v)[4], bf16x4 (&part)[3]) {
  if constexpr (F16) split4h(v, part);
  else split4<NP>(v, part);
}

// ---- operand planes (built by pack.h): part q of batch z starts at p + q*part_stride + z*batch_stride
struct Planes {
  const __bf16* p;
  long part_stride, batch_stride;
  int row_stride;      // elements between rows
  int rows;            // valid rows (tokens for natural planes, hd for transposed planes)
  int cols;            // valid columns (padded: HDP for natural planes, Tp for transposed planes)
};

// staged copy of an R x W tile (all NP parts): gload -> registers, lstore -> swizzled LDS image [part][R][W].
// rows >= pl.rows or columns >= pl.cols read as zero.
template <int W, int R, int NP>
struct TileStage {
  static constexpr int N = (R * (W / 8) + ATT_THREADS - 1) / ATT_THREADS;
  bf16x8 v[NP][N];
};

template <int W, int R, int NP>
__device__ __forceinline__ void gload_tile(TileStage<W, R, NP>& st, const Planes& pl, const __bf16* base, int row0,
                                           int col0, int tid) {
  constexpr int CPR = W / 8;
#pragma unroll
  for (int i = 0; i < TileStage<W, R, NP>::N; ++i) {
    const int id = tid + i * ATT_THREADS;
    const int row = id / CPR, c = id % CPR;
    const bool ok = id < R * CPR && row0 + row < pl.rows && col0 + c * 8 < pl.cols;
    const long off = (long)(row0 + row) * pl.row_stride + col0 + c * 8;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      bf16x8 z = {};
      st.v[q][i] = ok ? *reinterpret_cast<const bf16x8*>(base + q * pl.part_stride + off) : z;
    }
  }
}

template <int W, int R, int NP>
__device__ __forceinline__ void lstore_tile(const TileStage<W, R, NP>& st, __bf16* lds, int tid) {
  constexpr int CPR = W / 8;
#pragma unroll
  for (int i = 0; i < TileStage<W, R, NP>::N; ++i) {
    const int id = tid + i * ATT_THREADS;
    if (id >= R * CPR) continue;
    const int o = toff<W>(id / CPR, id % CPR);
#pragma unroll
    for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8*>(lds + q * R * W + o) = st.v[q][i];
  }
}

// fragment read.  fb = toff<W>(lane & 15, lane >> 4) is computed once per lane; rows are only ever added in
// multiples of 16 (the swizzle is invariant under row += 16) and the second 32-wide k-step of a 64-wide row is
// the XOR of 4 chunks (32 elements), so every address is fb plus compile-time constants.
template <int W>
__device__ __forceinline__ bf16x8 frag(const __bf16* tile, int fb, int row_add, int ks) {
  if (W == 160) return *reinterpret_cast<const bf16x8*>(tile + fb + row_add * W + (ks << 5));   // the group is not swizzled
  return *reinterpret_cast<const bf16x8*>(tile + ((fb + row_add * W) ^ (ks << 5)));
}

// acc += sum over part pairs of A_parts x B_parts (smallest terms first)
template <int NP, bool F16>
__device__ __forceinline__ f32x4 mfma_parts(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c) {
  if constexpr (F16) {
    const f16x8 a0 = __builtin_bit_cast(f16x8, a[0]), a1 = __builtin_bit_cast(f16x8, a[1]);
    const f16x8 b0 = __builtin_bit_cast(f16x8, b[0]), b1 = __builtin_bit_cast(f16x8, b[1]);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c, 0, 0, 0);
  } else {
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
}

// fp16 x2 mode: q/k/v/dO planes hold x * s (per-tensor power of two, pack.h); products computed in-kernel are brought
// into fp16 range by fixed powers of two: P <= 1 is stored as P * 2^15, and dS = P (dP - delta) as dS * 2^-22 sdO sV,
// which is < 2^15 because |dP|, |delta| <= hd * amax(dO) * amax(V) with hd <= 64 and amax * s < 2^15.
constexpr float P_SCALE = 32768.f, P_INV = 1.f / 32768.f, DS_SCALE = 1.f / 4194304.f, DS_INV = 4194304.f;
// the general kernels at head dims 65..128: |dP| <= hd * 2^30 doubles, so dS carries one more power of two (2^-23)
template <int HDP> constexpr float ds_scale() { return HDP > 128 ? 0.25f * DS_SCALE : (HDP > 64 ? 0.5f * DS_SCALE : DS_SCALE); }
template <int HDP> constexpr float ds_inv() { return HDP > 128 ? 4.f * DS_INV : (HDP > 64 ? 2.f * DS_INV : DS_INV); }
struct AttnScales { float iq, sq, ik, sk, iv, sv, ido, sdo; };    // {1/s, s} of q, k, v, dO (pack.h order)

struct AttnArgs {
  const float* q; const float* k; const float* v; const float* bias;
  float* o; float* lse;
  const int* kv_len;
  int B, H, Tq, Tk, hd, C;
  float scale;
  int mode;
  int window;               // mode 4: half-width w of the local attention window
  // backward
  const float* dout; float* delta;     // delta[b,h,i] = dO_i . O_i: written by attn_bwd_dq, read by attn_bwd_dkdv
  const float* o_in;                   // forward output (backward only)
  float* dq; float* dk; float* dv; float* dbias;
  // 16-bit planes (natural [tok][HDP] and transposed [d][Tp]) of k, v, q, dout
  Planes kn, vn, kt, vt, qn, don, qt, dot;
  const AttnScales* sc;   // fp16 x2 mode only
  // dropout on the attention probabilities (modeling_xlnet_x.py:308, blocks.py:226): keep iff hash(seed, (bh*Tq+i)*Tk+j) >= thresh
  uint32_t drop_thresh, drop_seed;
  float drop_inv_keep;
  const uint32_t* seed_word;   // device step word mixed into drop_seed (common.h: vilco_step_seed)
  // optional max|x| partials of the outputs (one per workgroup), left for the operand pack of the next product
  // (vilco_pack_item.amax); written by the hd = 64 fast kernels only (vilco_attn_amax_parts)
  float* am_o; float* am_dq; float* am_dk; float* am_dv; float* am_ds;
  // optional (hd = 64 fast forward kernels): o is also written as the fp16 x2 operand planes of the output projection
  // (pack.h's layout for [B * Tq][C]).  Every output row is a convex combination of rows of v (times 1 / keep under probability
  // dropout), so max|v| -- whose scale this call has just folded for its own V planes -- bounds max|o|: the planes' scale is
  // sv * o_scale_mul, o_scale_mul = 2^-ceil(log2(1 / keep)).
  _Float16* op0; long o_plane_stride; float* o_inv_scale; long o_rows32; float o_scale_mul;
  // XL backward (round 5): dS written as the fp16 x2 operand planes of XLNet's UNSHIFTED [Tq][Tq + Tk] view (pack.h tap mode 4
  // layout: [part][b*H + h][rows32][cols32], element (i, Tq - i + j)) by attn_bwd_dq64_kernel<true> itself -- the very two
  // halves it builds for its own dQ product -- instead of fp32 dS + a pack pass over it.  Null: fp32 dS to `dbias`.
  _Float16* dsp; long ds_plane_stride; long ds_batch_stride; int ds_ld; float* ds_inv_scale;
};

// registers holding the B-operand fragments of this wave's 16 query rows (all k-steps, all parts)
template <int HDP, int NP>
struct QFrag { bf16x8 f[HDP / 32][3]; };

// B fragments straight from a natural plane: lane supplies row (lane & 15), k = ks*32 + 8*(lane >> 4) + j
template <int HDP, int NP>
__device__ __forceinline__ void load_qfrag(QFrag<HDP, NP>& qf, const Planes& pl, const __bf16* base, int qrow, int lane) {
  const int r = qrow + (lane & 15);
#pragma unroll
  for (int ks = 0; ks < HDP / 32; ++ks) {
    const long off = (long)r * pl.row_stride + ks * 32 + (lane >> 4) * 8;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      bf16x8 z = {};
      qf.f[ks][q] = r < pl.rows ? *reinterpret_cast<const bf16x8*>(base + q * pl.part_stride + off) : z;
    }
  }
}

// exp(x) on the hardware exp2 unit (v_exp_f32): one multiply + one transcendental instead of the ~15-instruction
// expf; |relative error| <= ~2^-21 for the |x| <= 40 range of softmax arguments, far inside the 1e-3 budget.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// additive bias of (query i, key j).  mode 3: `bias` holds XLNet's UNSHIFTED position scores bd[b,h,i,p] = qr_i . kr_p,
// p in [0, Tq+Tk); rel_shift_bnij (modeling_xlnet_x.py:204-214) picks p = Tq - i + j, and the 1/sqrt(d) scale of the
// score applies to it: the shifted [Tq,Tk] bias tensor is never materialised.
__device__ __forceinline__ float bias_at(const float* bias, int i, int j, int ld, const AttnArgs& a) {
  if (a.mode == 3) return a.scale * bias[(long)i * ld + (a.Tq - i + j)];
  return bias[(long)i * ld + j];
}

// 4 consecutive keys jb..jb+3 of query row i: ONE 16-byte access per lane (the address is only 4-byte aligned in mode 3;
// gfx950 global loads take that) instead of four scattered dword accesses -- the bias / dS traffic of the XLNet layer is
// 680 MB per pass and was 3/4 of that layer's attention time as scalar loads.
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };
__device__ __forceinline__ void bias4(float (&bv)[4], const float* bias, int i, int jb, int ld, const AttnArgs& a) {
  if (jb + 4 <= a.Tk) {
    const float* p = a.mode == 3 ? bias + (long)i * ld + (a.Tq - i + jb) : bias + (long)i * ld + jb;
    const f4u t = *reinterpret_cast<const f4u*>(p);
    const float sc = a.mode == 3 ? a.scale : 1.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = sc * t.v[r];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = jb + r < a.Tk ? bias_at(bias, i, jb + r, ld, a) : 0.f;
  }
}

// dropout keep flag (0 or 1) of attention probability (bh, i, j).  The kernels work with the 0/1 flag and apply the
// 1/(1-p) of inverted dropout to their outputs: O = inv_keep * (P.M) V, dV = inv_keep * dO^T (P.M), and
// dS = P (inv_keep M dP - delta) = inv_keep * P (M dP - (1-p) delta), so nothing that enters an MFMA grows beyond its
// undropped bound (the fp16 operand scales of P and dS stay valid for any p).
__device__ __forceinline__ uint32_t drop_row(const AttnArgs& a, int bh, int i) {      // key of row (bh, i): common.h
  return vilco_attn_drop_row(a.drop_seed, (uint64_t)bh * (uint64_t)a.Tq + (uint64_t)i);
}
__device__ __forceinline__ float drop_keep(const AttnArgs& a, int bh, int i, int j) {
  return vilco_attn_drop_keep(drop_row(a, bh, i), (uint32_t)j, a.drop_thresh) ? 1.f : 0.f;
}

// score of (query i, key j) after scale + bias: apply the mask
// mode 4 (NLQ's LocalMaskedMHCA, NLQ/libs/modeling/blocks.py:417-755): keys inside the sliding window |i - j| <= w and
// below kv_len; the reference adds -1e4 to masked keys inside the window instead of removing them, which is exp(-1e4) = 0
// in fp32 next to the always-valid key j = i
__device__ __forceinline__ float mask_score(float s, int i, int j, int len, int Tk, int mode, int w = 0) {
  if (j >= Tk) return -INFINITY;
  if (mode == 0) return j < len ? s : -INFINITY;
  if (mode == 1) return (j >= len && j != i) ? s - 1e30f : s;
  if (mode == 4) return (j < len && j - i <= w && i - j <= w) ? s : -INFINITY;
  return s;
}

// ------------------------------------------------------------------------------------------ forward
template <int HDP, int NP, bool F16, bool DROP>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(AttnArgs a) {
  if (a.drop_thresh) a.drop_seed = vilco_step_seed(a.drop_seed, a.seed_word);
  constexpr int BKV = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sK = reinterpret_cast<__bf16*>(smem_raw);          // [NP][64 keys][HDP]
  __bf16* sVt = sK + NP * BKV * HDP;                          // [NP][HDP d][64 keys]
  __bf16* sP = sVt + NP * HDP * BKV;                          // [4 waves][NP][16 q][64 keys]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int bh = b * a.H + h;
  const long ld = a.C;
  const int q0 = qt * 64 + wave * 16;
  const int qi = q0 + (lane & 15);                 // this lane's query
  const int len = a.mode == 2 ? a.Tk : a.kv_len[b];
  const int bias_ld = a.mode == 3 ? a.Tq + a.Tk : a.Tk;       // mode 3: unshifted XLNet position scores (see AttnArgs)
  const float* bias = a.bias ? a.bias + ((long)bh * a.Tq) * bias_ld : nullptr;
  const int mmode = a.mode == 3 ? 1 : a.mode;
  const __bf16* kbase = a.kn.p + (long)bh * a.kn.batch_stride;
  const __bf16* vbase = a.vt.p + (long)bh * a.vt.batch_stride;
  const int fbH = toff<HDP>(lane & 15, lane >> 4);        // fragment bases (see frag())
  const int fb64 = toff<64>(lane & 15, lane >> 4);
  AttnScales sc = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  if (F16) sc = *a.sc;
  const float qk_scale = F16 ? a.scale * sc.iq * sc.ik : a.scale;     // S = acc / (sq sk)

  QFrag<HDP, NP> qf;
  load_qfrag<HDP, NP>(qf, a.qn, a.qn.p + (long)bh * a.qn.batch_stride, q0, lane);

  f32x4 oacc[HDP / 16];
#pragma unroll
  for (int i = 0; i < HDP / 16; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  int kend = a.Tk;
  if ((mmode == 0 || mmode == 4) && len < kend) kend = len;      // tiles entirely beyond kv_len contribute nothing
  int ntiles = (kend + BKV - 1) / BKV, tlo = 0;
  if (mmode == 4) {                                  // only the key tiles the 64 queries' windows reach
    const int klo = qt * 64 - a.window, khi = qt * 64 + 63 + a.window;
    tlo = klo > 0 ? klo / BKV : 0;
    if (khi / BKV + 1 < ntiles) ntiles = khi / BKV + 1;
  }
  const int kfull = mmode == 4 ? 0 : (len < a.Tk ? len : a.Tk);      // keys below this index need no masking
  __bf16* myP = sP + wave * NP * 16 * BKV;

  TileStage<HDP, BKV, NP> stK;
  TileStage<BKV, HDP, NP> stV;
  if (ntiles > tlo) {
    gload_tile<HDP, BKV, NP>(stK, a.kn, kbase, tlo * BKV, 0, tid);
    gload_tile<BKV, HDP, NP>(stV, a.vt, vbase, 0, tlo * BKV, tid);
  }
  for (int t = tlo; t < ntiles; ++t) {
    const int k0 = t * BKV;
    __syncthreads();                                // previous tile fully consumed
    lstore_tile<HDP, BKV, NP>(stK, sK, tid);
    lstore_tile<BKV, HDP, NP>(stV, sVt, tid);
    __syncthreads();
    if (t + 1 < ntiles) {                           // next tile's loads fly under this tile's MFMAs
      gload_tile<HDP, BKV, NP>(stK, a.kn, kbase, k0 + BKV, 0, tid);
      gload_tile<BKV, HDP, NP>(stV, a.vt, vbase, 0, k0 + BKV, tid);
    }

    // the additive bias of this tile is requested before the S MFMAs so its latency hides under them
    const bool plain = (bias == nullptr) && (k0 + BKV <= kfull);      // wave-uniform: no bias, nothing masked
    float bvt[4][4];
    if (bias) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bvt[mi][r] = 0.f;
        if (qi < a.Tq) bias4(bvt[mi], bias, qi, k0 + mi * 16 + (lane >> 4) * 4, bias_ld, a);
      }
    }
    unsigned keep_bits = 0xffffu;          // dropout keep flags of this lane's 16 (mi, r) probabilities
    if (DROP && a.drop_thresh) {
      keep_bits = 0u;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          keep_bits |= (drop_keep(a, bh, qi, k0 + mi * 16 + (lane >> 4) * 4 + r) != 0.f ? 1u : 0u) << (mi * 4 + r);
    }
    // S^T[key][q] = K Q^T : 4 m-tiles of 16 keys, this wave's 16 queries
    f32x4 s[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HDP / 32; ++ks) {
        bf16x8 ka[3];
#pragma unroll
        for (int q = 0; q < NP; ++q) ka[q] = frag<HDP>(sK + q * BKV * HDP, fbH, mi * 16, ks);
        c = mfma_parts<NP, F16>(ka, qf.f[ks], c);
      }
      s[mi] = c;
    }
    // lane holds query qi, keys k0 + mi*16 + 4*(lane>>4) + r
    float tmax = -INFINITY;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int jb = k0 + mi * 16 + (lane >> 4) * 4;
      if (plain) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[mi][r] *= qk_scale; tmax = fmaxf(tmax, s[mi][r]); }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = mask_score(s[mi][r] * qk_scale + (bias ? bvt[mi][r] : 0.f), qi, jb + r, len, a.Tk, mmode, a.window);
          s[mi][r] = x;
          tmax = fmaxf(tmax, x);
        }
      }
    }
    tmax = quad16_max(tmax);
    const float m_new = fmaxf(m_run, tmax);
    const bool dead = m_new == -INFINITY;
    const float alpha = dead ? 1.f : fast_exp(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = dead ? 0.f : fast_exp(s[mi][r] - m_new);
        psum += p[r];                                   // the softmax denominator is over the undropped probabilities
        if (DROP && !((keep_bits >> (mi * 4 + r)) & 1u)) p[r] = 0.f;
        if (F16) p[r] *= P_SCALE;
      }
      bf16x4 pp[3];
      split4s<NP, F16>(p, pp);
      // P[q][keys mi*16 + 4*(lane>>4) .. +3]: chunk = 2*mi + (lane>>5), half = (lane>>4)&1
      const int off = toff<BKV>(lane & 15, 2 * mi + (lane >> 5)) + ((lane >> 4) & 1) * 4;
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x4*>(myP + q * 16 * BKV + off) = pp[q];
    }
    psum = quad16_sum(psum);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < HDP / 16; ++i) {
      oacc[i][0] *= alpha; oacc[i][1] *= alpha; oacc[i][2] *= alpha; oacc[i][3] *= alpha;
    }
    // P is written and read by this wave only: the LDS queue of a wave is in order, no workgroup barrier needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // O^T[d][q] += V^T P^T : m-tiles over d, k-steps over the 64 keys
#pragma unroll
    for (int ks = 0; ks < BKV / 32; ++ks) {
      bf16x8 pb[3];
#pragma unroll
      for (int q = 0; q < NP; ++q) pb[q] = frag<BKV>(myP + q * 16 * BKV, fb64, 0, ks);
#pragma unroll
      for (int di = 0; di < HDP / 16; ++di) {
        bf16x8 va[3];
#pragma unroll
        for (int q = 0; q < NP; ++q) va[q] = frag<BKV>(sVt + q * HDP * BKV, fb64, di * 16, ks);
        oacc[di] = mfma_parts<NP, F16>(va, pb, oacc[di]);
      }
    }
  }

  // finish: lane holds query qi, channels di*16 + 4*(lane>>4) + r
  if (qi < a.Tq) {
    const float inv = (l_run > 0.f ? 1.f / l_run : 0.f) * (F16 ? P_INV * sc.iv : 1.f) * a.drop_inv_keep;
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
    if (a.lse && (lane >> 4) == 0) a.lse[(long)bh * a.Tq + qi] = m_run + logf(l_run);
  }
}

// ------------------------------------------------------------------------------------------ forward, hd = 64 fast path
// The MQ blocks' attention (blocks.py:191-247: hd = 64, prefix key mask, no bias, attn_pdrop = 0) in fp16 x2 planes.
// Differences from the general kernel above, all of them about instruction count -- the general kernel spends ~360
// VALU + ~50 SALU instructions per 16-query x 64-key wave tile next to 48 MFMAs (rocprofv3 SQ_INSTS_*), i.e. the
// vector ALU, not the matrix pipe or the LDS, sets its speed:
//  * 128 queries per workgroup, 32 per wave: two MFMA column groups share every K / V fragment read and every loop
//    overhead instruction;
//  * P never goes through LDS.  S^T = K Q^T leaves lane (q, g4) with keys 16 mi + 4 g4 + r; the contraction index of
//    O^T += V^T P^T may be enumerated in any order as long as both operands agree, so k-step ks, element e of a lane is
//    key 32 ks + 16 (e >> 2) + 4 g4 + (e & 3): P's B fragments are then the lane's own registers, and the V^T
//    fragments come out of the NATURAL V tile by gfx950's transposing LDS read with its two 4-key reads placed on
//    exactly those keys (tr_frag64) -- no transposed V planes are packed for this path;
//  * softmax in the exp2 domain with the score scale, log2(e) and the 2^15 fp16 range shift of P folded into one FMA
//    per score; the row sums stay per lane until the epilogue; O is rescaled only when some row's maximum moved;
//  * interior tiles carry no masking or bounds code at all (out-of-range rows / columns of the last tile are clamped
//    to valid memory and masked by the key-length test).
__device__ __forceinline__ f32x4 mfma3h(const bf16x8 (&a)[2], const bf16x8 (&b)[2], f32x4 c) {
  const f16x8 a0 = __builtin_bit_cast(f16x8, a[0]), a1 = __builtin_bit_cast(f16x8, a[1]);
  const f16x8 b0 = __builtin_bit_cast(f16x8, b[0]), b1 = __builtin_bit_cast(f16x8, b[1]);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c, 0, 0, 0);
}

constexpr int F64_Q = 128;       // queries per workgroup of the fast path

// max over the workgroup of a per-thread value -> out[linear block id] (smem: the kernel's LDS, free again after a barrier)
__device__ __forceinline__ void block_amax_out(float m, float* out, float* smem_f) {
  m = wave_max(m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem_f[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    out[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = fmaxf(fmaxf(smem_f[0], smem_f[1]), fmaxf(smem_f[2], smem_f[3]));
}
constexpr int RS64 = 80;                 // LDS row stride (elements) of a natural [64 rows][64 d] tile
constexpr int PL64 = 64 * RS64;          // elements per part

__device__ __forceinline__ bf16x8 tr_frag64(const __bf16* p) {     // k rows at p, k rows + 16 at p + 16 rows
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 16 * RS64));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}


#ifdef VILCO_LAB_ATTN   // tools/lab only: in-kernel cycle stamps of block (0,0,0) wave 0 (never compiled into the product)
__device__ unsigned long long vilco_lab_attn_stamps[64 * 8];
#define ASTAMP(i)                                                                                             \
  do {                                                                                                         \
    if (stamp_on && stamp_t < 64) {                                                                            \
      const unsigned long long c_ = __builtin_amdgcn_s_memtime();                                              \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
      if (lane == 0) vilco_lab_attn_stamps[stamp_t * 8 + (i)] = c_;                                            \
    }                                                                                                          \
  } while (0)
#else
#define ASTAMP(i) do {} while (0)
#endif

// XL kernels: NR rows of position scores (256 B each: 64 keys of one query row of the unshifted matrix, element
// bd[i][Tq - i + j] = bias + i (ld - 1) + Tq + j) straight into LDS by LDS-DMA, one row per instruction, row r to lds0 + r * 272.
// Inline asm with scalar addressing (row address and LDS address in SGPRs, the lane's key offset `voff` = 4 min(j, Tk - 1) in
// one VGPR), for two measured reasons (tools/lab/attn_stamps_xl.py, round 5: the phase that issued the 32 rows took 4.4 k cycles
// against 1.4 k without them): (1) through __builtin_amdgcn_global_load_lds the compiler's wait-count pass treats every later
// LDS read as a possible alias of the DMA's LDS write and puts s_waitcnt vmcnt(0) right behind the issue loop -- the prefetch
// was synchronous; the asm is opaque to it and the kernels wait explicitly where they consume; (2) the per-lane 64-bit
// address arithmetic was ~10 VALU instructions per row.  M0 carries the LDS address (no other user of M0 in these kernels).
// `q_first` and `bias` must be wave-uniform.
template <int NR>
__device__ __forceinline__ void xl_dma_rows(const float* bias, int ld, int Tq, int q_first, unsigned voff, unsigned lds0) {
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    int qi = q_first + r;
    qi = qi < Tq ? qi : Tq - 1;
    const float* row = bias + ((long)qi * (ld - 1) + Tq);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds0 + (unsigned)r * 272u), "v"(voff), "s"(row) : "memory");
  }
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// XL: XLNet's relative attention (mask mode 3: additive position scores read unshifted from bd[b,h,i,Tq-i+j], the XLNet
// mask -- keys >= kv_len masked except the diagonal --, dropout on the probabilities): same kernel, three more steps
template <bool XL>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_fwd64_kernel(AttnArgs a) {
  if (a.drop_thresh) a.drop_seed = vilco_step_seed(a.drop_seed, a.seed_word);
  constexpr int HDP = 64, BKV = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sK = reinterpret_cast<__bf16*>(smem_raw);          // [2 parts][64 keys][RS64]
  __bf16* sV = sK + 2 * PL64;                                 // [2 parts][64 keys][RS64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int bh = b * a.H + h;
  const int q0 = blockIdx.x * F64_Q + wave * 32;
  const int len = a.mode == 2 ? a.Tk : a.kv_len[b];
  const int kend = XL ? a.Tk : (len < a.Tk ? len : a.Tk);       // XL: every key tile is visited (the diagonal is always visible)
  const int ntiles = (kend + BKV - 1) / BKV;
  const AttnScales sc = *a.sc;
  const int bias_ld = a.Tq + a.Tk;
  const float* bias = XL ? a.bias + ((long)bh * a.Tq) * bias_ld : nullptr;
  // XL: the position scores of a key tile (this wave's 32 query rows x 64 keys, 256 B per row at a 4-byte aligned
  // offset of the unshifted matrix) are fetched straight into LDS one tile ahead (global_load_lds_dword: one row per
  // instruction, no registers), so the fetch -- bandwidth-bound, ~11 k cycles per tile when waited for in place -- runs
  // under the previous tile's MFMAs.  Rows are wave-private: no workgroup barrier is involved.
  constexpr int RSBF = 68;                                   // floats per staged row (272 B: 16-byte aligned, 4 banks per row)
  float* sBias = reinterpret_cast<float*>(sV + 2 * PL64) + wave * 32 * RSBF;
  const int q0s = __builtin_amdgcn_readfirstlane(q0);
  const unsigned sbias_lds = XL ? (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr_of(sBias)) : 0u;
  [[maybe_unused]] auto bias_dma = [&](int k0, int g) {      // the 16 rows of query group g
    const int j = k0 + lane;
    xl_dma_rows<16>(bias, bias_ld, a.Tq, q0s + 16 * g, 4u * (unsigned)(j < a.Tk ? j : a.Tk - 1), sbias_lds + (unsigned)g * 16u * 272u);
  };
  const float c2 = a.scale * sc.iq * sc.ik * 1.44269504088896340736f;      // log2-domain score = acc * c2
  const __bf16* kbase = a.kn.p + (long)bh * a.kn.batch_stride;
  const __bf16* vbase = a.vn.p + (long)bh * a.vn.batch_stride;
  const int g4 = lane >> 4;
  const int fbn = (lane & 15) * RS64 + g4 * 8;                          // natural fragment: row lane & 15, chunk g4 (+ 4 ks)
  const int fbt = (4 * g4 + ((lane & 15) >> 2)) * RS64 + 4 * (lane & 3);   // transposing fragment (tr_frag64)

  QFrag<HDP, 2> qf[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) load_qfrag<HDP, 2>(qf[g], a.qn, a.qn.p + (long)bh * a.qn.batch_stride, q0 + 16 * g, lane);

  f32x4 oacc[2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) oacc[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  [[maybe_unused]] uint32_t rkey[2] = {0u, 0u};               // XL + dropout: this lane's two rows of the mask (common.h)
  if constexpr (XL) {
    if (a.drop_thresh) { rkey[0] = drop_row(a, bh, q0 + (lane & 15)); rkey[1] = drop_row(a, bh, q0 + 16 + (lane & 15)); }
  }

  // staging: thread owns chunks (row = tid >> 3 (+32), 16-byte chunk c = tid & 7) of the K and the V tile (both natural)
  bf16x8 stK[2][2], stV[2][2];
  const int srow = tid >> 3, sc8 = tid & 7;
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int kr = k0 + srow + 32 * i;
      kr = kr < a.kn.rows ? kr : a.kn.rows - 1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        stK[q][i] = *reinterpret_cast<const bf16x8*>(kbase + q * a.kn.part_stride + (long)kr * HDP + sc8 * 8);
        stV[q][i] = *reinterpret_cast<const bf16x8*>(vbase + q * a.vn.part_stride + (long)kr * HDP + sc8 * 8);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = (srow + 32 * i) * RS64 + sc8 * 8;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<bf16x8*>(sK + q * PL64 + o) = stK[q][i];
        *reinterpret_cast<bf16x8*>(sV + q * PL64 + o) = stV[q][i];
      }
    }
  };

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const f32x2 c2v = {c2, c2};
#ifdef VILCO_LAB_ATTN
  const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && wave == 0;
#endif
  // one 64-key tile (MASKED only for the tile that holds the end of the keys)
  auto tile = [&](int t, bool more, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int k0 = t * BKV;
#ifdef VILCO_LAB_ATTN
    const int stamp_t = t;
#endif
    ASTAMP(0);
    // XL: this tile's position scores (issued during the previous tile) have landed -- the K / V registers below need every
    // outstanding load anyway; the next tile's prefetches are issued after this point
    if constexpr (XL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                // previous tile fully consumed
    lstore();
    __syncthreads();
    ASTAMP(1);
    if (more) gload(k0 + BKV);                      // next tile's loads fly under this tile's MFMAs

    // S^T[key][q] = K Q^T for both query groups.  Fragment reads run one step ahead of the MFMAs that use them, and the
    // two accumulators of a step alternate so that consecutive MFMAs never depend on each other.
    f32x4 s[2][4];
    bf16x8 kf[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) kf[0][q] = *reinterpret_cast<const bf16x8*>(sK + q * PL64 + fbn);
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int mi = st >> 1, ks = st & 1;
      if (st + 1 < 8) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
          kf[(st + 1) & 1][q] = *reinterpret_cast<const bf16x8*>(sK + q * PL64 + fbn + ((st + 1) >> 1) * 16 * RS64 + ((st + 1) & 1) * 32);
      }
      const f16x8 k0h = __builtin_bit_cast(f16x8, kf[st & 1][0]), k1h = __builtin_bit_cast(f16x8, kf[st & 1][1]);
      const f16x8 a0 = __builtin_bit_cast(f16x8, qf[0].f[ks][0]), a1 = __builtin_bit_cast(f16x8, qf[0].f[ks][1]);
      const f16x8 b0 = __builtin_bit_cast(f16x8, qf[1].f[ks][0]), b1 = __builtin_bit_cast(f16x8, qf[1].f[ks][1]);
      f32x4 c0 = ks ? s[0][mi] : f32x4{0.f, 0.f, 0.f, 0.f}, c1 = ks ? s[1][mi] : f32x4{0.f, 0.f, 0.f, 0.f};
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, a0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, b0, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, a1, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, b1, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, a0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, b0, c1, 0, 0, 0);
      s[0][mi] = c0; s[1][mi] = c1;
    }
    ASTAMP(2);
    if constexpr (MASKED && !XL) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (k0 + mi * 16 + g4 * 4 + r >= kend) { s[0][mi][r] = -INFINITY; s[1][mi][r] = -INFINITY; }
    }
    bf16x8 pb[2][2][2];                             // [group][k-step][part]: P^T B fragments, built in registers
    [[maybe_unused]] const uint32_t jw0 = (uint32_t)(k0 + g4 * 4) * VILCO_ATTN_DROP_W;      // dropout: key part of the element hash
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      // scores to the log2 domain, then the row maximum over this lane's 16 keys and the 4 lanes of the query
      [[maybe_unused]] const int qi_g = q0 + 16 * g + (lane & 15);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        f32x2 a01 = f32x2{s[g][mi][0], s[g][mi][1]} * c2v, a23 = f32x2{s[g][mi][2], s[g][mi][3]} * c2v;
        if constexpr (XL) {
          const int jb = k0 + mi * 16 + g4 * 4;
          const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + (16 * g + (lane & 15)) * RSBF + mi * 16 + g4 * 4);   // bd[i][Tq - i + j]
          const float bsc = a.scale * 1.44269504088896340736f;
          a01 += f32x2{bv[0], bv[1]} * bsc;
          a23 += f32x2{bv[2], bv[3]} * bsc;
          if constexpr (MASKED) {                                                     // keys >= kv_len except the diagonal, keys >= Tk
            if ((jb + 0 >= len && jb + 0 != qi_g) || jb + 0 >= a.Tk) a01[0] = -INFINITY;
            if ((jb + 1 >= len && jb + 1 != qi_g) || jb + 1 >= a.Tk) a01[1] = -INFINITY;
            if ((jb + 2 >= len && jb + 2 != qi_g) || jb + 2 >= a.Tk) a23[0] = -INFINITY;
            if ((jb + 3 >= len && jb + 3 != qi_g) || jb + 3 >= a.Tk) a23[1] = -INFINITY;
          }
        }
        s[g][mi] = f32x4{a01[0], a01[1], a23[0], a23[1]};
      }
      float mx = fmaxf(fmaxf(s[g][0][0], s[g][0][1]), fmaxf(s[g][0][2], s[g][0][3]));
#pragma unroll
      for (int mi = 1; mi < 4; ++mi) mx = fmaxf(mx, fmaxf(fmaxf(s[g][mi][0], s[g][mi][1]), fmaxf(s[g][mi][2], s[g][mi][3])));
      mx = quad16_max(mx);
      const float m_new = fmaxf(m_run[g], mx);             // finite (prefix masks: every tile holds at least one valid key)
      const bool dead = XL && m_new == -INFINITY;          // XL: a row can have seen only masked keys so far
      const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m_run[g] - m_new);
      const float off = dead ? 0.f : 15.f - m_new;         // P is formed as P * 2^15 (fp16 range)
      const f32x2 offv = {off, off};
      f32x2 psum = {0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 h0, h1;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f32x2 sv = {s[g][2 * ks + (e >> 2)][e & 3], s[g][2 * ks + (e >> 2)][(e & 3) + 1]};
          const f32x2 arg = sv + offv;                     // v_pk_add_f32
          f32x2 p = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
          psum += p;                                       // v_pk_add_f32 (the denominator is over the undropped probabilities)
          if constexpr (XL) {
            if (a.drop_thresh) {
              const uint32_t jw = jw0 + (uint32_t)((2 * ks + (e >> 2)) * 16 + (e & 3)) * VILCO_ATTN_DROP_W;
              p[0] = vilco_attn_drop_keep_w(rkey[g], jw, a.drop_thresh) ? p[0] : 0.f;
              p[1] = vilco_attn_drop_keep_w(rkey[g], jw + VILCO_ATTN_DROP_W, a.drop_thresh) ? p[1] : 0.f;
            }
          }
          const f16x2 hp = {(_Float16)p[0], (_Float16)p[1]};                 // v_cvt_pk_f16_f32 (round to nearest even)
          const uint32_t hpu = __builtin_bit_cast(uint32_t, hp);
          uint32_t lo;                                     // second part = fp16(p - h0): the mixed-precision FMA reads the
          asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hpu), "v"(p[0]));      // fp16 half in place
          asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hpu), "v"(p[1]));
          h0[e >> 1] = hpu;
          h1[e >> 1] = lo;
        }
        pb[g][ks][0] = __builtin_bit_cast(bf16x8, h0);
        pb[g][ks][1] = __builtin_bit_cast(bf16x8, h1);
      }
      l_run[g] = l_run[g] * alpha + (psum[0] + psum[1]);   // per-lane partial row sum (this lane's keys only)
      m_run[g] = m_new;
      if (__any(alpha != 1.f)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { oacc[g][i][0] *= alpha; oacc[g][i][1] *= alpha; oacc[g][i][2] *= alpha; oacc[g][i][3] *= alpha; }
      }
      if constexpr (XL) {
        if (more) {                                 // next tile's position scores of this group's 16 rows: the wave has read them
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (LDS queue in order); group 0's fly under group 1's softmax
          bias_dma(k0 + BKV, g);
        }
      }
    }
    ASTAMP(3);
    // O^T[d][q] += V^T P^T, V^T fragments by transposing reads of the natural tile (keys in the order P sits in the registers)
    bf16x8 vf[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) vf[0][q] = tr_frag64(sV + q * PL64 + fbt);
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int ks = st >> 2, di = st & 3;
      if (st + 1 < 8) {
#pragma unroll
        for (int q = 0; q < 2; ++q) vf[(st + 1) & 1][q] = tr_frag64(sV + q * PL64 + fbt + ((st + 1) >> 2) * 32 * RS64 + ((st + 1) & 3) * 16);
      }
      const f16x8 v0h = __builtin_bit_cast(f16x8, vf[st & 1][0]), v1h = __builtin_bit_cast(f16x8, vf[st & 1][1]);
      const f16x8 a0 = __builtin_bit_cast(f16x8, pb[0][ks][0]), a1 = __builtin_bit_cast(f16x8, pb[0][ks][1]);
      const f16x8 b0 = __builtin_bit_cast(f16x8, pb[1][ks][0]), b1 = __builtin_bit_cast(f16x8, pb[1][ks][1]);
      f32x4 c0 = oacc[0][di], c1 = oacc[1][di];
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1h, a0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1h, b0, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0h, a1, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0h, b1, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0h, a0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0h, b0, c1, 0, 0, 0);
      oacc[0][di] = c0; oacc[1][di] = c1;
    }
    ASTAMP(5);
  };

  if (ntiles > 0) gload(0);
  if constexpr (XL) { if (ntiles > 0) { bias_dma(0, 0); bias_dma(0, 1); } }
  const int nfull = (XL ? (len < a.Tk ? len : a.Tk) : kend) / BKV;      // tiles with all 64 keys valid
  for (int t = 0; t < nfull; ++t) tile(t, t + 1 < ntiles, std::false_type{});
  for (int t = nfull; t < ntiles; ++t) tile(t, t + 1 < ntiles, std::true_type{});

  // finish: lane holds query q0 + 16 g + (lane & 15), channels di*16 + 4*g4 + r
  typedef _Float16 h4o __attribute__((ext_vector_type(4)));
  const float so = a.op0 ? sc.sv * a.o_scale_mul : 0.f;      // scale of the output planes (see AttnArgs.op0)
  float am = 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    float l = l_run[g];
    l = quad16_sum(l);
    const int qi = q0 + 16 * g + (lane & 15);
    if (qi < a.Tq) {
      const float inv = (l > 0.f ? 1.f / l : 0.f) * sc.iv * (XL ? a.drop_inv_keep : 1.f);
      float* og = a.o + ((long)b * a.Tq + qi) * a.C + h * HDP;
#pragma unroll
      for (int di = 0; di < 4; ++di) {
        const float4 v = make_float4(oacc[g][di][0] * inv, oacc[g][di][1] * inv, oacc[g][di][2] * inv, oacc[g][di][3] * inv);
        *reinterpret_cast<float4*>(og + di * 16 + g4 * 4) = v;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        if (a.op0) {
          const float xs[4] = {v.x * so, v.y * so, v.z * so, v.w * so};      // exact (power of two)
          h4o h0, h1;
#pragma unroll
          for (int e = 0; e < 4; ++e) { h0[e] = (_Float16)xs[e]; h1[e] = (_Float16)(xs[e] - (float)h0[e]); }
          const long po = ((long)b * a.Tq + qi) * a.C + h * HDP + di * 16 + g4 * 4;
          *reinterpret_cast<h4o*>(a.op0 + po) = h0;
          *reinterpret_cast<h4o*>(a.op0 + a.o_plane_stride + po) = h1;
        }
      }
      if (a.lse && g4 == 0) a.lse[(long)bh * a.Tq + qi] = 0.69314718055994530942f * (m_run[g] + __builtin_amdgcn_logf(l) - 15.f);
    }
  }
  if (a.op0) {
    if ((blockIdx.x | blockIdx.y | blockIdx.z) == 0 && tid == 0) { a.o_inv_scale[0] = sc.iv / a.o_scale_mul; a.o_inv_scale[1] = so; }
    if (blockIdx.x == gridDim.x - 1 && b == a.B - 1) {          // the planes' zero rows, this head's 64 columns
      const long rows = (long)a.B * a.Tq;
      const h4o hz = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      for (long i = tid; i < (a.o_rows32 - rows) * 16; i += ATT_THREADS) {
        const long po = (rows + i / 16) * a.C + h * HDP + (i % 16) * 4;
        *reinterpret_cast<h4o*>(a.op0 + po) = hz;
        *reinterpret_cast<h4o*>(a.op0 + a.o_plane_stride + po) = hz;
      }
    }
  }
  if (a.am_o) block_amax_out(am, a.am_o, reinterpret_cast<float*>(smem_raw));
}

// ------------------------------------------------------------------------------------------ XLNet position scores
// bd[b,h,i,p] = qr_i . kr_p for the band p in [T - i, 2T - i) of the UNSHIFTED [T][2T] matrix (modeling_xlnet_x.py:256-288: the
// only part rel_shift_bnij keeps) -- round 5, replacing the band-limited batched GEMM with K = 64: one K interval per 192 x 128
// tile means a tile is all prologue and epilogue there (one workgroup per CU, DMA latency and the 98 KB write-back strictly one
// after the other: 395 us = 1.9 TB/s of writes at config P).  This is the attention kernels' own first product instead: 128
// query rows per workgroup, 32 per wave with their qr fragments in registers for the whole kernel, the kr tiles of the
// workgroup's band streamed through LDS (next tile's loads under the current tile's MFMAs), S^T = KR QR^T leaves a lane with 4
// consecutive p of one query: one float4 store per 16-row block.  a.qn = qr planes, a.kn = kr planes ([H] or, a.mode = 1, [B*H]
// batches of 2T rows), a.o = bd, a.sc = {1/s, s} of qr and kr.
__global__ __launch_bounds__(ATT_THREADS, 2) void xl_scores64_kernel(AttnArgs a) {
  constexpr int HDP = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sK = reinterpret_cast<__bf16*>(smem_raw);          // [2 parts][64 rows of kr][RS64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int bh = b * a.H + h;
  const int T = a.Tq, P = 2 * a.Tq;
  const int Q0 = blockIdx.x * F64_Q, q0 = Q0 + wave * 32;
  const AttnScales sc = *a.sc;
  const float unscale = sc.iq * sc.ik;
  const __bf16* kbase = a.kn.p + (long)(a.mode ? bh : h) * a.kn.batch_stride;
  const int g4 = lane >> 4;
  const int fbn = (lane & 15) * RS64 + g4 * 8;

  QFrag<HDP, 2> qf[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) load_qfrag<HDP, 2>(qf[g], a.qn, a.qn.p + (long)bh * a.qn.batch_stride, q0 + 16 * g, lane);

  bf16x8 stK[2][2];
  const int srow = tid >> 3, sc8 = tid & 7;
  auto gload = [&](int p0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int kr = p0 + srow + 32 * i;
      kr = kr < a.kn.rows ? kr : a.kn.rows - 1;
#pragma unroll
      for (int q = 0; q < 2; ++q) stK[q][i] = *reinterpret_cast<const bf16x8*>(kbase + q * a.kn.part_stride + (long)kr * HDP + sc8 * 8);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = (srow + 32 * i) * RS64 + sc8 * 8;
#pragma unroll
      for (int q = 0; q < 2; ++q) *reinterpret_cast<bf16x8*>(sK + q * PL64 + o) = stK[q][i];
    }
  };
  // the workgroup's band: rows Q0 .. min(Q0 + 127, T - 1)
  const int qlast = (Q0 + F64_Q - 1 < T ? Q0 + F64_Q - 1 : T - 1);
  const int t_lo = (T - qlast) / 64, t_hi = (P - Q0 + 63) / 64;            // p tiles [t_lo, t_hi)
  float* orow[2];
  int irow[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    irow[g] = q0 + 16 * g + (lane & 15);
    orow[g] = a.o + ((long)bh * T + (irow[g] < T ? irow[g] : T - 1)) * P;
  }
  if (t_lo < t_hi) gload(t_lo * 64);
  for (int t = t_lo; t < t_hi; ++t) {
    const int p0 = t * 64;
    __syncthreads();
    lstore();
    __syncthreads();
    if (t + 1 < t_hi) gload(p0 + 64);
    if (p0 + 63 < T - (q0 + 31) || p0 >= P - q0) continue;      // (wave-uniform) nothing of this tile lies in this wave's band
    f32x4 s[2][4];
    bf16x8 kf[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) kf[0][q] = *reinterpret_cast<const bf16x8*>(sK + q * PL64 + fbn);
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int mi = st >> 1, ks = st & 1;
      if (st + 1 < 8) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
          kf[(st + 1) & 1][q] = *reinterpret_cast<const bf16x8*>(sK + q * PL64 + fbn + ((st + 1) >> 1) * 16 * RS64 + ((st + 1) & 1) * 32);
      }
      const f16x8 k0h = __builtin_bit_cast(f16x8, kf[st & 1][0]), k1h = __builtin_bit_cast(f16x8, kf[st & 1][1]);
      const f16x8 a0 = __builtin_bit_cast(f16x8, qf[0].f[ks][0]), a1 = __builtin_bit_cast(f16x8, qf[0].f[ks][1]);
      const f16x8 b0 = __builtin_bit_cast(f16x8, qf[1].f[ks][0]), b1 = __builtin_bit_cast(f16x8, qf[1].f[ks][1]);
      f32x4 c0 = ks ? s[0][mi] : f32x4{0.f, 0.f, 0.f, 0.f}, c1 = ks ? s[1][mi] : f32x4{0.f, 0.f, 0.f, 0.f};
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, a0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, b0, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, a1, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, b1, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, a0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, b0, c1, 0, 0, 0);
      s[0][mi] = c0; s[1][mi] = c1;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int i = irow[g];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int p = p0 + mi * 16 + g4 * 4;
        if (i < T && p + 3 >= T - i && p < P - i && p + 3 < P)      // some of the 4 columns lie in row i's band
          *reinterpret_cast<float4*>(orow[g] + p) = make_float4(s[g][mi][0] * unscale, s[g][mi][1] * unscale, s[g][mi][2] * unscale, s[g][mi][3] * unscale);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ backward: dQ (+ dBias)
template <int HDP, int NP, bool F16, bool DROP>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dq_kernel(AttnArgs a) {
  if (a.drop_thresh) a.drop_seed = vilco_step_seed(a.drop_seed, a.seed_word);
  constexpr int BKV = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sK = reinterpret_cast<__bf16*>(smem_raw);          // [NP][64 keys][HDP]
  __bf16* sV = sK + NP * BKV * HDP;                           // [NP][64 keys][HDP]
  __bf16* sKt = sV + NP * BKV * HDP;                          // [NP][HDP d][64 keys]
  __bf16* sS = sKt + NP * HDP * BKV;                          // [4 waves][NP][16 q][64 keys]  (dS)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int bh = b * a.H + h;
  const long ld = a.C;
  const int q0 = qt * 64 + wave * 16;
  const int qi = q0 + (lane & 15);
  const int len = a.mode == 2 ? a.Tk : a.kv_len[b];
  const long row_bh = (long)bh * a.Tq;
  const int bias_ld = a.mode == 3 ? a.Tq + a.Tk : a.Tk;
  const float* bias = a.bias ? a.bias + row_bh * bias_ld : nullptr;
  float* dbias = a.dbias ? a.dbias + row_bh * a.Tk : nullptr;
  const int mmode = a.mode == 3 ? 1 : a.mode;
  const __bf16* knb = a.kn.p + (long)bh * a.kn.batch_stride;
  const __bf16* vnb = a.vn.p + (long)bh * a.vn.batch_stride;
  const __bf16* ktb = a.kt.p + (long)bh * a.kt.batch_stride;
  const int fbH = toff<HDP>(lane & 15, lane >> 4);
  const int fb64 = toff<64>(lane & 15, lane >> 4);
  AttnScales sc = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  if (F16) sc = *a.sc;
  const float qk_scale = F16 ? a.scale * sc.iq * sc.ik : a.scale;
  // fp16 x2: dp stays in plane units (dP * sdO * sV), delta is brought there, and dS' = P (dp - delta') * 2^-22
  const float ds_unscale = (F16 ? ds_inv<HDP>() * sc.ido * sc.iv : 1.f) * a.drop_inv_keep;       // dS = dS' * this

  QFrag<HDP, NP> qf, dof;
  load_qfrag<HDP, NP>(qf, a.qn, a.qn.p + (long)bh * a.qn.batch_stride, q0, lane);
  load_qfrag<HDP, NP>(dof, a.don, a.don.p + (long)bh * a.don.batch_stride, q0, lane);
  const float lse = qi < a.Tq ? a.lse[row_bh + qi] : 0.f;
  // (1-p) delta in plane units.  Multiplied by the two scales one after the other: their product alone overflows fp32
  // when both tensors are tiny or zero (a clip dropped by stochastic depth has dO == 0 -> s = 2^126), delta * sdO does not
  // delta_i = dO_i . O_i in exact fp32: lane group g covers channels ks*32 + 8g .. +7 of this lane's query, two
  // shuffles sum the four groups; the result is also left in a.delta for attn_bwd_dkdv (which runs after this kernel)
  float delta_i = 0.f;
  if (qi < a.Tq) {
    const float* po = a.o_in + ((long)b * a.Tq + qi) * ld + h * a.hd;
    const float* pg = a.dout + ((long)b * a.Tq + qi) * ld + h * a.hd;
#pragma unroll
    for (int ks = 0; ks < HDP / 32; ++ks)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int d = ks * 32 + (lane >> 4) * 8 + hh * 4;
        if (d + 4 <= a.hd) {
          const float4 x = *reinterpret_cast<const float4*>(po + d), y = *reinterpret_cast<const float4*>(pg + d);
          delta_i += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
      }
  }
  delta_i = quad16_sum(delta_i);
  if ((lane >> 4) == 0 && qi < a.Tq) a.delta[row_bh + qi] = delta_i;
  const float dlt = F16 ? ((delta_i * sc.sdo) * sc.sv) / a.drop_inv_keep : delta_i / a.drop_inv_keep;

  f32x4 dqacc[HDP / 16];
#pragma unroll
  for (int i = 0; i < HDP / 16; ++i) dqacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kend = a.Tk;
  if ((mmode == 0 || mmode == 4) && len < kend) kend = len;
  int ntiles = (kend + BKV - 1) / BKV, tlo = 0;
  if (mmode == 4) {
    const int klo = qt * 64 - a.window, khi = qt * 64 + 63 + a.window;
    tlo = klo > 0 ? klo / BKV : 0;
    if (khi / BKV + 1 < ntiles) ntiles = khi / BKV + 1;
  }
  const int kfull = mmode == 4 ? 0 : (len < a.Tk ? len : a.Tk);
  __bf16* myS = sS + wave * NP * 16 * BKV;

  TileStage<HDP, BKV, NP> stK, stV;
  TileStage<BKV, HDP, NP> stKt;
  if (ntiles > tlo) {
    gload_tile<HDP, BKV, NP>(stK, a.kn, knb, tlo * BKV, 0, tid);
    gload_tile<HDP, BKV, NP>(stV, a.vn, vnb, tlo * BKV, 0, tid);
    gload_tile<BKV, HDP, NP>(stKt, a.kt, ktb, 0, tlo * BKV, tid);
  }
  for (int t = tlo; t < ntiles; ++t) {
    const int k0 = t * BKV;
    __syncthreads();
    lstore_tile<HDP, BKV, NP>(stK, sK, tid);
    lstore_tile<HDP, BKV, NP>(stV, sV, tid);
    lstore_tile<BKV, HDP, NP>(stKt, sKt, tid);
    __syncthreads();
    if (t + 1 < ntiles) {
      gload_tile<HDP, BKV, NP>(stK, a.kn, knb, k0 + BKV, 0, tid);
      gload_tile<HDP, BKV, NP>(stV, a.vn, vnb, k0 + BKV, 0, tid);
      gload_tile<BKV, HDP, NP>(stKt, a.kt, ktb, 0, k0 + BKV, tid);
    }
    const bool plain = (bias == nullptr) && (k0 + BKV <= kfull);
    float bvt[4][4];
    if (bias) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bvt[mi][r] = 0.f;
        if (qi < a.Tq) bias4(bvt[mi], bias, qi, k0 + mi * 16 + (lane >> 4) * 4, bias_ld, a);
      }
    }
    unsigned keep_bits = 0xffffu;          // dropout keep flags of this lane's 16 (mi, r) probabilities
    if (DROP && a.drop_thresh) {
      keep_bits = 0u;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          keep_bits |= (drop_keep(a, bh, qi, k0 + mi * 16 + (lane >> 4) * 4 + r) != 0.f ? 1u : 0u) << (mi * 4 + r);
    }

#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HDP / 32; ++ks) {
        bf16x8 ka[3], va[3];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          ka[q] = frag<HDP>(sK + q * BKV * HDP, fbH, mi * 16, ks);
          va[q] = frag<HDP>(sV + q * BKV * HDP, fbH, mi * 16, ks);
        }
        s = mfma_parts<NP, F16>(ka, qf.f[ks], s);       // S^T  = K Q^T
        dp = mfma_parts<NP, F16>(va, dof.f[ks], dp);    // dP^T = V dO^T
      }
      const int jb = k0 + mi * 16 + (lane >> 4) * 4;
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = s[r] * qk_scale;
        if (!plain) x = mask_score(x + (bias ? bvt[mi][r] : 0.f), qi, jb + r, len, a.Tk, mmode, a.window);
        const float p = (x == -INFINITY) ? 0.f : fast_exp(x - lse);
        const float dpr = (!DROP || ((keep_bits >> (mi * 4 + r)) & 1u)) ? dp[r] : 0.f;
        ds[r] = p * (dpr - dlt);
        if (F16) ds[r] *= ds_scale<HDP>();
      }
      if (dbias && qi < a.Tq) {
        if (jb + 4 <= a.Tk) {
          f4u t;
#pragma unroll
          for (int r = 0; r < 4; ++r) t.v[r] = ds[r] * ds_unscale;
          *reinterpret_cast<f4u*>(dbias + (long)qi * a.Tk + jb) = t;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (jb + r < a.Tk) dbias[(long)qi * a.Tk + jb + r] = ds[r] * ds_unscale;
        }
      }
      bf16x4 pp[3];
      split4s<NP, F16>(ds, pp);
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
      for (int q = 0; q < NP; ++q) sb[q] = frag<BKV>(myS + q * 16 * BKV, fb64, 0, ks);
#pragma unroll
      for (int di = 0; di < HDP / 16; ++di) {
        bf16x8 ka[3];
#pragma unroll
        for (int q = 0; q < NP; ++q) ka[q] = frag<BKV>(sKt + q * HDP * BKV, fb64, di * 16, ks);
        dqacc[di] = mfma_parts<NP, F16>(ka, sb, dqacc[di]);
      }
    }
  }
  // columns of dbias beyond the visited key tiles (mode 0, keys >= kv_len) are zero
  if (dbias && qi < a.Tq) {
    for (int j = ntiles * BKV + (lane >> 4); j < a.Tk; j += 4) dbias[(long)qi * a.Tk + j] = 0.f;
  }
  if (qi < a.Tq) {
    float* g = a.dq + ((long)b * a.Tq + qi) * ld + h * a.hd;
#pragma unroll
    for (int di = 0; di < HDP / 16; ++di) {
      const int d = di * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) if (d + r < a.hd) g[d + r] = dqacc[di][r] * (a.scale * ds_unscale * sc.ik);
    }
  }
}

// ------------------------------------------------------------------------------------------ backward dQ, hd = 64 fast path
// Same conditions and the same ideas as attn_fwd64_kernel (32 queries per wave, exp2-domain softmax with every scale
// folded into one FMA, dS kept in registers as the B fragments of the dQ product, interior tiles without mask code), plus:
// the K tile is kept in LDS ONCE, natural [key][d] with a row stride of 80 elements (40 dwords = 8 mod 64 banks), and
// read both ways -- 16-byte fragment reads for S^T = K Q^T, and gfx950's transposing ds_read_b64_tr_b16 for the K^T
// operand of dQ^T += K^T dS^T (lane 4q+p of a 16-lane group points at key row q / channels 4p..4p+3 and receives channel
// i's four keys; the two reads of a fragment take keys 32 ks + 4 g4 + {0..3} and + 16, the order dS sits in the
// registers).  No transposed K planes, no transposed tile in LDS.
// XL: XLNet's relative attention as in attn_fwd64_kernel<true> (position scores prefetched into LDS by LDS-DMA, XLNet
// mask, probability dropout) plus the dS store the position-term gradients are derived from (a.dbias, [B,H,Tq,Tk])
// delta_i = dO_i . O_i of query qi, head h (64 channels), as the four lanes {l, l^16, l^32, l^48} of the fast-path kernels
// compute it: lane group g4 sums channels ks*32 + 8 g4 .. +7, the groups are combined as (g0 + g1) + (g2 + g3).
// ONE function for attn_bwd_dq64_kernel and attn_delta64_kernel: both must produce the same bits (a captured step may take
// delta from the latter, the eager step from the former; tests/test_graph_gpu.py compares them bit for bit).
__device__ __forceinline__ float delta64(const AttnArgs& a, int b, int h, int qi, int g4) {
  float delta_i = 0.f;
  if (qi < a.Tq) {
    const float* po = a.o_in + ((long)b * a.Tq + qi) * a.C + h * 64;
    const float* pg = a.dout + ((long)b * a.Tq + qi) * a.C + h * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int d = ks * 32 + g4 * 8 + hh * 4;
        const float4 x = *reinterpret_cast<const float4*>(po + d), y = *reinterpret_cast<const float4*>(pg + d);
        delta_i += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      }
  }
  return quad16_sum(delta_i);
}

// delta alone (16 queries per wave, the lane layout of the kernels above): lets attn_bwd_dkdv64 start without waiting for
// attn_bwd_dq64 when the two run on different streams (launch_bwd: fork)
__global__ __launch_bounds__(256) void attn_delta64_kernel(AttnArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qi = blockIdx.x * 64 + wave * 16 + (lane & 15);
  const float d = delta64(a, b, h, qi, lane >> 4);
  if ((lane >> 4) == 0 && qi < a.Tq) a.delta[(long)(b * a.H + h) * a.Tq + qi] = d;
}

template <bool XL>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_bwd_dq64_kernel(AttnArgs a) {
  if (a.drop_thresh) a.drop_seed = vilco_step_seed(a.drop_seed, a.seed_word);
  constexpr int HDP = 64, BKV = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sK = reinterpret_cast<__bf16*>(smem_raw);          // [2 parts][64 keys][RS64]
  __bf16* sV = sK + 2 * PL64;                                 // [2 parts][64 keys][RS64]
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int bh = b * a.H + h;
  const int q0 = blockIdx.x * F64_Q + wave * 32;
  const int len = a.mode == 2 ? a.Tk : a.kv_len[b];
  const int kend = XL ? a.Tk : (len < a.Tk ? len : a.Tk);
  const int ntiles = (kend + BKV - 1) / BKV;
  const long row_bh = (long)bh * a.Tq;
  const AttnScales sc = *a.sc;
  const float c2 = a.scale * sc.iq * sc.ik * 1.44269504088896340736f;      // log2-domain score = acc * c2
  const float ds_unscale = DS_INV * sc.ido * sc.iv * (XL ? a.drop_inv_keep : 1.f);      // dS = dS' * this
  const int bias_ld = a.Tq + a.Tk;
  const float* bias = XL ? a.bias + row_bh * bias_ld : nullptr;
  float* dbias = (XL && a.dbias) ? a.dbias + row_bh * a.Tk : nullptr;
  [[maybe_unused]] _Float16* dsp = (XL && a.dsp) ? a.dsp + (long)bh * a.ds_batch_stride : nullptr;
  constexpr int RSBF = 68;                                   // see attn_fwd64_kernel
  float* sBias = reinterpret_cast<float*>(sV + 2 * PL64) + wave * 32 * RSBF;
  const int q0s = __builtin_amdgcn_readfirstlane(q0);
  const unsigned sbias_lds = XL ? (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr_of(sBias)) : 0u;
  [[maybe_unused]] auto bias_dma = [&](int k0) {
    const int j = k0 + lane;
    xl_dma_rows<32>(bias, bias_ld, a.Tq, q0s, 4u * (unsigned)(j < a.Tk ? j : a.Tk - 1), sbias_lds);
  };
  const __bf16* knb = a.kn.p + (long)bh * a.kn.batch_stride;
  const __bf16* vnb = a.vn.p + (long)bh * a.vn.batch_stride;
  const int g4 = lane >> 4;
  const int fbn = (lane & 15) * RS64 + g4 * 8;                          // natural fragment: row lane & 15, chunk g4 (+ 4 ks)
  const int fbt = (4 * g4 + ((lane & 15) >> 2)) * RS64 + 4 * (lane & 3);   // transposing fragment (see above)

  QFrag<HDP, 2> qf[2], dof[2];
  float lse2[2], dlt[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    load_qfrag<HDP, 2>(qf[g], a.qn, a.qn.p + (long)bh * a.qn.batch_stride, q0 + 16 * g, lane);
    load_qfrag<HDP, 2>(dof[g], a.don, a.don.p + (long)bh * a.don.batch_stride, q0 + 16 * g, lane);
    const int qi = q0 + 16 * g + (lane & 15);
    // P * 2^-22 = exp2(acc * c2 - lse2): natural-log lse to the log2 domain, the fp16 range shift of dS folded in
    lse2[g] = qi < a.Tq ? a.lse[row_bh + qi] * 1.44269504088896340736f + 22.f : 0.f;
    // delta_i = dO_i . O_i in exact fp32 (lane group g4 covers channels ks*32 + 8 g4 .. +7); also left for attn_bwd_dkdv
    const float delta_i = delta64(a, b, h, qi, g4);
    if (g4 == 0 && qi < a.Tq) a.delta[row_bh + qi] = delta_i;
    dlt[g] = (delta_i * sc.sdo) * sc.sv;                   // plane units; never form sdO * sV (see attn_bwd_dq_kernel)
    if constexpr (XL) dlt[g] /= a.drop_inv_keep;           // dS = inv_keep P (M dP - (1-p) delta)
  }

  f32x4 dqacc[2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) dqacc[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] uint32_t rkey[2] = {0u, 0u};               // XL + dropout: this lane's two rows of the mask (common.h)
  if constexpr (XL) {
    if (a.drop_thresh) { rkey[0] = drop_row(a, bh, q0 + (lane & 15)); rkey[1] = drop_row(a, bh, q0 + 16 + (lane & 15)); }
  }

  // staging: thread owns chunks (row = tid >> 3 (+32), 16-byte chunk c = tid & 7) of the K and the V tile
  bf16x8 stK[2][2], stV[2][2];
  const int srow = tid >> 3, sc8 = tid & 7;
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int kr = k0 + srow + 32 * i;
      kr = kr < a.kn.rows ? kr : a.kn.rows - 1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        stK[q][i] = *reinterpret_cast<const bf16x8*>(knb + q * a.kn.part_stride + (long)kr * HDP + sc8 * 8);
        stV[q][i] = *reinterpret_cast<const bf16x8*>(vnb + q * a.vn.part_stride + (long)kr * HDP + sc8 * 8);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = (srow + 32 * i) * RS64 + sc8 * 8;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<bf16x8*>(sK + q * PL64 + o) = stK[q][i];
        *reinterpret_cast<bf16x8*>(sV + q * PL64 + o) = stV[q][i];
      }
    }
  };
  const f32x2 c2v = {c2, c2};
  [[maybe_unused]] float am_ds = 0.f;               // XL: max |dS| this thread stored (for the relshift pack's scale)
  [[maybe_unused]] uint32_t ds_carry[2][2] = {{0u, 0u}, {0u, 0u}};      // XL planes: [group][part] the row stream's previous dword

  auto tile = [&](int t, bool more, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int k0 = t * BKV;
    // XL: this tile's position scores (issued during the previous tile) have landed; waited for HERE, where the K / V
    // registers need every outstanding load anyway -- not behind the next tile's prefetch just below
    if constexpr (XL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                // previous tile fully consumed
    lstore();
    __syncthreads();
    if (more) gload(k0 + BKV);

    bf16x8 dsf[2][2][2];                            // [group][32-key half][part]: dS^T B fragments, built in registers
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      f32x4 s[2][2], dp[2][2];                      // [group][16-key block bb of this half]
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int mi = 2 * kh + bb;
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int o = fbn + mi * 16 * RS64 + ks * 32;
          const f16x8 k0h = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(sK + o));
          const f16x8 k1h = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(sK + PL64 + o));
          const f16x8 v0h = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(sV + o));
          const f16x8 v1h = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(sV + PL64 + o));
          const f16x8 qa0 = __builtin_bit_cast(f16x8, qf[0].f[ks][0]), qa1 = __builtin_bit_cast(f16x8, qf[0].f[ks][1]);
          const f16x8 qb0 = __builtin_bit_cast(f16x8, qf[1].f[ks][0]), qb1 = __builtin_bit_cast(f16x8, qf[1].f[ks][1]);
          const f16x8 oa0 = __builtin_bit_cast(f16x8, dof[0].f[ks][0]), oa1 = __builtin_bit_cast(f16x8, dof[0].f[ks][1]);
          const f16x8 ob0 = __builtin_bit_cast(f16x8, dof[1].f[ks][0]), ob1 = __builtin_bit_cast(f16x8, dof[1].f[ks][1]);
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, qa0, s0, 0, 0, 0);      // S^T  = K Q^T
          s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, qb0, s1, 0, 0, 0);
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1h, oa0, d0, 0, 0, 0);      // dP^T = V dO^T
          d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1h, ob0, d1, 0, 0, 0);
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, qa1, s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, qb1, s1, 0, 0, 0);
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0h, oa1, d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0h, ob1, d1, 0, 0, 0);
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, qa0, s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, qb0, s1, 0, 0, 0);
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0h, oa0, d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0h, ob0, d1, 0, 0, 0);
        }
        if constexpr (MASKED && !XL) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + mi * 16 + g4 * 4 + r >= kend) { s0[r] = -INFINITY; s1[r] = -INFINITY; }
        }
        s[0][bb] = s0; s[1][bb] = s1; dp[0][bb] = d0; dp[1][bb] = d1;
      }
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x2 lv = {-lse2[g], -lse2[g]}, dv = {-dlt[g], -dlt[g]};
        [[maybe_unused]] const int qi_g = q0 + 16 * g + (lane & 15);
        u32x4 h0, h1;
        [[maybe_unused]] float dsv[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f32x2 sv = {s[g][e >> 2][e & 3], s[g][e >> 2][(e & 3) + 1]};
          f32x2 pv = {dp[g][e >> 2][e & 3], dp[g][e >> 2][(e & 3) + 1]};
          f32x2 arg = sv * c2v + lv;                       // v_pk_fma_f32
          [[maybe_unused]] const int j = k0 + (2 * kh + (e >> 2)) * 16 + g4 * 4 + (e & 3);
          if constexpr (XL) {
            const float* bp = sBias + (16 * g + (lane & 15)) * RSBF + (j - k0);
            const float bsc = a.scale * 1.44269504088896340736f;
            arg += f32x2{bp[0], bp[1]} * bsc;                                     // + scale * bd[i][Tq - i + j]
            if constexpr (MASKED) {
              if ((j >= len && j != qi_g) || j >= a.Tk) arg[0] = -INFINITY;
              if ((j + 1 >= len && j + 1 != qi_g) || j + 1 >= a.Tk) arg[1] = -INFINITY;
            }
            if (a.drop_thresh) {                                                   // dP of a dropped probability is zero
              const uint32_t jw = (uint32_t)j * VILCO_ATTN_DROP_W;
              pv[0] = vilco_attn_drop_keep_w(rkey[g], jw, a.drop_thresh) ? pv[0] : 0.f;
              pv[1] = vilco_attn_drop_keep_w(rkey[g], jw + VILCO_ATTN_DROP_W, a.drop_thresh) ? pv[1] : 0.f;
            }
          }
          const f32x2 p = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};      // P * 2^-22
          const f32x2 d = p * (pv + dv);                   // dS' = P (dP - delta) 2^-22, |dS'| < 2^15
          if constexpr (XL) if (!dsp) {                    // dS of this lane's 4 consecutive keys for the position-term gradients:
            const float v0 = d[0] * ds_unscale, v1 = d[1] * ds_unscale;      // ONE 16-byte store per two pairs (4-byte aligned: fine)
            if ((e & 3) == 0) { dsv[0] = v0; dsv[1] = v1; }
            else if (qi_g < a.Tq) {
              float* dst = dbias + (long)qi_g * a.Tk + (j - 2);
              if (j + 1 < a.Tk) {
                *reinterpret_cast<float4*>(dst) = make_float4(dsv[0], dsv[1], v0, v1);
                am_ds = fmaxf(fmaxf(am_ds, fmaxf(fabsf(dsv[0]), fabsf(dsv[1]))), fmaxf(fabsf(v0), fabsf(v1)));
              } else {
                const float vs[4] = {dsv[0], dsv[1], v0, v1};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (j - 2 + r < a.Tk) { dst[r] = vs[r]; am_ds = fmaxf(am_ds, fabsf(vs[r])); }
              }
            }
          }
          const f16x2 hp = {(_Float16)d[0], (_Float16)d[1]};
          const uint32_t hpu = __builtin_bit_cast(uint32_t, hp);
          uint32_t lo;
          asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hpu), "v"(d[0]));
          asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hpu), "v"(d[1]));
          h0[e >> 1] = hpu;
          h1[e >> 1] = lo;
        }
        if constexpr (XL) {
          // dS' as the operand planes of the unshifted view (a.dsp): this lane's 4 consecutive keys of a 16-key block are 4
          // consecutive columns Tq - i + j of row i, both halves exactly as they enter this kernel's own dQ product (value =
          // part sum * ds_unscale).  A row's stream of columns starts at Tq - i: on odd rows every 8-byte group would sit at a
          // 2-byte aligned address (measured: such stores made this kernel 40 % slower), so odd rows write the stream shifted by
          // one element -- each lane's first dword takes its low half from the stream's previous element (lane - 16 holds it:
          // ds_bpermute; for the first lane group it is the last element of the previous block, carried in ds_carry) -- at
          // 4-byte aligned addresses; the row's very last element is flushed after the key loop.  Tk % 64 == 0 (host check).
          if (dsp) {
            const bool odd = ((a.Tq - qi_g) & 1) != 0;
            _Float16* rowp = dsp + (long)qi_g * a.ds_ld + (a.Tq - qi_g) - (odd ? 1 : 0);
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
              for (int bb = 0; bb < 2; ++bb) {
                const uint32_t d0 = part ? h1[2 * bb] : h0[2 * bb], d1 = part ? h1[2 * bb + 1] : h0[2 * bb + 1];
                const uint32_t r = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane - 16) & 63) * 4, (int)d1);
                const uint32_t pred = g4 > 0 ? r : ds_carry[g][part];
                ds_carry[g][part] = r;
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 w = {odd ? __builtin_amdgcn_alignbit(d0, pred, 16) : d0, odd ? __builtin_amdgcn_alignbit(d1, d0, 16) : d1};
                if (qi_g < a.Tq)
                  *reinterpret_cast<u32x2*>(rowp + part * a.ds_plane_stride + (k0 + (2 * kh + bb) * 16 + g4 * 4)) = w;      // (nontemporal: 2x slower -- the 32-byte pieces need the L2's write combining)
              }
          }
        }
        dsf[g][kh][0] = __builtin_bit_cast(bf16x8, h0);
        dsf[g][kh][1] = __builtin_bit_cast(bf16x8, h1);
      }
    }
    if constexpr (XL) {
      if (more) {                                   // next tile's position scores (this wave has read its rows: LDS queue in order)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bias_dma(k0 + BKV);
      }
    }
    // dQ^T[d][q] += K^T dS^T, K^T fragments by transposing reads of the natural tile
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
      for (int di = 0; di < 4; ++di) {
        const int o = fbt + kh * 32 * RS64 + di * 16;
        const f16x8 k0h = __builtin_bit_cast(f16x8, tr_frag64(sK + o)), k1h = __builtin_bit_cast(f16x8, tr_frag64(sK + PL64 + o));
        const f16x8 a0 = __builtin_bit_cast(f16x8, dsf[0][kh][0]), a1 = __builtin_bit_cast(f16x8, dsf[0][kh][1]);
        const f16x8 b0 = __builtin_bit_cast(f16x8, dsf[1][kh][0]), b1 = __builtin_bit_cast(f16x8, dsf[1][kh][1]);
        f32x4 c0 = dqacc[0][di], c1 = dqacc[1][di];
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, a0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1h, b0, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, a1, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, b1, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, a0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0h, b0, c1, 0, 0, 0);
        dqacc[0][di] = c0; dqacc[1][di] = c1;
      }
    }
  };

  if (ntiles > 0) gload(0);
  if constexpr (XL) { if (ntiles > 0) bias_dma(0); }
  const int nfull = (XL ? (len < a.Tk ? len : a.Tk) : kend) / BKV;
  for (int t = 0; t < nfull; ++t) tile(t, t + 1 < ntiles, std::false_type{});
  for (int t = nfull; t < ntiles; ++t) tile(t, t + 1 < ntiles, std::true_type{});
  if constexpr (XL) {
    if (dsp && g4 == 0 && ntiles > 0) {              // odd rows: the stream's last element (key Tk - 1), left in the carry
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int qi = q0 + 16 * g + (lane & 15);
        if (qi < a.Tq && ((a.Tq - qi) & 1)) {
          _Float16* lastp = dsp + (long)qi * a.ds_ld + (a.Tq - qi) + (a.Tk - 1);      // 4-byte aligned; its upper half is out of band
          *reinterpret_cast<uint32_t*>(lastp) = ds_carry[g][0] >> 16;
          *reinterpret_cast<uint32_t*>(lastp + a.ds_plane_stride) = ds_carry[g][1] >> 16;
        }
      }
    }
  }

  const float oscale = a.scale * ds_unscale * sc.ik;
  float am = 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int qi = q0 + 16 * g + (lane & 15);
    if (qi < a.Tq) {
      float* gq = a.dq + ((long)b * a.Tq + qi) * a.C + h * HDP;
#pragma unroll
      for (int di = 0; di < 4; ++di) {
        const float4 v = make_float4(dqacc[g][di][0] * oscale, dqacc[g][di][1] * oscale, dqacc[g][di][2] * oscale, dqacc[g][di][3] * oscale);
        *reinterpret_cast<float4*>(gq + di * 16 + g4 * 4) = v;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
    }
  }
  if (a.am_dq) block_amax_out(am, a.am_dq, reinterpret_cast<float*>(smem_raw));
  if constexpr (XL) {
    if (a.am_ds) block_amax_out(am_ds, a.am_ds, reinterpret_cast<float*>(smem_raw));
    if (a.dsp && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && tid == 0) { a.ds_inv_scale[0] = ds_unscale; a.ds_inv_scale[1] = 1.f / ds_unscale; }
  }
}

// ------------------------------------------------------------------------------------------ backward dK / dV, hd = 64 fast path
// One workgroup per (b, h, 64 keys); a wave owns 16 keys for everything: its K and V fragments stay in registers for
// the whole kernel, S = Q K^T and dP = dO V^T for a 64-query tile leave lane (key, g4) with queries 16 mi + 4 g4 + r,
// which is (with the contraction index enumerated as in attn_fwd64_kernel) exactly the B fragment of
// dV^T += dO^T P and dK^T += Q^T dS: P and dS never leave the registers.  The Q and dO tiles sit in LDS once,
// natural layout (row stride 80), read by 16-byte fragment reads for S / dP and by the transposing ds_read_b64_tr_b16
// for the dO^T / Q^T operands -- no transposed planes of q and dO are packed for this path.
// XL: XLNet's relative attention.  The position scores of a (64-query tile, this workgroup's 64 keys) pair are 64 rows of
// 256 B in the unshifted matrix; they are fetched by LDS-DMA (16 rows per wave) one tile ahead into a double-buffered
// [64 q][68] LDS image that all four waves read (lane = key, 16 queries per lane and tile: scalar LDS reads).
template <bool XL>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_bwd_dkdv64_kernel(AttnArgs a) {
  if (a.drop_thresh) a.drop_seed = vilco_step_seed(a.drop_seed, a.seed_word);
  constexpr int HDP = 64, BQ = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* sQ = reinterpret_cast<__bf16*>(smem_raw);          // [2 parts][64 q][RS64]
  __bf16* sdO = sQ + 2 * PL64;                                // [2 parts][64 q][RS64]
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int bh = b * a.H + h;
  const int k0 = blockIdx.x * 64;
  const int len = a.mode == 2 ? a.Tk : a.kv_len[b];
  const int kend = len < a.Tk ? len : a.Tk;
  const long row_bh = (long)bh * a.Tq;
  const AttnScales sc = *a.sc;
  const float c2 = a.scale * sc.iq * sc.ik * 1.44269504088896340736f;      // log2-domain score = acc * c2
  const __bf16* qnb = a.qn.p + (long)bh * a.qn.batch_stride;
  const __bf16* donb = a.don.p + (long)bh * a.don.batch_stride;
  const int g4 = lane >> 4;
  const int fbn = (lane & 15) * RS64 + g4 * 8;
  const int fbt = (4 * g4 + ((lane & 15) >> 2)) * RS64 + 4 * (lane & 3);
  const int key = k0 + wave * 16 + (lane & 15);
  const float koff = (XL || key < kend) ? 0.f : -INFINITY;       // a masked key: every probability of this lane is exp2(-inf) = 0
  [[maybe_unused]] const uint32_t keyw = (uint32_t)key * VILCO_ATTN_DROP_W;      // dropout: this lane's key in the element hash
  const int bias_ld = a.Tq + a.Tk;
  const float* bias = XL ? a.bias + row_bh * bias_ld : nullptr;
  constexpr int RSBF = 68;
  float* sBias = reinterpret_cast<float*>(sdO + 2 * PL64);     // [2 stages][64 q][RSBF]
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const unsigned sbias_lds = XL ? (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr_of(sBias)) : 0u;
  const unsigned bias_voff = 4u * (unsigned)(k0 + lane < a.Tk ? k0 + lane : a.Tk - 1);
  [[maybe_unused]] auto bias_dma = [&](int q0, int stage) {     // this wave's 16 query rows of the tile
    xl_dma_rows<16>(bias, bias_ld, a.Tq, q0 + wave_s * 16, bias_voff, sbias_lds + (unsigned)(stage * 64 + wave_s * 16) * 272u);
  };
  const bool xl_edge = XL && (k0 + 64 > len || k0 + 64 > a.Tk);      // some key of this workgroup is beyond kv_len / Tk

  QFrag<HDP, 2> kf, vf;                                     // B fragments of this wave's 16 keys (rows >= Tk read as zero)
  load_qfrag<HDP, 2>(kf, a.kn, a.kn.p + (long)bh * a.kn.batch_stride, k0 + wave * 16, lane);
  load_qfrag<HDP, 2>(vf, a.vn, a.vn.p + (long)bh * a.vn.batch_stride, k0 + wave * 16, lane);

  f32x4 dvacc[4], dkacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { dvacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dkacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // staging: thread owns chunks (row = tid >> 3 (+32), 16-byte chunk c = tid & 7) of the Q and the dO tile; query rows
  // >= Tq are ZERO (their probabilities are garbage-but-finite and must meet zeros in dV / dK)
  bf16x8 stQ[2][2], stO[2][2];
  // lse (threads 0..63) and delta (64..127) of the tile's 64 queries take the same road as the Q / dO tile: one register per
  // thread, loaded a tile ahead, stored to LDS between the barriers.  Read from global memory inside the tile (round 4) they
  // were the youngest loads in flight, and the wait for them (vmcnt counts in order) drained the prefetch issued just before.
  __shared__ __attribute__((aligned(16))) float sLD[192];     // XL + dropout: [128..191] the rows' mask keys (common.h), as bits
  float stL = 0.f;
  const int srow = tid >> 3, sc8 = tid & 7;
  auto gload = [&](int q0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qr = q0 + srow + 32 * i;
      const bool ok = qr < a.Tq;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bf16x8 z = {};
        stQ[q][i] = ok ? *reinterpret_cast<const bf16x8*>(qnb + q * a.qn.part_stride + (long)qr * HDP + sc8 * 8) : z;
        stO[q][i] = ok ? *reinterpret_cast<const bf16x8*>(donb + q * a.don.part_stride + (long)qr * HDP + sc8 * 8) : z;
      }
    }
    if (tid < 128) {
      const int qi = q0 + (tid & 63);
      stL = qi < a.Tq ? (tid < 64 ? a.lse : a.delta)[row_bh + qi] : 0.f;      // queries >= Tq: 0 (their Q / dO rows are zero)
    } else if (XL && tid < 192 && a.drop_thresh) {
      stL = __uint_as_float(drop_row(a, bh, q0 + (tid & 63)));
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = (srow + 32 * i) * RS64 + sc8 * 8;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<bf16x8*>(sQ + q * PL64 + o) = stQ[q][i];
        *reinterpret_cast<bf16x8*>(sdO + q * PL64 + o) = stO[q][i];
      }
    }
    if (tid < (XL ? 192 : 128)) sLD[tid] = stL;
  };
  const f32x2 c2v = {c2, c2};
  constexpr float T37 = 7.2759576141834259e-12f;           // 2^-37: dS' = (P 2^15) ((dP - delta) 2^-37) = dS 2^-22 in plane units
  const f32x2 t37v = {T37, T37};
  const f16x8 k0h[2] = {__builtin_bit_cast(f16x8, kf.f[0][0]), __builtin_bit_cast(f16x8, kf.f[1][0])};
  const f16x8 k1h[2] = {__builtin_bit_cast(f16x8, kf.f[0][1]), __builtin_bit_cast(f16x8, kf.f[1][1])};
  const f16x8 v0h[2] = {__builtin_bit_cast(f16x8, vf.f[0][0]), __builtin_bit_cast(f16x8, vf.f[1][0])};
  const f16x8 v1h[2] = {__builtin_bit_cast(f16x8, vf.f[0][1]), __builtin_bit_cast(f16x8, vf.f[1][1])};

  const int nq = (XL || k0 < kend) ? (a.Tq + BQ - 1) / BQ : 0;     // every key of this tile masked: the gradients are zero
  if (nq > 0) gload(0);
  if constexpr (XL) { if (nq > 0) bias_dma(0, 0); }
  for (int t = 0; t < nq; ++t) {
    const int q0 = t * BQ;
    if constexpr (XL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of tile t's position scores has landed
    __syncthreads();                                // previous tile fully consumed (and every wave's DMA rows visible)
    lstore();
    __syncthreads();
    if (t + 1 < nq) {
      gload(q0 + BQ);
      if constexpr (XL) bias_dma(q0 + BQ, (t + 1) & 1);        // stage (t+1)&1 was last read in tile t-1: before the barrier above
    }
    [[maybe_unused]] const float* sB = sBias + (t & 1) * 64 * RSBF;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {                // 32 queries at a time: one k-step of the dV / dK products
      // lse and delta of this lane's 8 queries (16 bb + 4 g4 + r of this half), from the staged copy; queries >= Tq: 0
      float4 ls[2], dl[2];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        ls[bb] = *reinterpret_cast<const float4*>(sLD + 32 * kh + 16 * bb + 4 * g4);
        dl[bb] = *reinterpret_cast<const float4*>(sLD + 64 + 32 * kh + 16 * bb + 4 * g4);
      }
      f32x4 s[2], dp[2];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int mi = 2 * kh + bb;
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, d0 = s0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int o = fbn + mi * 16 * RS64 + ks * 32;
          const f16x8 q0h = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(sQ + o));
          const f16x8 q1h = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(sQ + PL64 + o));
          const f16x8 o0h = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(sdO + o));
          const f16x8 o1h = __builtin_bit_cast(f16x8, *reinterpret_cast<const bf16x8*>(sdO + PL64 + o));
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(q1h, k0h[ks], s0, 0, 0, 0);      // S  = Q K^T
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(o1h, v0h[ks], d0, 0, 0, 0);      // dP = dO V^T
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(q0h, k1h[ks], s0, 0, 0, 0);
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(o0h, v1h[ks], d0, 0, 0, 0);
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(q0h, k0h[ks], s0, 0, 0, 0);
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(o0h, v0h[ks], d0, 0, 0, 0);
        }
        s[bb] = s0; dp[bb] = d0;
      }
      u32x4 p0, p1, e0, e1;                         // fp16 x2 parts of P * 2^15 and of dS'
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const int bb = e >> 2, r = e & 3;
        const float la = r == 0 ? ls[bb].x : ls[bb].z, lb = r == 0 ? ls[bb].y : ls[bb].w;
        const float da = r == 0 ? dl[bb].x : dl[bb].z, db = r == 0 ? dl[bb].y : dl[bb].w;
        // P 2^15 = exp2(acc c2 - lse log2(e) + 15 [+ -inf for a masked key])
        const f32x2 lv = {__builtin_fmaf(la, -1.44269504088896340736f, 15.f) + koff, __builtin_fmaf(lb, -1.44269504088896340736f, 15.f) + koff};
        const f32x2 sv = {s[bb][r], s[bb][r + 1]};
        f32x2 arg = sv * c2v + lv;
        [[maybe_unused]] const int qa = q0 + 32 * kh + 16 * bb + 4 * g4 + r;        // this pair's queries qa, qa + 1
        [[maybe_unused]] float keep0 = 1.f, keep1 = 1.f;
        if constexpr (XL) {
          const int ql = 32 * kh + 16 * bb + 4 * g4 + r, jj = wave * 16 + (lane & 15);
          const float bsc = a.scale * 1.44269504088896340736f;
          arg += f32x2{sB[ql * RSBF + jj], sB[(ql + 1) * RSBF + jj]} * bsc;          // + scale * bd[i][Tq - i + j]
          if (xl_edge) {
            if ((key >= len && key != qa) || key >= a.Tk) arg[0] = -INFINITY;
            if ((key >= len && key != qa + 1) || key >= a.Tk) arg[1] = -INFINITY;
          }
          // a query row beyond Tq has zero Q / dO rows and lse = 0, but the position score of the clamped row is added:
          // keep its probability out (2^15 * 2^score would leave the fp16 range and meet the zeros as inf * 0)
          if (qa >= a.Tq) arg[0] = -INFINITY;
          if (qa + 1 >= a.Tq) arg[1] = -INFINITY;
          if (a.drop_thresh) {
            const float* rk = sLD + 128 + 32 * kh + 16 * bb + 4 * g4 + r;
            keep0 = vilco_attn_drop_keep_w(__float_as_uint(rk[0]), keyw, a.drop_thresh) ? 1.f : 0.f;
            keep1 = vilco_attn_drop_keep_w(__float_as_uint(rk[1]), keyw, a.drop_thresh) ? 1.f : 0.f;
          }
        }
        f32x2 p = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
        // (dP - delta) 2^-37, delta brought to plane units one scale at a time (see attn_bwd_dq_kernel)
        f32x2 dv = {(da * sc.sdo) * sc.sv, (db * sc.sdo) * sc.sv};
        f32x2 pv = {dp[bb][r], dp[bb][r + 1]};
        if constexpr (XL) { dv = dv / f32x2{a.drop_inv_keep, a.drop_inv_keep}; pv = pv * f32x2{keep0, keep1}; }
        const f32x2 d = p * ((pv - dv) * t37v);
        if constexpr (XL) p = p * f32x2{keep0, keep1};       // dV sees the dropped probabilities
        const f16x2 hp = {(_Float16)p[0], (_Float16)p[1]}, hd = {(_Float16)d[0], (_Float16)d[1]};
        const uint32_t hpu = __builtin_bit_cast(uint32_t, hp), hdu = __builtin_bit_cast(uint32_t, hd);
        uint32_t lp, ld;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hpu), "v"(p[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp) : "v"(hpu), "v"(p[1]));
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ld) : "v"(hdu), "v"(d[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(ld) : "v"(hdu), "v"(d[1]));
        p0[e >> 1] = hpu; p1[e >> 1] = lp; e0[e >> 1] = hdu; e1[e >> 1] = ld;
      }
      const f16x8 pb0 = __builtin_bit_cast(f16x8, p0), pb1 = __builtin_bit_cast(f16x8, p1);
      const f16x8 sb0 = __builtin_bit_cast(f16x8, e0), sb1 = __builtin_bit_cast(f16x8, e1);
      // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
      for (int di = 0; di < 4; ++di) {
        const int o = fbt + kh * 32 * RS64 + di * 16;
        const f16x8 o0h = __builtin_bit_cast(f16x8, tr_frag64(sdO + o)), o1h = __builtin_bit_cast(f16x8, tr_frag64(sdO + PL64 + o));
        const f16x8 q0h = __builtin_bit_cast(f16x8, tr_frag64(sQ + o)), q1h = __builtin_bit_cast(f16x8, tr_frag64(sQ + PL64 + o));
        f32x4 cv = dvacc[di], ck = dkacc[di];
        cv = __builtin_amdgcn_mfma_f32_16x16x32_f16(o1h, pb0, cv, 0, 0, 0);
        ck = __builtin_amdgcn_mfma_f32_16x16x32_f16(q1h, sb0, ck, 0, 0, 0);
        cv = __builtin_amdgcn_mfma_f32_16x16x32_f16(o0h, pb1, cv, 0, 0, 0);
        ck = __builtin_amdgcn_mfma_f32_16x16x32_f16(q0h, sb1, ck, 0, 0, 0);
        cv = __builtin_amdgcn_mfma_f32_16x16x32_f16(o0h, pb0, cv, 0, 0, 0);
        ck = __builtin_amdgcn_mfma_f32_16x16x32_f16(q0h, sb0, ck, 0, 0, 0);
        dvacc[di] = cv; dkacc[di] = ck;
      }
    }
  }

  // lane: key = k0 + wave*16 + (lane & 15), channels di*16 + 4 g4 + r
  float amk = 0.f, amv = 0.f;
  if (key < a.Tk) {
    float* gk = a.dk + ((long)b * a.Tk + key) * a.C + h * HDP;
    float* gv = a.dv + ((long)b * a.Tk + key) * a.C + h * HDP;
    const float ik = XL ? a.drop_inv_keep : 1.f;
    const float ksc = a.scale * (DS_INV * sc.ido * sc.iv) * sc.iq * ik, vsc = P_INV * sc.ido * ik;
#pragma unroll
    for (int di = 0; di < 4; ++di) {
      const int d = di * 16 + g4 * 4;
      const float4 vk = make_float4(dkacc[di][0] * ksc, dkacc[di][1] * ksc, dkacc[di][2] * ksc, dkacc[di][3] * ksc);
      const float4 vv = make_float4(dvacc[di][0] * vsc, dvacc[di][1] * vsc, dvacc[di][2] * vsc, dvacc[di][3] * vsc);
      *reinterpret_cast<float4*>(gk + d) = vk;
      *reinterpret_cast<float4*>(gv + d) = vv;
      amk = fmaxf(fmaxf(amk, fmaxf(fabsf(vk.x), fabsf(vk.y))), fmaxf(fabsf(vk.z), fabsf(vk.w)));
      amv = fmaxf(fmaxf(amv, fmaxf(fabsf(vv.x), fabsf(vv.y))), fmaxf(fabsf(vv.z), fabsf(vv.w)));
    }
  }
  if (a.am_dk) block_amax_out(amk, a.am_dk, reinterpret_cast<float*>(smem_raw));
  if (a.am_dv) block_amax_out(amv, a.am_dv, reinterpret_cast<float*>(smem_raw));
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// one workgroup per (b, h, 64-key tile); inner loop over 32-query tiles.  S = Q K^T orientation: the C layout
// gives each lane one key (col) and 4 consecutive queries (rows), so P^T / dS^T go to LDS as packed stores.
template <int HDP, int NP, bool F16, bool HASB, bool DROP>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dkdv_kernel(AttnArgs a) {
  if (a.drop_thresh) a.drop_seed = vilco_step_seed(a.drop_seed, a.seed_word);
  constexpr int BKV = 64, BQ = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // HDP > 128 (head dims 129..160): K and V tiles + the Q / dO tiles would need 180 KB of LDS.  The K fragments a wave
  // uses never change (its 32 keys, all k-steps), so they are read into registers once -- through the region the Q / dO
  // tiles take over afterwards -- and the kernel keeps 139 KB (both K and V in registers spilled).
  constexpr bool KV_REG = HDP > 128;
  __bf16* sV = reinterpret_cast<__bf16*>(smem_raw);          // [NP][64 keys][HDP]
  __bf16* sK = sV + NP * BKV * HDP;                           // [NP][64 keys][HDP]
  __bf16* sQ = KV_REG ? sK : sK + NP * BKV * HDP;             // [NP][32 q][HDP]
  __bf16* sdO = sQ + NP * BQ * HDP;                           // [NP][32 q][HDP]
  __bf16* sQt = sdO + NP * BQ * HDP;                          // [NP][HDP d][32 q]
  __bf16* sdOt = sQt + NP * HDP * BQ;                         // [NP][HDP d][32 q]
  __bf16* sPt = sdOt + NP * HDP * BQ;                         // [NP][64 keys][32 q]
  __bf16* sdSt = sPt + NP * BKV * BQ;                         // [NP][64 keys][32 q]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int bh = b * a.H + h;
  const long ld = a.C;
  const int k0 = kt * BKV;
  const int len = a.mode == 2 ? a.Tk : a.kv_len[b];
  const long row_bh = (long)bh * a.Tq;
  const int bias_ld = a.mode == 3 ? a.Tq + a.Tk : a.Tk;
  const float* bias = (HASB && a.bias) ? a.bias + row_bh * bias_ld : nullptr;
  const int mmode = a.mode == 3 ? 1 : a.mode;
  const __bf16* qnb = a.qn.p + (long)bh * a.qn.batch_stride;
  const __bf16* donb = a.don.p + (long)bh * a.don.batch_stride;
  const __bf16* qtb = a.qt.p + (long)bh * a.qt.batch_stride;
  const __bf16* dotb = a.dot.p + (long)bh * a.dot.batch_stride;
  const int fbH = toff<HDP>(lane & 15, lane >> 4);
  const int fb32 = toff<32>(lane & 15, lane >> 4);
  AttnScales sc = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  if (F16) sc = *a.sc;
  const float qk_scale = F16 ? a.scale * sc.iq * sc.ik : a.scale;
  const float ds_unscale = (F16 ? ds_inv<HDP>() * sc.ido * sc.iv : 1.f) * a.drop_inv_keep;

  {
    TileStage<HDP, BKV, NP> st;
    gload_tile<HDP, BKV, NP>(st, a.kn, a.kn.p + (long)bh * a.kn.batch_stride, k0, 0, tid);
    lstore_tile<HDP, BKV, NP>(st, sK, tid);
    gload_tile<HDP, BKV, NP>(st, a.vn, a.vn.p + (long)bh * a.vn.batch_stride, k0, 0, tid);
    lstore_tile<HDP, BKV, NP>(st, sV, tid);
  }

  // S / dP tiles: wave -> (m-tile of 16 queries = wave & 1, key half = wave >> 1 : 2 n-tiles of 16 keys)
  const int mq = wave & 1, kh = wave >> 1;
  bf16x8 kreg[KV_REG ? 2 : 1][KV_REG ? HDP / 32 : 1][3];
  if constexpr (KV_REG) {
    __syncthreads();
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int ks = 0; ks < HDP / 32; ++ks)
#pragma unroll
        for (int q = 0; q < NP; ++q) kreg[nj][ks][q] = frag<HDP>(sK + q * BKV * HDP, fbH, kh * 32 + nj * 16, ks);
  }
  // dV^T / dK^T accumulators: wave owns keys [wave*16, wave*16+16) (n-tile), all d (HDP/16 m-tiles)
  f32x4 dvacc[HDP / 16], dkacc[HDP / 16];
#pragma unroll
  for (int i = 0; i < HDP / 16; ++i) { dvacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dkacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const bool tile_dead = ((mmode == 0 || mmode == 4) && k0 >= len);      // every key of this tile is masked: grads are zero
  const int kfull = mmode == 4 ? 0 : (len < a.Tk ? len : a.Tk);
  const bool plain = (bias == nullptr) && (k0 + BKV <= kfull);
  int nq = tile_dead ? 0 : (a.Tq + BQ - 1) / BQ, qlo = 0;
  if (mmode == 4) {                                   // only the query tiles whose windows reach these 64 keys
    const int ilo = k0 - a.window, ihi = k0 + BKV - 1 + a.window;
    qlo = ilo > 0 ? ilo / BQ : 0;
    if (ihi / BQ + 1 < nq) nq = ihi / BQ + 1;
    if (nq < qlo) nq = qlo;
  }
  TileStage<HDP, BQ, NP> stQ, stdO;
  TileStage<BQ, HDP, NP> stQt, stdOt;
  // Additive bias of the [32 q][64 keys] tile.  In this kernel's orientation a lane owns one key and four query rows,
  // so direct loads are single dwords from four rows per instruction (0.7 ms per XLNet pass at P); instead every
  // thread brings 8 consecutive keys of one query row with the next Q tile (row-contiguous 16-byte loads) and the tile
  // goes through LDS.  It borrows the P^T buffer (free until this iteration's probabilities are written; exactly
  // 32 x 64 floats when NP >= 2), at the price of one extra barrier per iteration.
  constexpr bool BIAS_LDS = HASB && NP >= 2;      // HASB: instantiated separately so the bias-free kernel pays nothing
  float* sBias = reinterpret_cast<float*>(sPt);
  float stB[8];
  const int brow = tid >> 3, bcol = (tid & 7) * 8;
  auto gload_bias = [&](int q0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v4[4] = {0.f, 0.f, 0.f, 0.f};
      if (q0 + brow < a.Tq) bias4(v4, bias, q0 + brow, k0 + bcol + 4 * h, bias_ld, a);
#pragma unroll
      for (int e = 0; e < 4; ++e) stB[4 * h + e] = v4[e];
    }
  };
  if (nq > qlo) {
    if (BIAS_LDS && bias) gload_bias(qlo * BQ);
    gload_tile<HDP, BQ, NP>(stQ, a.qn, qnb, qlo * BQ, 0, tid);
    gload_tile<HDP, BQ, NP>(stdO, a.don, donb, qlo * BQ, 0, tid);
    gload_tile<BQ, HDP, NP>(stQt, a.qt, qtb, 0, qlo * BQ, tid);
    gload_tile<BQ, HDP, NP>(stdOt, a.dot, dotb, 0, qlo * BQ, tid);
  }
  for (int t = qlo; t < nq; ++t) {
    const int q0 = t * BQ;
    __syncthreads();
    lstore_tile<HDP, BQ, NP>(stQ, sQ, tid);
    lstore_tile<HDP, BQ, NP>(stdO, sdO, tid);
    lstore_tile<BQ, HDP, NP>(stQt, sQt, tid);
    lstore_tile<BQ, HDP, NP>(stdOt, sdOt, tid);
    if (BIAS_LDS && bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sBias[brow * 64 + bcol + e] = stB[e];
    }
    __syncthreads();
    if (t + 1 < nq) {
      if (BIAS_LDS && bias) gload_bias(q0 + BQ);
      gload_tile<HDP, BQ, NP>(stQ, a.qn, qnb, q0 + BQ, 0, tid);
      gload_tile<HDP, BQ, NP>(stdO, a.don, donb, q0 + BQ, 0, tid);
      gload_tile<BQ, HDP, NP>(stQt, a.qt, qtb, 0, q0 + BQ, tid);
      gload_tile<BQ, HDP, NP>(stdOt, a.dot, dotb, 0, q0 + BQ, tid);
    }

    // lane: key col = lane & 15 (per n-tile), query rows 4*(lane>>4) + r of m-tile mq
    float lse4[4], dl4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = q0 + mq * 16 + (lane >> 4) * 4 + r;
      lse4[r] = qi < a.Tq ? a.lse[row_bh + qi] : 0.f;
      const float dl_ = qi < a.Tq ? a.delta[row_bh + qi] : 0.f;
      dl4[r] = (F16 ? (dl_ * sc.sdo) * sc.sv : dl_) / a.drop_inv_keep;      // see attn_bwd_dq_kernel: never form sdO * sV
    }
    float bvt[2][4];
    if (bias) {
      if (BIAS_LDS) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            bvt[nj][r] = sBias[(mq * 16 + (lane >> 4) * 4 + r) * 64 + kh * 32 + nj * 16 + (lane & 15)];
        __syncthreads();                       // every wave has its bias before anyone writes P^T into the same buffer
      } else {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = q0 + mq * 16 + (lane >> 4) * 4 + r, j = k0 + kh * 32 + nj * 16 + (lane & 15);
            bvt[nj][r] = (qi < a.Tq && j < a.Tk) ? bias_at(bias, qi, j, bias_ld, a) : 0.f;
          }
      }
    }
    unsigned keep_bits = 0xffu;            // dropout keep flags of this lane's 8 (nj, r) probabilities
    if (DROP && a.drop_thresh) {
      keep_bits = 0u;
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          keep_bits |= (drop_keep(a, bh, q0 + mq * 16 + (lane >> 4) * 4 + r, k0 + kh * 32 + nj * 16 + (lane & 15)) != 0.f ? 1u : 0u)
                       << (nj * 4 + r);
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
          qa[q] = frag<HDP>(sQ + q * BQ * HDP, fbH, mq * 16, ks);
          da[q] = frag<HDP>(sdO + q * BQ * HDP, fbH, mq * 16, ks);
          if constexpr (KV_REG) kb[q] = kreg[nj][ks][q];
          else kb[q] = frag<HDP>(sK + q * BKV * HDP, fbH, ncol, ks);
          vb[q] = frag<HDP>(sV + q * BKV * HDP, fbH, ncol, ks);
        }
        s = mfma_parts<NP, F16>(qa, kb, s);        // S  = Q K^T
        dp = mfma_parts<NP, F16>(da, vb, dp);      // dP = dO V^T
      }
      const int j = k0 + ncol + (lane & 15);
      float p[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q0 + mq * 16 + (lane >> 4) * 4 + r;
        float x = s[r] * qk_scale;
        if (!plain) {
          x = mask_score(x + (bias ? bvt[nj][r] : 0.f), qi, j, len, a.Tk, mmode, a.window);
        }
        p[r] = (qi < a.Tq && x != -INFINITY) ? fast_exp(x - lse4[r]) : 0.f;
        const float mf = (!DROP || ((keep_bits >> (nj * 4 + r)) & 1u)) ? 1.f : 0.f;
        ds[r] = p[r] * (dp[r] * mf - dl4[r]);
        p[r] *= mf;                                      // dV sees the dropped probabilities
        if (F16) { ds[r] *= ds_scale<HDP>(); p[r] *= P_SCALE; }
      }
      bf16x4 pp[3], dd[3];
      split4s<NP, F16>(p, pp);
      split4s<NP, F16>(ds, dd);
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
      pb[q] = frag<BQ>(sPt + q * BKV * BQ, fb32, wave * 16, 0);
      sb[q] = frag<BQ>(sdSt + q * BKV * BQ, fb32, wave * 16, 0);
    }
#pragma unroll
    for (int di = 0; di < HDP / 16; ++di) {
      bf16x8 oa[3], qa[3];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        oa[q] = frag<BQ>(sdOt + q * HDP * BQ, fb32, di * 16, 0);
        qa[q] = frag<BQ>(sQt + q * HDP * BQ, fb32, di * 16, 0);
      }
      dvacc[di] = mfma_parts<NP, F16>(oa, pb, dvacc[di]);
      dkacc[di] = mfma_parts<NP, F16>(qa, sb, dkacc[di]);
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
          gk[d + r] = dkacc[di][r] * (a.scale * ds_unscale * sc.iq);
          gv[d + r] = dvacc[di][r] * ((F16 ? P_INV * sc.ido : 1.f) * a.drop_inv_keep);
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
size_t dkdv_lds() {        // HDP > 128: the K tile is staged through the Q / dO region (attn_bwd_dkdv_kernel: KV_REG)
  return (size_t)NP * ((HDP > 128 ? 64 * HDP : 2 * 64 * HDP) + 4 * 32 * HDP + 2 * 64 * 32) * sizeof(__bf16);
}

template <typename K>
void set_lds(K kernel, size_t bytes) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  (void)hipGetLastError();
}

bool fwd64_enabled();
// the MQ blocks' attention: hd = 64, fp16 x2 planes, prefix key mask, no bias / dBias, no dropout -> the *64 kernels
inline bool fast64(const AttnArgs& a, int precision) {
  return precision == 3 && a.hd == 64 && !a.bias && !a.dbias && (a.mode == 0 || a.mode == 2) && !a.drop_thresh && fwd64_enabled();
}
bool xl64_enabled();
// XLNet's relative attention (unshifted position scores as bias, XLNet mask, optional dropout) at hd = 64: attn_fwd64_kernel<true>
inline bool fast64_xl_fwd(const AttnArgs& a, int precision) {
  return precision == 3 && a.hd == 64 && a.bias && a.mode == 3 && a.Tq == a.Tk && fwd64_enabled() && xl64_enabled();
}
bool xl64_enabled() { static const bool on = [] { const char* e = getenv("VILCO_ATTN_XL_FAST"); return !(e && e[0] == '0'); }(); return on; }
bool fwd64_enabled() { static const bool on = [] { const char* e = getenv("VILCO_ATTN_FAST"); return !(e && e[0] == '0'); }(); return on; }

template <int HDP, int NP, bool F16 = false>
int launch_fwd(const AttnArgs& a, hipStream_t s) {
  if constexpr (HDP == 64 && NP == 2 && F16) {
    if (fast64(a, 3)) {
      size_t lds = 2 * 2 * PL64 * sizeof(__bf16);
#ifdef VILCO_LAB_ATTN
      if (const char* e = getenv("VILCO_LAB_ATTN_LDS")) {       // lab: pad the LDS request to limit workgroups per CU
        lds = (size_t)atoi(e);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      }
#endif
      hipLaunchKernelGGL(attn_fwd64_kernel<false>, dim3((a.Tq + F64_Q - 1) / F64_Q, a.H, a.B), dim3(ATT_THREADS), lds, s, a);
      return vilco_launch_status();
    }
    if (fast64_xl_fwd(a, 3)) {
      const size_t lds = 2 * 2 * PL64 * sizeof(__bf16) + 4 * 32 * 68 * sizeof(float);
      static const bool oncexl = [] { set_lds(&attn_fwd64_kernel<true>, 2 * 2 * PL64 * sizeof(__bf16) + 4 * 32 * 68 * sizeof(float)); return true; }();
      (void)oncexl;
      hipLaunchKernelGGL(attn_fwd64_kernel<true>, dim3((a.Tq + F64_Q - 1) / F64_Q, a.H, a.B), dim3(ATT_THREADS), lds, s, a);
      return vilco_launch_status();
    }
  }
  static const bool once = [] {
    set_lds(&attn_fwd_kernel<HDP, NP, F16, false>, fwd_lds<HDP, NP>());
    set_lds(&attn_fwd_kernel<HDP, NP, F16, true>, fwd_lds<HDP, NP>());
    return true;
  }();
  (void)once;
  dim3 grid((a.Tq + 63) / 64, a.H, a.B);
  const size_t lds = fwd_lds<HDP, NP>();
  if (a.drop_thresh) hipLaunchKernelGGL((attn_fwd_kernel<HDP, NP, F16, true>), grid, dim3(ATT_THREADS), lds, s, a);
  else hipLaunchKernelGGL((attn_fwd_kernel<HDP, NP, F16, false>), grid, dim3(ATT_THREADS), lds, s, a);
  return vilco_launch_status();
}

struct ForkCtx { hipStream_t s2; hipEvent_t e0, e1; };

// the side stream + events of the dQ || dK-dV fork, or null: VILCO_ATTN_FORK != 1 (the default: same-box A/B of the replayed P
// step, round 4: 25.16 / 25.23 ms without, 25.27 / 25.12 ms with -- no gain), or `s` is not capturing.  The stream and the
// events are created by the first call that finds `s` NOT capturing (resource creation inside a capture is not safe in every
// capture mode); a process's first backward is always eager (graph.py captures after eager_steps >= 1 iterations).
inline ForkCtx* attn_fork(hipStream_t s) {
  static const bool enabled = [] { const char* e = getenv("VILCO_ATTN_FORK"); return e && e[0] == '1'; }();      // off by default
  if (!enabled) return nullptr;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) return nullptr;
  static ForkCtx ctx;
  static int state = 0;                 // 0: not created, 1: ready, -1: creation failed
  if (st != hipStreamCaptureStatusActive) {
    if (state == 0)
      state = (hipStreamCreateWithFlags(&ctx.s2, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&ctx.e0, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&ctx.e1, hipEventDisableTiming) == hipSuccess) ? 1 : -1;
    return nullptr;
  }
  return state == 1 ? &ctx : nullptr;
}

template <int HDP, int NP, bool F16 = false>
int launch_bwd(const AttnArgs& a, hipStream_t s) {
  static const bool once = [] {
    set_lds(&attn_bwd_dq_kernel<HDP, NP, F16, false>, dq_lds<HDP, NP>());
    set_lds(&attn_bwd_dq_kernel<HDP, NP, F16, true>, dq_lds<HDP, NP>());
    set_lds(&attn_bwd_dkdv_kernel<HDP, NP, F16, false, false>, dkdv_lds<HDP, NP>());
    set_lds(&attn_bwd_dkdv_kernel<HDP, NP, F16, true, false>, dkdv_lds<HDP, NP>());
    set_lds(&attn_bwd_dkdv_kernel<HDP, NP, F16, false, true>, dkdv_lds<HDP, NP>());
    set_lds(&attn_bwd_dkdv_kernel<HDP, NP, F16, true, true>, dkdv_lds<HDP, NP>());
    return true;
  }();
  (void)once;
  dim3 gq((a.Tq + 63) / 64, a.H, a.B), gk((a.Tk + 63) / 64, a.H, a.B);
  const size_t lq = dq_lds<HDP, NP>(), lk = dkdv_lds<HDP, NP>();
  (void)attn_fork(s);                   // (creates the fork resources on the first eager call)
  bool fast = false;
  if constexpr (HDP == 64 && NP == 2 && F16)
    fast = fast64(a, 3);
  if (fast) {
    if constexpr (HDP == 64 && NP == 2 && F16) {
      constexpr size_t l64 = 2 * 2 * PL64 * sizeof(__bf16);
      const dim3 gq64((a.Tq + F64_Q - 1) / F64_Q, a.H, a.B);
      // dQ and dK/dV are independent once delta exists.  At the long levels neither grid fills the chip in whole rounds
      // (T = 2304, B H = 32: 576 and 1152 workgroups on 512 resident slots -> 2 and 3 rounds for 1.125 and 2.25 rounds of
      // work); on two streams the runtime packs them together.  Only inside a stream capture (a replayed graph has no host
      // in the loop; an eager second queue made step times erratic, DESIGN_LOG.md 3.6) and only where there is a tail to fill.
      ForkCtx* f = (long)gq64.x * gq64.y * gq64.z > 512 ? attn_fork(s) : nullptr;
      if (f) {
        hipLaunchKernelGGL(attn_delta64_kernel, dim3((a.Tq + 63) / 64, a.H, a.B), dim3(256), 0, s, a);
        hipEventRecord(f->e0, s);
        hipStreamWaitEvent(f->s2, f->e0, 0);
        hipLaunchKernelGGL(attn_bwd_dq64_kernel<false>, gq64, dim3(ATT_THREADS), l64, s, a);      // (writes the same delta again)
        hipLaunchKernelGGL(attn_bwd_dkdv64_kernel<false>, gk, dim3(ATT_THREADS), l64, f->s2, a);
        hipEventRecord(f->e1, f->s2);
        hipStreamWaitEvent(s, f->e1, 0);
        return vilco_launch_status();
      }
      hipLaunchKernelGGL(attn_bwd_dq64_kernel<false>, gq64, dim3(ATT_THREADS), l64, s, a);
      hipLaunchKernelGGL(attn_bwd_dkdv64_kernel<false>, gk, dim3(ATT_THREADS), l64, s, a);
      return vilco_launch_status();
    }
  }
  bool xl_dq = false;                               // XLNet's relative attention: dQ + dS on the fast-path structure
  if constexpr (HDP == 64 && NP == 2 && F16) xl_dq = fast64_xl_fwd(a, 3) && (a.dbias != nullptr || a.dsp != nullptr);
  if (xl_dq) {
    if constexpr (HDP == 64 && NP == 2 && F16) {
      constexpr size_t lxl = 2 * 2 * PL64 * sizeof(__bf16) + 4 * 32 * 68 * sizeof(float);
      static const bool oncexl = [] { set_lds(&attn_bwd_dq64_kernel<true>, 2 * 2 * PL64 * sizeof(__bf16) + 4 * 32 * 68 * sizeof(float)); return true; }();
      (void)oncexl;
      hipLaunchKernelGGL(attn_bwd_dq64_kernel<true>, dim3((a.Tq + F64_Q - 1) / F64_Q, a.H, a.B), dim3(ATT_THREADS), lxl, s, a);
      constexpr size_t lkx = 2 * 2 * PL64 * sizeof(__bf16) + 2 * 64 * 68 * sizeof(float);
      static const bool oncekx = [] { set_lds(&attn_bwd_dkdv64_kernel<true>, 2 * 2 * PL64 * sizeof(__bf16) + 2 * 64 * 68 * sizeof(float)); return true; }();
      (void)oncekx;
      hipLaunchKernelGGL(attn_bwd_dkdv64_kernel<true>, gk, dim3(ATT_THREADS), lkx, s, a);
      return vilco_launch_status();
    }
  } else if (a.drop_thresh) hipLaunchKernelGGL((attn_bwd_dq_kernel<HDP, NP, F16, true>), gq, dim3(ATT_THREADS), lq, s, a);
  else hipLaunchKernelGGL((attn_bwd_dq_kernel<HDP, NP, F16, false>), gq, dim3(ATT_THREADS), lq, s, a);
  const bool dr = a.drop_thresh != 0;
  if (a.bias && dr) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<HDP, NP, F16, true, true>), gk, dim3(ATT_THREADS), lk, s, a);
  else if (a.bias) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<HDP, NP, F16, true, false>), gk, dim3(ATT_THREADS), lk, s, a);
  else if (dr) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<HDP, NP, F16, false, true>), gk, dim3(ATT_THREADS), lk, s, a);
  else hipLaunchKernelGGL((attn_bwd_dkdv_kernel<HDP, NP, F16, false, false>), gk, dim3(ATT_THREADS), lk, s, a);
  return vilco_launch_status();
}

template <int HDP>
int dispatch(const AttnArgs& a, int precision, bool bwd, hipStream_t s) {
  if (precision == 1) return bwd ? launch_bwd<HDP, 1>(a, s) : launch_fwd<HDP, 1>(a, s);
  if (precision == 0) return bwd ? launch_bwd<HDP, 2>(a, s) : launch_fwd<HDP, 2>(a, s);
  if (precision == 3) return bwd ? launch_bwd<HDP, 2, true>(a, s) : launch_fwd<HDP, 2, true>(a, s);
  if constexpr (HDP <= 64) return bwd ? launch_bwd<HDP, 3>(a, s) : launch_fwd<HDP, 3>(a, s);
  else return VILCO_ERR_UNSUPPORTED;               // three bf16 planes of a wider tile exceed the LDS (check_common)
}

int check_common(int B, int H, int Tq, int Tk, int hd, int mode, int precision, int window = 0) {
  if (B < 0 || H <= 0 || Tq < 0 || Tk < 0 || hd <= 0) return VILCO_ERR_BADARG;
  if (mode < 0 || mode > 4 || precision < 0 || precision > 3 || (mode == 4 && (window < 0 || Tq != Tk))) return VILCO_ERR_BADARG;
  if (hd > 160 || (hd % 4) != 0) return VILCO_ERR_UNSUPPORTED;      // head dims 4..160 (P: 64, cfg1: 128, W: 144, tests: 8, 16, 32)
  if (hd > 64 && precision == 2) return VILCO_ERR_UNSUPPORTED;      // three bf16 planes of a 128-wide tile exceed the 160 KB LDS (dkdv)
  return VILCO_OK;
}

inline long up(long x, long a) { return (x + a - 1) / a * a; }
inline int hdp_of(int hd) { return hd <= 32 ? 32 : (hd <= 64 ? 64 : (hd <= 128 ? 128 : 160)); }   // padded head width of the tiles
inline int np_of(int precision) { return precision == 1 ? 1 : ((precision == 0 || precision == 3) ? 2 : 3); }
constexpr long ATT_SCALE_BYTES = 4 * AMAX_MAX_BLOCKS * 4 + 256;    // fp16 x2: amax partials of q, k, v, dO + AttnScales

// one operand -> bf16 planes.  natural: [part][B*H][T][HDP]; transposed: [part][B*H][hd][Tp]
struct PlaneSpec { long elems_per_part; long batch; int row_stride, rows, cols; };

PlaneSpec spec_nat(int B, int H, int T, int HDP) {
  PlaneSpec p; p.batch = (long)T * HDP; p.elems_per_part = p.batch * B * H; p.row_stride = HDP; p.rows = T; p.cols = HDP;
  return p;
}
PlaneSpec spec_tr(int B, int H, int T, int hd) {
  const int Tp = (int)up(T, 32);
  PlaneSpec p; p.batch = (long)hd * Tp; p.elems_per_part = p.batch * B * H; p.row_stride = Tp; p.rows = hd; p.cols = Tp;
  return p;
}

// runs the pack kernel for x [B,T,C] (head slices) into `dst` and returns the Planes view + advanced pointer
// natural (kc) packs queued here go out together in ONE launch (flush_packs)
struct PackQueue { PackArgs4 a; int n = 0; PackArgs4 t; int nt = 0; };

// optnone: hipcc 7.2's -O3 HOST code for this function queues pack descriptors whose amax pointer is not the one
// passed in (non-fp16 modes then packed fp16 planes; found by bisection with per-function optnone, r02) -- it is a
// dozen scalar assignments per call, so nothing is lost
__attribute__((optnone)) Planes pack_operand(const float* x, __bf16*& dst, const PlaneSpec& sp, bool tr, int B, int H, int T, int hd, int HDP,
                    int NP, hipStream_t s, const float* amax = nullptr, int namax = 0, float* inv_scale = nullptr,
                    PackQueue* queue = nullptr) {
  PackArgs pa;
  pa.src = x; pa.dst = dst; pa.ld = (long)H * hd;
  pa.rows = tr ? hd : T; pa.K = tr ? T : hd; pa.Kp = tr ? sp.row_stride : HDP;
  pa.plane_stride = sp.elems_per_part; pa.batch_stride = sp.batch; pa.nbi = H;
  pa.so = (long)T * H * hd; pa.si = hd;
  pa.tap = 0; pa.tapC = 1; pa.tapT = 1; pa.out_rows = T;
  pa.vec = vilco_aligned(x, 16) && (hd % 4) == 0;
  pa.amax = amax; pa.namax = namax; pa.inv_scale = inv_scale;
  if (queue && !tr) queue->a.a[queue->n++] = pa;
  else if (queue) queue->t.a[queue->nt++] = pa;
  else dispatch_pack(NP, pa, tr, B * H, s);
  Planes pl;
  pl.p = dst; pl.part_stride = sp.elems_per_part; pl.batch_stride = sp.batch;
  pl.row_stride = sp.row_stride; pl.rows = sp.rows; pl.cols = sp.cols;
  dst += up(sp.elems_per_part * NP, 128);
  return pl;
}

long planes_bytes(const PlaneSpec& sp, int NP) { return up(sp.elems_per_part * NP, 128) * 2; }

// fp16 x2: amax over the [B*T][C] operands (q, k, v[, dO]): the views are planned here and run either inside the fused
// kc-pack launch (flush_packs) or by one amax launch
struct ScaleWs {
  float* parts[4] = {nullptr, nullptr, nullptr, nullptr};
  int n[4] = {0, 0, 0, 0};
  bool ext[4] = {false, false, false, false};     // partials supplied by the caller (vilco_attn_amax_in): no amax pass
  float* out = nullptr;
  AmaxArgs am;
  int nops = 0;
};
ScaleWs plan_amax(unsigned char* region, const float* const (&x)[4], const int (&T)[4], int nops, int B, int C) {
  ScaleWs w;
  float* f = reinterpret_cast<float*>(region);
  w.out = f + 4 * AMAX_MAX_BLOCKS;
  w.nops = nops;
  for (int i = 0; i < nops; ++i) {
    PackArgs pa = {};
    pa.src = x[i]; pa.ld = C; pa.rows = B * T[i]; pa.K = C; pa.nbi = 1; pa.so = 0; pa.si = 0; pa.tap = 0;
    pa.vec = vilco_aligned(x[i], 16) && (C % 4) == 0;
    w.parts[i] = f + i * AMAX_MAX_BLOCKS;
    w.am.op[i] = amax_view(pa, false, 1, w.parts[i]);
    w.n[i] = w.am.op[i].nblocks;
  }
  return w;
}

void use_external_amax(ScaleWs& sw, const vilco_attn_amax_in* in, int nops) {
  if (!in) return;
  const float* p[4] = {in->q, in->k, in->v, in->dout};
  const int n[4] = {in->nq, in->nk, in->nv, in->ndo};
  for (int i = 0; i < nops; ++i)
    if (p[i] && n[i] > 0) { sw.parts[i] = const_cast<float*>(p[i]); sw.n[i] = n[i]; sw.ext[i] = true; }
}

// all queued packs of one attention call: [amax +] natural packs in one launch, transposing packs in another
void flush_packs(PackQueue& pq, ScaleWs& sw, bool f16, int NP, int nbatch, hipStream_t s) {
  if (f16) {
    PackArgs fa[4];
    AmaxOp fm[4];
    int nf = 0;
    bool covered[4] = {false, false, false, false};
    for (int i = 0; i < pq.n; ++i)
      for (int j = 0; j < sw.nops; ++j)
        if (pq.a.a[i].amax == sw.parts[j]) { fa[nf] = pq.a.a[i]; fm[nf] = sw.am.op[j]; covered[j] = true; ++nf; break; }
    bool ok = nf == pq.n;
    for (int j = 0; j < sw.nops && ok; ++j) {
      if (covered[j]) continue;
      if (nf == 4) { ok = false; break; }
      PackArgs d = {};                          // amax only: a pack of zero rows that still leaves {1/s, s}
      d.Kp = 8; d.nbi = 1; d.tapC = 1; d.tapT = 1; d.out_rows = 0;
      d.amax = sw.parts[j]; d.inv_scale = sw.out + 2 * j;
      fa[nf] = d; fm[nf] = sw.am.op[j]; ++nf;
    }
    for (int j = 0; j < sw.nops; ++j) ok = ok && !sw.ext[j];
    if (ok && dispatch_pack_fused(NP, fa, fm, nf, nbatch, s, VILCO_SITE_ATTNPACK)) {
      for (int i = 0; i < pq.nt; ++i) pq.t.a[i].namax = fa[0].namax;      // every operand got gx * nbatch partials
    } else {
      AmaxArgs am;
      int nam = 0;
      for (int j = 0; j < sw.nops; ++j) if (!sw.ext[j]) am.op[nam++] = sw.am.op[j];
      if (nam) launch_amax(am, nam, s);
      if (pq.n) dispatch_pack_multi(NP, pq.a, pq.n, s, nbatch);
    }
  } else if (pq.n) {
    dispatch_pack_multi(NP, pq.a, pq.n, s, nbatch);
  }
  if (pq.nt) dispatch_pack_tr_multi(NP, pq.t, pq.nt, s, nbatch);
}

}  // namespace

#ifdef VILCO_LAB_ATTN
extern "C" int vilco_lab_attn_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vilco_lab_attn_stamps), sizeof(unsigned long long) * 64 * 8);
}
#endif

extern "C" int vilco_attn_supported(int32_t hd) { return hd > 0 && hd <= 160 && (hd % 4) == 0; }

extern "C" int32_t vilco_attn_amax_parts(int32_t B, int32_t H, int32_t T, int32_t hd, int32_t mode, int32_t precision,
                                         int32_t has_bias, float drop_p, int32_t key_side) {
  AttnArgs a = {};
  a.hd = hd; a.mode = mode; a.drop_thresh = vilco_drop_threshold_host(drop_p);
  if (has_bias) a.bias = reinterpret_cast<const float*>(&a);      // only its nullness is looked at
  if (has_bias == 2) {           // the dS (dbias) partials of XLNet's relative attention: attn_bwd_dq64_kernel<true>, one per workgroup
    a.Tq = a.Tk = T;
    if (B <= 0 || H <= 0 || T <= 0 || key_side || !fast64_xl_fwd(a, precision)) return 0;
    const long nx = (long)((T + F64_Q - 1) / F64_Q) * H * B;
    return nx <= 8192 ? (int32_t)nx : 0;
  }
  if (B <= 0 || H <= 0 || T <= 0 || !fast64(a, precision)) return 0;
  const long n = (long)((T + (key_side ? 63 : F64_Q - 1)) / (key_side ? 64 : F64_Q)) * H * B;
  return n <= 8192 ? (int32_t)n : 0;
}

// 1 when vilco_attn_fwd_planes can write o's operand planes for this configuration (the hd = 64 forward kernels)
extern "C" int32_t vilco_attn_planes_supported(int32_t Tq, int32_t Tk, int32_t hd, int32_t mode, int32_t precision, int32_t has_bias,
                                               float drop_p) {
  AttnArgs a = {};
  a.hd = hd; a.mode = mode; a.Tq = Tq; a.Tk = Tk; a.drop_thresh = vilco_drop_threshold_host(drop_p);
  if (has_bias) a.bias = reinterpret_cast<const float*>(&a);      // only its nullness is looked at
  return (fast64(a, precision) || fast64_xl_fwd(a, precision)) ? 1 : 0;
}

extern "C" size_t vilco_attn_fwd_workspace(int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t hd, int32_t precision) {
  const int HDP = hdp_of(hd), NP = np_of(precision);
  return (size_t)(planes_bytes(spec_nat(B, H, Tq, HDP), NP) + planes_bytes(spec_nat(B, H, Tk, HDP), NP) +
                  planes_bytes(spec_tr(B, H, Tk, hd), NP) + 1024 + ATT_SCALE_BYTES);
}

extern "C" int vilco_attn_fwd(const float* q, const float* k, const float* v, const float* bias,
                              const int32_t* kv_len, float* o, float* lse, int32_t B, int32_t H, int32_t Tq,
                              int32_t Tk, int32_t hd, float scale, int32_t mode, int32_t window, int32_t precision, float drop_p,
                              uint32_t drop_seed, const vilco_attn_amax_in* amax_in, float* o_amax, void* workspace,
                              size_t workspace_bytes, void* stream) {
  return vilco_attn_fwd_planes(q, k, v, bias, kv_len, o, lse, B, H, Tq, Tk, hd, scale, mode, window, precision, drop_p, drop_seed,
                               amax_in, o_amax, workspace, workspace_bytes, nullptr, 0, stream);
}

extern "C" int vilco_attn_fwd_planes(const float* q, const float* k, const float* v, const float* bias,
                              const int32_t* kv_len, float* o, float* lse, int32_t B, int32_t H, int32_t Tq,
                              int32_t Tk, int32_t hd, float scale, int32_t mode, int32_t window, int32_t precision, float drop_p,
                              uint32_t drop_seed, const vilco_attn_amax_in* amax_in, float* o_amax, void* workspace,
                              size_t workspace_bytes, void* o_planes, size_t o_planes_bytes, void* stream) {
  int rc = check_common(B, H, Tq, Tk, hd, mode, precision, window);
  if (!(drop_p >= 0.f) || drop_p >= 1.f) return VILCO_ERR_BADARG;
  if (rc != VILCO_OK) return rc;
  if (!q || !k || !v || !o || !lse) return VILCO_ERR_BADARG;
  if (mode != 2 && !kv_len) return VILCO_ERR_BADARG;
  if (B == 0 || Tq == 0) return VILCO_OK;
  if (Tk == 0) return VILCO_ERR_BADARG;
  if (!workspace || workspace_bytes < vilco_attn_fwd_workspace(B, H, Tq, Tk, hd, precision)) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int HDP = hdp_of(hd), NP = np_of(precision);
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.bias = bias; a.o = o; a.lse = lse; a.kv_len = kv_len;
  a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.hd = hd; a.C = H * hd; a.scale = scale; a.mode = mode; a.window = window;
  a.drop_thresh = vilco_drop_threshold_host(drop_p); a.drop_seed = drop_seed; a.drop_inv_keep = 1.f / (1.f - drop_p); a.seed_word = vilco_seed_word_dev();
  if (o_amax && !fast64(a, precision)) return VILCO_ERR_UNSUPPORTED;      // see vilco_attn_amax_parts
  a.am_o = o_amax;
  if (o_planes) {                  // the hd = 64 forward kernels only (fast64 / fast64_xl_fwd)
    if (!(fast64(a, precision) || fast64_xl_fwd(a, precision)) || !vilco_aligned(o_planes, 256)) return VILCO_ERR_UNSUPPORTED;
    const long rows32 = ((long)B * Tq + 31) / 32 * 32;
    if (o_planes_bytes < (size_t)(VILCO_PACK_HDR + rows32 * (long)(H * hd) * 4)) return VILCO_ERR_WORKSPACE;
    unsigned char* u = reinterpret_cast<unsigned char*>(o_planes);
    a.op0 = reinterpret_cast<_Float16*>(u + VILCO_PACK_HDR);
    a.o_plane_stride = rows32 * (long)(H * hd);
    a.o_inv_scale = reinterpret_cast<float*>(u) + VILCO_AMAX_MAX_BLOCKS;
    a.o_rows32 = rows32;
    a.o_scale_mul = 1.f;
    for (float ik = a.drop_thresh ? a.drop_inv_keep : 1.f; ik > 1.f; ik *= 0.5f) a.o_scale_mul *= 0.5f;      // 2^-ceil(log2(1 / keep))
  }
  unsigned char* wsb = reinterpret_cast<unsigned char*>(up((long)reinterpret_cast<uintptr_t>(workspace), 256));
  ScaleWs sw;
  if (precision == 3) {
    const float* const xs[4] = {q, k, v, nullptr};
    const int Ts[4] = {Tq, Tk, Tk, 0};
    sw = plan_amax(wsb, xs, Ts, 3, B, H * hd);
    use_external_amax(sw, amax_in, 3);
    a.sc = reinterpret_cast<const AttnScales*>(sw.out);
  }
  float* so = sw.out;
  __bf16* w = reinterpret_cast<__bf16*>(wsb + ATT_SCALE_BYTES);
  PackQueue pq;
  a.qn = pack_operand(q, w, spec_nat(B, H, Tq, HDP), false, B, H, Tq, hd, HDP, NP, s, sw.parts[0], sw.n[0], so, &pq);
  a.kn = pack_operand(k, w, spec_nat(B, H, Tk, HDP), false, B, H, Tk, hd, HDP, NP, s, sw.parts[1], sw.n[1], so + 2, &pq);
  if (fast64(a, precision) || fast64_xl_fwd(a, precision))       // the hd = 64 fast kernels read V from its natural planes (transposing LDS reads)
    a.vn = pack_operand(v, w, spec_nat(B, H, Tk, HDP), false, B, H, Tk, hd, HDP, NP, s, sw.parts[2], sw.n[2], so + 4, &pq);
  else
    a.vt = pack_operand(v, w, spec_tr(B, H, Tk, hd), true, B, H, Tk, hd, HDP, NP, s, sw.parts[2], sw.n[2], so + 4, &pq);
  flush_packs(pq, sw, precision == 3, NP, B * H, s);
  return hd <= 32 ? dispatch<32>(a, precision, false, s) : (hd <= 64 ? dispatch<64>(a, precision, false, s) : (hd <= 128 ? dispatch<128>(a, precision, false, s) : dispatch<160>(a, precision, false, s)));
}

// XLNet position scores (xl_scores64_kernel): bd[B][H][T][2T] (fp32, only the band of every row is written) from qr [B][T][H*64]
// and kr [2T][H*64] (per_clip = 0) or [B][2T][H*64] (per_clip = 1).  precision 3 (fp16 x2 operand planes), hd = 64.
extern "C" size_t vilco_xl_scores_workspace(int32_t B, int32_t H, int32_t T, int32_t per_clip) {
  return (size_t)(planes_bytes(spec_nat(B, H, T, 64), 2) + planes_bytes(spec_nat(per_clip ? B : 1, H, 2 * T, 64), 2) + 2 * ATT_SCALE_BYTES + 1024);
}

extern "C" int vilco_xl_scores(const float* qr, const float* kr, float* bd, int32_t B, int32_t H, int32_t T, int32_t hd,
                               int32_t per_clip, int32_t precision, void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 0 || H <= 0 || T < 0 || !qr || !kr || !bd) return VILCO_ERR_BADARG;
  if (hd != 64 || precision != 3) return VILCO_ERR_UNSUPPORTED;
  if (B == 0 || T == 0) return VILCO_OK;
  if (!workspace || workspace_bytes < vilco_xl_scores_workspace(B, H, T, per_clip)) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  unsigned char* wsb = reinterpret_cast<unsigned char*>(up((long)reinterpret_cast<uintptr_t>(workspace), 256));
  const int Bk = per_clip ? B : 1;
  // max|qr|, max|kr| -> {1/s, s}: two amax views (their row counts differ), one launch; the packs fold the partials
  const float* const xq[4] = {qr, nullptr, nullptr, nullptr};
  const float* const xk[4] = {kr, nullptr, nullptr, nullptr};
  const int Tq_[4] = {T, 0, 0, 0}, Tk_[4] = {2 * T, 0, 0, 0};
  ScaleWs sq = plan_amax(wsb, xq, Tq_, 1, B, H * hd);
  ScaleWs sk = plan_amax(wsb + ATT_SCALE_BYTES, xk, Tk_, 1, Bk, H * hd);
  AmaxArgs am;
  am.op[0] = sq.am.op[0]; am.op[1] = sk.am.op[0];
  launch_amax(am, 2, s);
  // AttnScales {iq, sq, ik, sk, ...}: q's pair at sq.out, k's pair right behind it
  float* scales = sq.out;
  __bf16* w = reinterpret_cast<__bf16*>(wsb + 2 * ATT_SCALE_BYTES);
  AttnArgs a = {};
  a.B = B; a.H = H; a.Tq = T; a.Tk = T; a.hd = hd; a.C = H * hd; a.mode = per_clip ? 1 : 0;
  a.o = bd;
  a.qn = pack_operand(qr, w, spec_nat(B, H, T, 64), false, B, H, T, hd, 64, 2, s, sq.parts[0], sq.n[0], scales, nullptr);
  a.kn = pack_operand(kr, w, spec_nat(Bk, H, 2 * T, 64), false, Bk, H, 2 * T, hd, 64, 2, s, sk.parts[0], sk.n[0], scales + 2, nullptr);
  a.sc = reinterpret_cast<const AttnScales*>(scales);
  hipLaunchKernelGGL(xl_scores64_kernel, dim3((T + F64_Q - 1) / F64_Q, H, B), dim3(ATT_THREADS), 2 * PL64 * sizeof(__bf16), s, a);
  return vilco_launch_status();
}

extern "C" size_t vilco_attn_bwd_workspace(int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t hd, int32_t precision) {
  const int HDP = hdp_of(hd), NP = np_of(precision);
  long bytes = 2 * planes_bytes(spec_nat(B, H, Tq, HDP), NP) + 2 * planes_bytes(spec_tr(B, H, Tq, hd), NP) +
               2 * planes_bytes(spec_nat(B, H, Tk, HDP), NP) + planes_bytes(spec_tr(B, H, Tk, hd), NP);
  bytes += up((long)B * H * (Tq > 0 ? Tq : 1) * 4, 256) + 1024 + ATT_SCALE_BYTES;
  return (size_t)bytes;
}

extern "C" int vilco_attn_bwd(const float* q, const float* k, const float* v, const float* bias,
                              const int32_t* kv_len, const float* o, const float* lse, const float* dout,
                              float* dq, float* dk, float* dv, float* dbias, int32_t B, int32_t H, int32_t Tq,
                              int32_t Tk, int32_t hd, float scale, int32_t mode, int32_t window, int32_t precision, float drop_p,
                              uint32_t drop_seed, const vilco_attn_amax_in* amax_in, float* dq_amax, float* dk_amax,
                              float* dv_amax, float* dbias_amax, void* workspace, size_t workspace_bytes, void* stream) {
  return vilco_attn_bwd_dsplanes(q, k, v, bias, kv_len, o, lse, dout, dq, dk, dv, dbias, B, H, Tq, Tk, hd, scale, mode, window, precision,
                                 drop_p, drop_seed, amax_in, dq_amax, dk_amax, dv_amax, dbias_amax, workspace, workspace_bytes, nullptr, 0,
                                 stream);
}

extern "C" size_t vilco_attn_dsplanes_bytes(int32_t B, int32_t H, int32_t T) {
  const long r32 = ((long)T + 31) / 32 * 32, c32 = (2L * T + 31) / 32 * 32;
  return (size_t)(VILCO_PACK_HDR + (long)B * H * r32 * c32 * 2 * 2);
}

extern "C" int vilco_attn_bwd_dsplanes(const float* q, const float* k, const float* v, const float* bias,
                              const int32_t* kv_len, const float* o, const float* lse, const float* dout,
                              float* dq, float* dk, float* dv, float* dbias, int32_t B, int32_t H, int32_t Tq,
                              int32_t Tk, int32_t hd, float scale, int32_t mode, int32_t window, int32_t precision, float drop_p,
                              uint32_t drop_seed, const vilco_attn_amax_in* amax_in, float* dq_amax, float* dk_amax,
                              float* dv_amax, float* dbias_amax, void* workspace, size_t workspace_bytes, void* ds_planes,
                              size_t ds_planes_bytes, void* stream) {
  int rc = check_common(B, H, Tq, Tk, hd, mode, precision, window);
  if (!(drop_p >= 0.f) || drop_p >= 1.f) return VILCO_ERR_BADARG;
  if (rc != VILCO_OK) return rc;
  if (!q || !k || !v || !o || !lse || !dout || !dq || !dk || !dv) return VILCO_ERR_BADARG;
  if (mode != 2 && !kv_len) return VILCO_ERR_BADARG;
  if (B == 0 || Tq == 0) return VILCO_OK;
  if (Tk == 0) return VILCO_ERR_BADARG;
  if (!workspace || workspace_bytes < vilco_attn_bwd_workspace(B, H, Tq, Tk, hd, precision)) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int HDP = hdp_of(hd), NP = np_of(precision);
  unsigned char* wsb = reinterpret_cast<unsigned char*>(up((long)reinterpret_cast<uintptr_t>(workspace), 256));
  float* delta = reinterpret_cast<float*>(wsb);
  wsb += up((long)B * H * Tq * 4, 256);
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.bias = bias; a.lse = const_cast<float*>(lse); a.kv_len = kv_len;
  a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.hd = hd; a.C = H * hd; a.scale = scale; a.mode = mode; a.window = window;
  a.drop_thresh = vilco_drop_threshold_host(drop_p); a.drop_seed = drop_seed; a.drop_inv_keep = 1.f / (1.f - drop_p); a.seed_word = vilco_seed_word_dev();
  a.dout = dout; a.delta = delta; a.o_in = o; a.dq = dq; a.dk = dk; a.dv = dv; a.dbias = dbias;
  if ((dq_amax || dk_amax || dv_amax) && !fast64(a, precision)) return VILCO_ERR_UNSUPPORTED;      // see vilco_attn_amax_parts
  a.am_dq = dq_amax; a.am_dk = dk_amax; a.am_dv = dv_amax;
  if (dbias_amax && !(fast64_xl_fwd(a, precision) && dbias)) return VILCO_ERR_UNSUPPORTED;       // XLNet fast path only (has_bias = 2)
  a.am_ds = dbias_amax;
  if (ds_planes) {                 // XLNet fast path only; `dbias` / `dbias_amax` are then not written
    if (dbias || dbias_amax || !fast64_xl_fwd(a, precision) || (Tk % 64) != 0 || !vilco_aligned(ds_planes, 256)) return VILCO_ERR_UNSUPPORTED;
    if (ds_planes_bytes < vilco_attn_dsplanes_bytes(B, H, Tq)) return VILCO_ERR_WORKSPACE;
    const long r32 = ((long)Tq + 31) / 32 * 32, c32 = ((long)Tq + Tk + 31) / 32 * 32;
    unsigned char* u = reinterpret_cast<unsigned char*>(ds_planes);
    a.dsp = reinterpret_cast<_Float16*>(u + VILCO_PACK_HDR);
    a.ds_plane_stride = (long)B * H * r32 * c32; a.ds_batch_stride = r32 * c32; a.ds_ld = (int)c32;
    a.ds_inv_scale = reinterpret_cast<float*>(u) + VILCO_AMAX_MAX_BLOCKS;
  }
  ScaleWs sw;
  if (precision == 3) {
    const float* const xs[4] = {q, k, v, dout};
    const int Ts[4] = {Tq, Tk, Tk, Tq};
    sw = plan_amax(wsb, xs, Ts, 4, B, H * hd);
    use_external_amax(sw, amax_in, 4);
    a.sc = reinterpret_cast<const AttnScales*>(sw.out);
  }
  float* so = sw.out;
  __bf16* w = reinterpret_cast<__bf16*>(wsb + ATT_SCALE_BYTES);
  PackQueue pq;
  // the hd = 64 fast kernels read every operand from its natural planes only (transposing LDS reads)
  const bool nat_only = fast64(a, precision) || (fast64_xl_fwd(a, precision) && (a.dbias != nullptr || a.dsp != nullptr));
  a.qn = pack_operand(q, w, spec_nat(B, H, Tq, HDP), false, B, H, Tq, hd, HDP, NP, s, sw.parts[0], sw.n[0], so, &pq);
  a.don = pack_operand(dout, w, spec_nat(B, H, Tq, HDP), false, B, H, Tq, hd, HDP, NP, s, sw.parts[3], sw.n[3], so + 6, &pq);
  if (!nat_only) {
    a.qt = pack_operand(q, w, spec_tr(B, H, Tq, hd), true, B, H, Tq, hd, HDP, NP, s, sw.parts[0], sw.n[0], so, &pq);
    a.dot = pack_operand(dout, w, spec_tr(B, H, Tq, hd), true, B, H, Tq, hd, HDP, NP, s, sw.parts[3], sw.n[3], so + 6, &pq);
  }
  a.kn = pack_operand(k, w, spec_nat(B, H, Tk, HDP), false, B, H, Tk, hd, HDP, NP, s, sw.parts[1], sw.n[1], so + 2, &pq);
  a.vn = pack_operand(v, w, spec_nat(B, H, Tk, HDP), false, B, H, Tk, hd, HDP, NP, s, sw.parts[2], sw.n[2], so + 4, &pq);
  if (!nat_only) a.kt = pack_operand(k, w, spec_tr(B, H, Tk, hd), true, B, H, Tk, hd, HDP, NP, s, sw.parts[1], sw.n[1], so + 2, &pq);
  flush_packs(pq, sw, precision == 3, NP, B * H, s);
  return hd <= 32 ? dispatch<32>(a, precision, true, s) : (hd <= 64 ? dispatch<64>(a, precision, true, s) : (hd <= 128 ? dispatch<128>(a, precision, true, s) : dispatch<160>(a, precision, true, s)));
}
