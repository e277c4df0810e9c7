// MFMA GEMM for gfx950:  C[m][n] = epilogue(alpha * sum_k A(m,k) * B(k,n)),  fp32 in / fp32 out.
//
// Two stages per call, both on the caller's stream:
//   1. pack (pack.h): each fp32 operand is split ONCE into NP 16-bit "planes" laid out [part][row][cols32] in the
//      tensor's natural row-major layout -- or arrives already packed (vilco_pack, a_planes / b_planes) and is shared
//      by every product the tensor appears in.  k=3 convs keep their overlapped-row trick: the plane gets one zero row
//      before and after every sequence, and row (b,t) of the A operand is the contiguous span of rows t-1,t,t+1
//      (K = 3*Cin, row stride Cin): no im2col.
//   2. gemm_pp_kernel<BM,NP,F16,AKM,BKM>: tile BM x 128 x 32 (BM = 256 or 128), 8 waves in two groups that run one
//      phase apart (ping-pong), each wave a 64x64 (32x64) sub-tile of v_mfma_f32_16x16x32_{f16,bf16}, fp32 accumulate.
//      Operands are consumed k-contiguous (ds_read_b128 from XOR-swizzled 64-byte rows) or k-major
//      (ds_read_b64_tr_b16 from [k][R+16] rows).  NP=1: 1, NP=2: 3, NP=3: 6 MFMAs per fragment pair (dropped cross
//      terms <= 2^-9, 2^-17 / 2^-22 (fp16), 2^-26).  Optional split-K (grid.y) writes fp32 partials that a reduce
//      kernel sums and finishes.  Epilogue through LDS: 16-byte row-contiguous stores.
//
// precision 0 = NP 2 bf16 ("split"), 1 = NP 1 (plain bf16), 2 = NP 3 bf16 ("split3": numerically an fp32 GEMM),
// 3 = two fp16 parts of the per-tensor scaled operands ("f16x2": 22 bits, 3 MFMAs; pack.h; the default).
// Reference arithmetic replaced: see include/vilco_hip.h (vilco_gemm).
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>
#include <type_traits>
#include "common.h"
#define VILCO_TU "gemm"
#include "pack.h"

namespace {

constexpr int BN = 128, BK = 32;
constexpr int EPI_LD = 68;            // floats per staged epilogue row (64 + 4: conflict-free b32 writes / b128 reads)

// rows are 64 B (4 chunks of 16 B); chunk' = chunk ^ f(row>>2), f = {0,3,2,1}: every ds_read_b128 lane group
// ({0-3,12-15,20-27}, ...) then touches 16 distinct 16-B slots of the 256-B bank row (DESIGN_LOG.md 3.1)
__device__ __forceinline__ int swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 32 + ((chunk ^ swz(row)) << 3); }

// ------------------------------------------------------------------------------------------ GEMM on planes
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
  const float* row_mask;
};

struct PlaneOp {
  const __bf16* p;      // part 0, batch 0
  long plane_stride;    // between parts
  long batch_stride;    // between packed batches
  int nbi;              // packed inner batch count
  int has_o, has_i;     // whether the planes vary with the outer / inner batch index
  int rows;             // valid rows (M or N); tile rows beyond are clamped (their outputs are never stored)
  int seqT;             // rows per sequence for the overlapped-row layout (INT_MAX otherwise)
  long seq_stride;      // elements between sequences
  long row_stride;      // elements between rows
  int cols;             // k-major planes: padded row length (multiple of 32); tile columns are clamped to cols - 8
  long bytes;           // valid bytes behind p (one part, one batch): the buffer range of gemm_gl_kernel's loads
};

struct GArgs {
  PlaneOp a, b;
  float* c;             // output, or the split-K partial slabs
  float* cfinal;        // the real output (beta / reduce)
  long ldc;
  int M, N, Kp;
  int batch_inner;
  long sCo, sCi;
  int tiles_n, ntiles;
  int tm0;                // first row tile of this launch (a product launched as two row ranges of different tile heights)
  int ksplit, kchunk;   // kchunk = K-steps per split
  long split_stride;    // elements between split slabs
  unsigned* tile_ctr;   // split-K fix-up: arrival counter per (batch, tile); null = partial slabs + splitk_reduce_kernel
  const float* inv_a;       // fp16 x2 format: {1/s, s} of each operand, left by the pack kernels
  const float* inv_b;
  int vec_out;              // N, ldc, batch strides multiples of 4 and every epilogue pointer 16-byte aligned
  int band, bandT;          // XLNet relative-position band (vilco_gemm_desc.band)
  uint32_t drop_thresh, drop_seed;   // fused output dropout (vilco_gemm_desc.drop_p): keep iff hash(seed, m*N+n) >= thresh
  float drop_inv_keep;
  const uint32_t* seed_word;   // device step word mixed into drop_seed (common.h: vilco_step_seed)
  float* amax_out;          // optional: max|stored value| per workgroup (vilco_gemm_desc.amax_out)
  Epi e;
};

__device__ __forceinline__ long row_off(const PlaneOp& o, int r) {
  if (r >= o.rows) r = o.rows - 1;
  if (o.seqT == 0x7fffffff) return (long)r * o.row_stride;          // (uniform: plain operands skip the division)
  return (long)(r / o.seqT) * o.seq_stride + (long)(r % o.seqT) * o.row_stride;
}

__device__ __forceinline__ float store_out(const GArgs& g, long idx, int n, float acc, bool valid) {
  const Epi& e = g.e;
  float v = e.alpha * acc;
  if (e.bias) v += e.bias[n];
  if (e.preact) e.preact[idx] = v;
  if (e.act == VILCO_ACT_RELU) v = fmaxf(v, 0.f);
  else if (e.act == VILCO_ACT_GELU) v = gelu_f(v);
  if (!valid) v = 0.f;
  if (e.colscale) v *= e.colscale[n];
  if (g.drop_thresh) v = vilco_drop_hash(vilco_step_seed(g.drop_seed, g.seed_word), (uint64_t)idx) >= g.drop_thresh ? v * g.drop_inv_keep : 0.f;
  if (e.residual && (valid || !e.res_masked)) v += e.residual[idx];
  if (e.beta != 0.f) v += e.beta * g.cfinal[idx];
  g.cfinal[idx] = v;
  return fabsf(v);
}

// 4 consecutive columns of one row (vec_out only)
__device__ __forceinline__ float store_out4(const GArgs& g, long idx, int n, const f32x4& acc, bool valid) {
  const Epi& e = g.e;
  f32x4 v = acc * e.alpha;
  if (e.bias) v += *reinterpret_cast<const f32x4*>(e.bias + n);
#ifndef VILCO_GEMM_NO_NT_C
  if (e.preact) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(e.preact + idx));      // (read again only in backward)
#else
  if (e.preact) *reinterpret_cast<f32x4*>(e.preact + idx) = v;
#endif
  if (e.act == VILCO_ACT_RELU) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
  } else if (e.act == VILCO_ACT_GELU) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = gelu_f(v[k]);
  }
  if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (e.colscale) v *= *reinterpret_cast<const f32x4*>(e.colscale + n);
  if (g.drop_thresh) {
    const uint32_t seed = vilco_step_seed(g.drop_seed, g.seed_word);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      v[k] = vilco_drop_hash(seed, (uint64_t)(idx + k)) >= g.drop_thresh ? v[k] * g.drop_inv_keep : 0.f;
  }
  if (e.residual && (valid || !e.res_masked)) v += *reinterpret_cast<const f32x4*>(e.residual + idx);
  if (e.beta != 0.f) v += *reinterpret_cast<const f32x4*>(g.cfinal + idx) * e.beta;
  // Round 6: C leaves as STREAMING (nontemporal) stores.  The output of a launch is 9-75 MB written once by the whole grid and read
  // next by another kernel through the memory-side cache; keeping it out of the L2 write path measured -0.1 ... -0.15 ms on the
  // replayed P step (five same-box alternations, profiles/r06_ab_nt_c.txt: 20.51 -> 20.37 ms; GEMM family 12.07 -> 11.85 ms).
  // -DVILCO_GEMM_NO_NT_C: ordinary stores.
#ifndef VILCO_GEMM_NO_NT_C
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(g.cfinal + idx));
#else
  *reinterpret_cast<f32x4*>(g.cfinal + idx) = v;
#endif
  return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

template <bool F16>
__device__ __forceinline__ f32x4 mma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------ ping-pong kernel
// Schedule: the 8 waves form two groups of four (one wave of each group per SIMD: wave w and w+4 share a SIMD) that
// run ONE PHASE APART.  A K-step is two phases,
//   MEM(t):  ds_write the staged registers (tile t+1), re-issue the global loads for tile t+2 into the same registers,
//            ds_read this wave's fragments of tile t;           MFMA(t):  the wave's MFMAs on those fragments.
// Group 0 runs [MEM(t) MFMA(t)] between workgroup barriers, group 1 [MFMA(t-1) MEM(t)]: while one wave of a SIMD
// issues its MFMAs back to back the other one does all its LDS / global traffic, so the matrix pipe does not idle
// through the write -> barrier -> read chain of a one-phase loop (measured there with in-kernel stamps: ~3000 of the
// ~4800-6000 cycles per K-step; here ~2500 cycles per K-step in total).  ONE barrier per K-step, placed exactly where
// the hazards are: tile t+1 is written during K-step t (by both groups) into the buffer whose last readers (tile t-1)
// finished before the previous barrier, and is first read after the next one.
#ifdef VILCO_LAB   // tools/lab only: in-kernel cycle stamps of block 0, waves 0 and 4 (never compiled into the product)
__device__ unsigned long long vilco_lab_stamps[2 * 64 * 8];
#define STAMP(i)                                                                                              \
  do {                                                                                                         \
    if (stamp_on && t < 64) {                                                                                  \
      const unsigned long long c_ = __builtin_amdgcn_s_memtime();                                              \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
      if (lane == 0) vilco_lab_stamps[((wave >> 2) * 64 + t) * 8 + (i)] = c_;                                  \
    }                                                                                                          \
  } while (0)
#define STAMPX(i)                                                                                             \
  do {                                                                                                         \
    if (stamp_on) {                                                                                            \
      const unsigned long long c_ = __builtin_amdgcn_s_memtime();                                              \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
      if (lane == 0) vilco_lab_stamps[((wave >> 2) * 64 + 63) * 8 + (i)] = c_;                                 \
    }                                                                                                          \
  } while (0)
#else
#define STAMP(i) do {} while (0)
#define STAMPX(i) do {} while (0)
#endif

// k-major ("km") operands.  A tensor stored [a][b] (b contiguous) is packed ONCE as planes [a][b]; an operand whose
// contraction index is a (the activations and output gradients in dW = dY^T X, the weights in dX = dY W) reads those
// same planes k-major: the tile is 32 k-rows x R contiguous row indices (coalesced 16-byte chunks), kept in LDS as
// [k][R + 16] with column bit 6 flipped on odd 8-row groups, and the MFMA fragment (8 consecutive k of one row per
// lane) comes out of two ds_read_b64_tr_b16 -- gfx950's transposing LDS read (lane 4q+p of a 16-lane group supplies
// row q / columns 4p..4p+3 of a 4 x 16 block, lane i receives column i).  Row stride = 8 dwords mod 64 banks plus the
// bit-6 flip puts the 8 row segments of a 32-lane half on 8 disjoint 8-bank ranges: conflict free.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* p, int rs4) {     // rows k..k+3 at p, rows k+4..k+7 at p + rs4
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + rs4));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// Split-K fix-up inside the launch.  Every split workgroup stores its accumulators -- in their REGISTER layout, one
// 16-byte piece per lane and 16 x 16 block, 1 KB per wave instruction -- to its slot of the workspace with `sc1`
// (write-through) stores, waits for them, and one lane bumps the tile's arrival counter (agent-scope atomic).  The
// workgroup whose add came last (it learns so from the value the add returned) loads the other splits' pieces with
// `sc1` loads, sums them IN SPLIT ORDER (its own piece from registers: the result does not depend on who was last, and
// equals what splitk_reduce_kernel computes, bit for bit), resets the counter and runs the ordinary epilogue.  This is
// the hand-off row "one lane of each storing workgroup adds to ONE counter / the workgroup whose add came last" of
// MI355X_MICROARCH.md (stores and loads all sc1, 16 bytes).  No second launch, no fp32 slab in the output's layout.
__device__ __forceinline__ void st_sc1(f32x4* p, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void ld_sc1(f32x4& v, const f32x4* p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}
// the loads above are invisible to the compiler's vmcnt bookkeeping: wait here, and tie the four registers to the wait
__device__ __forceinline__ void wait_sc1(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}


// ---- epilogue shared by the MFMA kernels: accumulators of the 4 x 2 wave layout (wave tile 16 MI x 64) -> C through LDS.
// Each wave transposes its 16 x 64 accumulator blocks so that a lane owns 4 consecutive columns of one row and the
// stores / residual loads are 16 bytes per lane, 256 contiguous bytes per row, instead of 4-byte scatters (measured
// before: 24k of a 112k-cycle block at K = 1024).  The pipeline buffers must be dead when this is called.
template <int MI, bool F16, bool FIXUP>
__device__ __forceinline__ void gemm_epilogue(const GArgs& g, f32x4 (&acc)[MI][4], unsigned char* smem_raw, int tid, int lane,
                                              int wave, int wm, int wn, int m0, int n0, int bid, int ks, int zo, int zi,
                                              int zidx) {          // zidx: batch index of this workgroup (0 in a grouped launch)
  if (F16) {
    const float inv = g.inv_a[0] * g.inv_b[0];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] *= inv;
  }

  // ---- epilogue through LDS (the pipeline buffers are dead: every fragment read retired before the last barrier):
  // each wave transposes its 16 x 64 accumulator blocks so that a lane owns 4 consecutive columns of one row and
  // the stores / residual loads are 16 bytes per lane, 256 contiguous bytes per row, instead of 4-byte scatters
  // (measured before: 24k of a 112k-cycle block at K = 1024).
  const long coff = zo * g.sCo + zi * g.sCi;
  if constexpr (FIXUP) if (g.ksplit > 1 && g.tile_ctr) {
    __shared__ unsigned arrived;
    constexpr long PIECES = MI * 4 * 64;                        // f32x4 pieces per wave
    const long tile = (long)zidx * g.ntiles + bid;
    f32x4* mine = reinterpret_cast<f32x4*>(g.c) + ((tile * g.ksplit + ks) * 8 + wave) * PIECES + lane;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) st_sc1(mine + (i * 4 + j) * 64, acc[i][j]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave, before the barrier the signal is behind
    __syncthreads();
    if (tid == 0) arrived = __hip_atomic_fetch_add(g.tile_ctr + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (arrived != (unsigned)(g.ksplit - 1)) return;
    if (tid == 0) __hip_atomic_store(g.tile_ctr + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
    const f32x4* all = reinterpret_cast<const f32x4*>(g.c) + (tile * g.ksplit * 8 + wave) * PIECES + lane;
    const long sstride = 8 * PIECES;
    // every piece of one other split is requested before the first is used (one memory round trip per split, not per
    // piece); two splits are in flight at a time.  The running sums start from split 0 and take the splits in order.
    constexpr int NPC = MI * 4;
    f32x4 sum[NPC];
#pragma unroll
    for (int q = 0; q < NPC; ++q) sum[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < g.ksplit; s0 += 2) {
      f32x4 va[NPC], vb[NPC];
      const bool ha = s0 != ks, hb = s0 + 1 < g.ksplit && s0 + 1 != ks;
#pragma unroll
      for (int q = 0; q < NPC; ++q) {
        va[q] = sum[q]; vb[q] = sum[q];
        if (ha) ld_sc1(va[q], all + (long)s0 * sstride + q * 64);
        if (hb) ld_sc1(vb[q], all + (long)(s0 + 1) * sstride + q * 64);
      }
#pragma unroll
      for (int q = 0; q < NPC; q += 2) wait_sc1(va[q], va[q + 1], vb[q], vb[q + 1]);
#pragma unroll
      for (int q = 0; q < NPC; ++q) {
        sum[q] += ha ? va[q] : acc[q / 4][q % 4];
        if (s0 + 1 < g.ksplit) sum[q] += hb ? vb[q] : acc[q / 4][q % 4];
      }
    }
#pragma unroll
    for (int q = 0; q < NPC; ++q) acc[q / 4][q % 4] = sum[q];
  }
  const bool partial = g.ksplit > 1 && !g.tile_ctr;
  float* cp = g.c + (partial ? (long)ks * g.split_stride : 0);
  // (round 5) ALL of the wave's blocks are staged first, then ONE rolled loop finishes them: the unrolled form was ~10 k
  // instructions per kernel, more than the instruction cache keeps -- every launch re-fetched its set-up and epilogue code
  // line by line (in-kernel stamps: 4.7 k cycles from entry to the first tile load, 3 k for an uncontended epilogue).
  float* stg = reinterpret_cast<float*>(smem_raw) + wave * (16 * MI * EPI_LD);
  const int n = n0 + wn * 64 + (lane & 15) * 4;
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) stg[(i * 16 + (lane >> 4) * 4 + rr) * EPI_LD + j * 16 + (lane & 15)] = acc[i][j][rr];
  // (Round 6, in-kernel stamps of one workgroup, tools/lab/r6_epi.sh: this loop takes 11 k cycles plain, 13 k with a bias, 24 k with
  // GELU + pre-activation and 28 k with bias + residual, against 65 k for the K loop of a 4608 x 1024 x 1024 tile.  It is the
  // chip's write path, not a latency chain: the 192 workgroups of a single-round launch finish together and put 18.9 MB of C --
  // plus 18.9 MB of residual reads / pre-activation writes -- through the fabric at once, 4-5 TB/s.  Requesting a row's residual
  // one iteration ahead and loading bias / column scale once: 28 k -> 24 k with the residual, 11 k -> 14 k without, step +0.3 ms.)
#pragma unroll 1
  for (int it = 0; it < MI * 4; ++it) {
    const int row = it * 4 + (lane >> 4);                       // row inside the wave tile (block it >> 2)
    const f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * EPI_LD + (lane & 15) * 4);
    const int m = m0 + wm * (16 * MI) + row;
    if (m >= g.M || n >= g.N) continue;
    const long idx = coff + (long)m * g.ldc + n;
    if (partial) {
#ifndef VILCO_GEMM_NO_NT_C      // (split-K slabs stream out too: the deferred sums read them at the end of backward; R6.11)
      if (g.vec_out) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(cp + idx));
#else
      if (g.vec_out) *reinterpret_cast<f32x4*>(cp + idx) = v;
#endif
      else
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n + e < g.N) cp[idx + e] = v[e];
    } else {
      bool valid = g.e.row_len ? (m % g.e.rowT) < g.e.row_len[m / g.e.rowT] : true;
      if (g.e.row_mask) valid = valid && g.e.row_mask[m] != 0.f;
      if (g.vec_out) am = fmaxf(am, store_out4(g, idx, n, v, valid));
      else
#pragma unroll 1
        for (int e = 0; e < 4; ++e) if (n + e < g.N) am = fmaxf(am, store_out(g, idx + e, n + e, v[e], valid));
    }
  }
  if (g.amax_out && !partial) {     // max|C| of this tile for the operand pack of the next product (one float per workgroup)
    am = wave_max(am);
    __syncthreads();                // every wave is done with its staging rows
    float* red = reinterpret_cast<float*>(smem_raw);
    if (lane == 0) red[wave] = am;
    __syncthreads();
    if (tid == 0) {
      float m = red[0];
#pragma unroll
      for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
      g.amax_out[(long)zidx * gridDim.x + blockIdx.x] = m;     // (split-K fix-up: one last arriver per tile)
    }
  }
}

// K2 (r03): the single-part products (precision 4: weight gradients with long contractions) keep the TWO-slot layout of
// the two-part kernels but fill the slots with part 0 of two CONSECUTIVE K-steps: a barrier interval then covers 64
// elements of K -- 2 MFMAs per fragment pair, the same fragment reads and staging traffic as a two-part step.  With one
// part and 32 elements per interval the loop was bound by its MEM phase (16 MFMAs per wave against ~1000 cycles of
// reads, stores and loads: ~2000 cycles per 32 k, MFMA pipe 25 % busy); the interval's fixed cost now buys twice the K.
template <int BM, int NP, bool F16, bool AKM, bool BKM, bool K2 = false>
#ifdef VILCO_GEMM_WPE            // (lab: 4 holds every instantiation at <= 128 VGPRs, tools/lab/occ_ab.py)
__global__ __launch_bounds__(512, VILCO_GEMM_WPE) void gemm_pp_kernel(GArgs g) {
#else
__global__ __launch_bounds__(512) void gemm_pp_kernel(GArgs g) {
#endif
  static_assert(!K2 || (NP == 2 && F16), "K2: two slots, fp16 parts");
  constexpr int NT = 512;
  constexpr int MI = BM / 64;                    // 16-row A fragments per wave (wave tile = 16*MI x 64)
  // k-major LDS row strides (elements): 16 mod 128, so that a row is 8 dwords mod 64 banks (the 192-row tile borrows
  // the 256-row stride: its columns 128..191 land on 192..255 under the bit-6 flip)
  constexpr int RSA = (BM == 192 ? 256 : BM) + 16, RSB = BN + 16;
  constexpr int A_EL = AKM ? 32 * RSA : BM * 32; // elements per part per stage
  constexpr int B_EL = BKM ? 32 * RSB : BN * 32;
  constexpr int TILE = A_EL + B_EL;
  constexpr int RA = (BM * 4 + NT - 1) / NT;     // 16-byte chunks per thread per part: 2 or 1 (192 rows: 1.5 -> the
  constexpr bool RA_TAIL = (BM * 4) % NT != 0;   //  second round is taken by the first half of the threads only)
  constexpr int RB = BN * 4 / NT;                // 1
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);   // [stage 2][part NP][A tile | B tile]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;       // 4 x 2 waves; group = wave >> 2
  const bool late = wave >= 4;
#ifdef VILCO_LAB
  const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (wave & 3) == 0;
#endif
  STAMPX(0);

  int bid = blockIdx.x;
  {
    const int nwg = g.ntiles, q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z, zo = z / g.batch_inner, zi = z % g.batch_inner;
  const int ks = blockIdx.y;

  const __bf16* pa = g.a.p + ((long)(g.a.has_o ? zo : 0) * g.a.nbi + (g.a.has_i ? zi : 0)) * g.a.batch_stride;
  const __bf16* pb = g.b.p + ((long)(g.b.has_o ? zo : 0) * g.b.nbi + (g.b.has_i ? zi : 0)) * g.b.batch_stride;

  long offA[RA], offB[RB];
  int ldsA[RA], ldsB[RB];
  const int crow = tid >> 2, cc = tid & 3;
  const bool tailA = !RA_TAIL || tid < (BM * 4) % NT;     // does this thread take part in the last A round?
#pragma unroll
  for (int r = 0; r < RA; ++r) {
    if (AKM) {
      constexpr int CPR = BM / 8;                // chunks per k-row
      int id = tid + r * NT;
      if (id >= 32 * CPR) id = 0;                // (unused slot of the tail round)
      const int k = id / CPR, c = id % CPR;
      int col = m0 + c * 8;
      if (col > g.a.cols - 8) col = g.a.cols - 8;
      offA[r] = (long)k * g.a.row_stride + col;
      ldsA[r] = k * RSA + ((c * 8) ^ (((k >> 3) & 1) << 6));
    } else {
      int row = crow + r * (NT / 4);
      if (row >= BM) row = 0;
      offA[r] = row_off(g.a, m0 + row) + cc * 8;
      ldsA[r] = lds_off(row, cc);
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    if (BKM) {
      constexpr int CPR = BN / 8;
      const int id = tid + r * NT, k = id / CPR, c = id % CPR;
      int col = n0 + c * 8;
      if (col > g.b.cols - 8) col = g.b.cols - 8;
      offB[r] = (long)k * g.b.row_stride + col;
      ldsB[r] = A_EL + k * RSB + ((c * 8) ^ (((k >> 3) & 1) << 6));
    } else {
      const int row = crow + r * (NT / 4);
      offB[r] = row_off(g.b, n0 + row) + cc * 8;
      ldsB[r] = A_EL + lds_off(row, cc);
    }
  }
  const long stepA = AKM ? (long)BK * g.a.row_stride : BK;   // elements per K-step
  const long stepB = BKM ? (long)BK * g.b.row_stride : BK;
  // The staging loads are BUFFER loads: a 128-bit resource (base of this workgroup's operand matrix, one per part) in
  // SGPRs + this lane's constant 32-bit byte offset + the K-step's byte offset as the instruction's scalar offset -- one
  // instruction per chunk and no vector address arithmetic at all.  In the MEM phase every instruction counts: the SIMD's
  // other wave issues MFMAs back to back at raised priority, which leaves this wave about one issue slot per MFMA
  // (~16 cycles) -- in-kernel stamps (tools/lab/stamps.py, r03): the 6 global loads with their 14 64-bit address VALUs
  // and 6 hazard nops took ~580 of a ~1100-cycle MEM phase.  (Host side: vilco_gemm refuses an operand matrix of 2^31 bytes or more.)
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t rsA[NP], rsB[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    rsA[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(pa + (K2 ? 0 : q * g.a.plane_stride)), 0, -1, 0x00020000);
    rsB[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(pb + (K2 ? 0 : q * g.b.plane_stride)), 0, -1, 0x00020000);
  }
  unsigned voffA[RA], voffB[RB];
#pragma unroll
  for (int r = 0; r < RA; ++r) voffA[r] = (unsigned)(offA[r] * 2);
#pragma unroll
  for (int r = 0; r < RB; ++r) voffB[r] = (unsigned)(offB[r] * 2);
  // fragment addresses (elements inside one part's tile)
  int fbA[MI], fbB[4];
  {
    const int grp = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
    for (int i = 0; i < MI; ++i)
      fbA[i] = AKM ? (8 * grp + q4) * RSA + ((wm * (16 * MI) + i * 16 + 4 * p4) ^ ((grp & 1) << 6))
                   : lds_off(wm * (16 * MI) + (lane & 15), lane >> 4) + i * 16 * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      fbB[j] = A_EL + (BKM ? (8 * grp + q4) * RSB + ((wn * 64 + j * 16 + 4 * p4) ^ ((grp & 1) << 6))
                           : lds_off(wn * 64 + (lane & 15), lane >> 4) + j * 16 * 32);
  }

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kt0 = ks * g.kchunk;
  int kt1 = kt0 + g.kchunk;
  const int nk32 = g.Kp / BK;                          // K-steps of 32 in the planes
  const int nk_all = K2 ? (nk32 + 1) / 2 : nk32;       // barrier intervals
  if (kt1 > nk_all) kt1 = nk_all;
  int kb = kt0;
  if (g.band) {   // block-uniform: relative-position band of this tile (see vilco_gemm_desc.band)
    const int T = g.bandT;
    if (g.band == 1) {                     // C[i][p]: rows m0.., columns n0..
      if (n0 >= 2 * T - m0 || n0 + BN - 1 < T - (m0 + BM - 1)) return;
    } else {
      int lo, hi;                          // non-zero k range of the A rows m0 .. m0+BM-1
      if (g.band == 2) { lo = T - (m0 + BM - 1); hi = 2 * T - m0; }                  // rows i, k = p
      else { lo = T - (m0 + BM - 1); hi = 2 * T - m0; if (hi > T) hi = T; }         // rows p, k = i in [T-p, 2T-p) /\ [0,T)
      if (lo < 0) lo = 0;
      const int b0 = lo / BK, b1 = (hi + BK - 1) / BK;
      if (b0 > kb) kb = b0;
      if (b1 < kt1) kt1 = b1;
      if (kt1 < kb) kt1 = kb;
    }
  }
  const int kt0b = kb;
  const int nk = kt1 - kt0b;

#ifndef VILCO_GEMM_DEPTH2
#define VILCO_GEMM_DEPTH2 1
#endif
  // D2: staging loads issued TWO K-steps ahead into a second register set (tiles alternate between the sets); every load
  // is unconditional (the K offset is clamped to the last tile) so that the compiler's vmcnt model sees no join between a
  // load and its use and leaves exactly the younger tile's loads in flight at the wait.
  constexpr bool D2 = VILCO_GEMM_DEPTH2 && !K2 && BM <= 192 && NP == 2 && F16;
  bf16x8 stA2[D2 ? NP : 1][D2 ? RA : 1], stB2[D2 ? NP : 1][D2 ? RB : 1];
  auto gload2 = [&](int kt, auto& sa_, auto& sb_) {
    if (kt > kt1 - 1) kt = kt1 - 1;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int sa = (int)(2 * (long)kt * stepA), sb = (int)(2 * (long)kt * stepB);
#pragma unroll
      for (int r = 0; r < RA; ++r)
        sa_[q][r] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsA[q], (int)voffA[r], sa, 0));
#pragma unroll
      for (int r = 0; r < RB; ++r)
        sb_[q][r] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsB[q], (int)voffB[r], sb, 0));
    }
  };
  auto lstore2 = [&](int st, const auto& sa_, const auto& sb_) {
    __bf16* s = smem + st * NP * TILE;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
      for (int r = 0; r < RA; ++r)
        if (r + 1 < RA || tailA) *reinterpret_cast<bf16x8*>(s + q * TILE + ldsA[r]) = sa_[q][r];
#pragma unroll
      for (int r = 0; r < RB; ++r) *reinterpret_cast<bf16x8*>(s + q * TILE + ldsB[r]) = sb_[q][r];
    }
  };
  bf16x8 stA[NP][RA], stB[NP][RB];
  auto gload = [&](int kt) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      // slot q: part q of this K-step, or (K2) part 0 of K-step 2 kt + q -- zeros beyond an odd count of steps
      const int sa = (int)(2 * (long)(K2 ? 2 * kt + q : kt) * stepA), sb = (int)(2 * (long)(K2 ? 2 * kt + q : kt) * stepB);
      const bool live = !K2 || 2 * kt + q < nk32;
#pragma unroll
      for (int r = 0; r < RA; ++r)
        if (r + 1 < RA || tailA)
          stA[q][r] = live ? __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsA[q], (int)voffA[r], sa, 0)) : bf16x8{};
#pragma unroll
      for (int r = 0; r < RB; ++r)
        stB[q][r] = live ? __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsB[q], (int)voffB[r], sb, 0)) : bf16x8{};
    }
  };
  auto lstore = [&](int st) {
    __bf16* s = smem + st * NP * TILE;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
      for (int r = 0; r < RA; ++r)
        if (r + 1 < RA || tailA) *reinterpret_cast<bf16x8*>(s + q * TILE + ldsA[r]) = stA[q][r];
#pragma unroll
      for (int r = 0; r < RB; ++r) *reinterpret_cast<bf16x8*>(s + q * TILE + ldsB[r]) = stB[q][r];
    }
  };

  if constexpr (D2) {
    if (nk > 0) {                          // tile 0 -> LDS; tiles 1 (set "stA2") and 2 (set "stA") in flight
      gload2(kt0b, stA, stB);
      lstore2(0, stA, stB);
      gload2(kt0b + 1, stA2, stB2);
      __builtin_amdgcn_sched_barrier(0);   // the two tiles' loads stay in tile order (the loop's vmcnt waits count on it)
      gload2(kt0b + 2, stA, stB);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (nk > 0) {
    gload(kt0b);
    lstore(0);
    if (nk > 1) gload(kt0b + 1);
  }
  __syncthreads();
  STAMPX(1);

// Loop variants (lab builds: -DVILCO_GEMM_VARIANT=n, tools/lab/gemm_variants.sh).  3 (default since r03): a wave's
// fragment reads of tile t are issued FIRST in its MEM phase, so their latency runs under the staging stores and the global
// loads of the same phase, and the MFMA phase starts on the compiler's counted lgkmcnt waits instead of a full drain:
// +3...6 % on every shape of the step (same box: 4608x1024x1024 41.1 -> 38.8 us, 9216x1024x1024 81.3 -> 76.8 us).
// 0 = r02's order (stores, loads, reads, s_waitcnt lgkmcnt(0)); 1 = 0 without the drain; 2 = reads first with the drain.
#ifndef VILCO_GEMM_VARIANT
#define VILCO_GEMM_VARIANT 3
#endif
#ifndef VILCO_GEMM_PRIO      // 1: MFMA phase at priority 1 (default); 0: no s_setprio; 2: MEM phase above the MFMA phase
#define VILCO_GEMM_PRIO 1
#endif
  bf16x8 fa[NP][MI], fb[NP][4];
  auto frag_reads = [&](int t) {
    const __bf16* s = smem + (t & 1) * NP * TILE;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
        fa[q][i] = AKM ? tr_frag(s + q * TILE + fbA[i], 4 * RSA) : *reinterpret_cast<const bf16x8*>(s + q * TILE + fbA[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        fb[q][j] = BKM ? tr_frag(s + q * TILE + fbB[j], 4 * RSB) : *reinterpret_cast<const bf16x8*>(s + q * TILE + fbB[j]);
    }
  };
  auto mem_phase = [&](int t) {          // stage tile t+1, re-issue tile t+2, fetch this wave's fragments of tile t
#if VILCO_GEMM_VARIANT == 2 || VILCO_GEMM_VARIANT == 3
    frag_reads(t);                       // (lab) fragment reads first: their latency runs under the staging traffic
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (t + 1 < nk) lstore((t + 1) & 1);
#ifdef VILCO_LAB_FINE
    STAMP(1);
#endif
    if (t + 2 < nk) gload(kt0b + t + 2);
#ifdef VILCO_LAB_FINE
    STAMP(2);
#endif
#if !(VILCO_GEMM_VARIANT == 2 || VILCO_GEMM_VARIANT == 3)
    frag_reads(t);
#endif
  };
  auto mfma_phase = [&]() {
#if VILCO_GEMM_PRIO == 1
    __builtin_amdgcn_s_setprio(1);
#elif VILCO_GEMM_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mma<F16>(fa[0][i], fb[0][j], acc[i][j]);
    if constexpr (K2) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mma<F16>(fa[1][i], fb[1][j], acc[i][j]);
    } else if constexpr (NP >= 2) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = mma<F16>(fa[0][i], fb[1][j], acc[i][j]);
          acc[i][j] = mma<F16>(fa[1][i], fb[0][j], acc[i][j]);
        }
    }
    if constexpr (NP == 3) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = mma<F16>(fa[1][i], fb[1][j], acc[i][j]);
          acc[i][j] = mma<F16>(fa[0][i], fb[2][j], acc[i][j]);
          acc[i][j] = mma<F16>(fa[2][i], fb[0][j], acc[i][j]);
        }
    }
#if VILCO_GEMM_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#elif VILCO_GEMM_PRIO == 2
    __builtin_amdgcn_s_setprio(2);       // (lab) the MEM phase that follows outranks the partner's MFMA stream
#endif
  };

  // D2 MEM phase of K-step t (odd tiles live in stA2 / stB2, even ones in stA / stB): tile t+1 -> LDS from its set, then
  // that set takes tile t+3.  Past the end the last tile is loaded / stored again: its LDS stage is never read.
  auto mem_phase2 = [&](int t, auto odd_) {
    const bool odd = decltype(odd_)::value;
    frag_reads(t);
    __builtin_amdgcn_sched_barrier(0);
    if (odd) { lstore2((t + 1) & 1, stA, stB); gload2(kt0b + t + 3, stA, stB); }
    else { lstore2((t + 1) & 1, stA2, stB2); gload2(kt0b + t + 3, stA2, stB2); }
  };
  if constexpr (D2) {
    // pairs of K-steps with NO conditional inside the loop body (a skipped second half would reach the loop header with
    // the two register sets' loads in the opposite age order, and the compiler's vmcnt model would drain both); an odd
    // last K-step runs after the loop
    int t = 0;
    if (!late) {
      for (; t + 1 < nk; t += 2) {
        mem_phase2(t, std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        mem_phase2(t + 1, std::true_type{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
      }
      if (t < nk) {
        mem_phase2(t, std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_phase();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
      }
    } else {
      for (; t + 1 < nk; t += 2) {
        if (t > 0) mfma_phase();
        __builtin_amdgcn_sched_barrier(0);
        mem_phase2(t, std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        mfma_phase();
        __builtin_amdgcn_sched_barrier(0);
        mem_phase2(t + 1, std::true_type{});
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
      }
      if (t < nk) {
        if (t > 0) mfma_phase();
        __builtin_amdgcn_sched_barrier(0);
        mem_phase2(t, std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
      }
      if (nk > 0) mfma_phase();
    }
  } else
  if (!late) {
    for (int t = 0; t < nk; ++t) {
      STAMP(0);
      mem_phase(t);
#if VILCO_GEMM_VARIANT == 0 || VILCO_GEMM_VARIANT == 2
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      __builtin_amdgcn_sched_barrier(0);
      STAMP(3);
      mfma_phase();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(5);
      __syncthreads();
      STAMP(6);
    }
  } else {
    for (int t = 0; t < nk; ++t) {
      STAMP(0);
      if (t > 0) mfma_phase();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(3);
      mem_phase(t);
#if VILCO_GEMM_VARIANT == 0 || VILCO_GEMM_VARIANT == 2
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      __builtin_amdgcn_sched_barrier(0);
      STAMP(5);
      __syncthreads();
      STAMP(6);
    }
    if (nk > 0) mfma_phase();
  }
  STAMPX(2);

  gemm_epilogue<MI, F16, true>(g, acc, smem_raw, tid, lane, wave, wm, wn, m0, n0, bid, ks, zo, zi, (int)blockIdx.z);
  STAMPX(3);
}


// ------------------------------------------------------------------------------------------ LDS-DMA kernel (round 5)
// gemm_gl_kernel: the fp16 x2 products (precision 3: two parts, three MFMAs per fragment pair; SINGLE = precision 4: the
// leading parts only) with the staging rebuilt around what round 5's load-path probe measured (tools/lab/ldpath.hip, one
// workgroup per CU streaming an L2-resident operand): 1-KiB load instructions made of 64-byte row segments -- the BK = 32
// K-steps of gemm_pp_kernel -- move 29 B / clk / CU, whole 128-byte lines 50 (to registers) to 58 (LDS-DMA).  A 192 x 128
// tile needs 40 KB per 32 k: 1400 cycles of the ~2400 a K-step took were the address path alone.  Here:
//   * a K CHUNK is 64 elements: every load instruction covers 8 rows x 128 B = whole lines (k-contiguous operands: 8 tile
//     rows x 64 k; k-major operands: 8 k-rows x 64 columns);
//   * the loads are LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write -- the VGPR -> LDS store path
//     (13 cycles per ds_write_b128, two SIMD halves) was the other half of the old MEM phase.  The LDS image of a DMA is
//     lane-linear, so the bank swizzle sits on the SOURCE address (lane i fetches the chunk that belongs at position i) and
//     on the fragment reads;
//   * every operand tile is a set of [R][128 B] panels (k-contiguous: R = tile rows; k-major: one panel per 64 columns,
//     R = 64 k), 16-byte chunk c of row r at c ^ ((r >> 1) & 7) (k-contiguous, ds_read_b128) or with the 32-byte pair index
//     XORed by ((k >> 1) & 1) | (((k >> 3) & 1) << 1) (k-major, ds_read_b64_tr_b16): both conflict-free by the lane groups
//     of MI355X_MICROARCH.md (checked exhaustively, tools/lab/swizzle_check.py);
//   * two LDS stages of one chunk (2 slots x (A + B) x 64 k: 160 KB at 192 rows), ping-pong as before: waves 0-3 ("early",
//     the upper half of the tile's rows) and 4-7 ("late") run one PHASE apart, two raw barriers per chunk:
//         early:  [ reads(t), DMA(t+1) | X | MFMA(t), vmcnt(0) | Y ]      late:  [ MFMA(t-1), vmcnt(0) | X | reads(t), DMA(t+1) | Y ]
//     A DMA needs a phase to land before its first reader, and the late group issues at the END of an interval -- so the work
//     is split by WHO READS FIRST: the early waves fetch what they read themselves at the start of the next interval (the A
//     rows of the upper half and all of B), the late waves the A rows only they read, one phase later.  Each group waits for
//     its own DMAs (vmcnt(0)) just before the barrier that precedes their first reader ("read a staged buffer one phase after
//     the wait that retires it", cdna_hip_programming.md 5).
// SINGLE keeps the two slots and fills them with part 0 of two consecutive 64-element chunks (128 k per interval).
// (Measured and removed, round 5: an L2 prefetch of every wave's own operand lines three chunks ahead -- one 4-byte load per
// 128-byte line, never waited for -- against the 25 % the step's cold launches lose to back-to-back ones: the replayed P step
// went 22.5 -> 23.4 ms, GEMM time 12.66 -> 13.7 ms.  The cold share is not load latency.  DESIGN_LOG.md R5.)
template <int BM, bool AKM, bool BKM, bool SINGLE>
__device__ __forceinline__ void gemm_gl_body(const GArgs& g, const int z) {
  constexpr int MI = BM / 64;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;       // one slot of one operand
  constexpr int SLOT = A_BYTES + B_BYTES, STAGE = 2 * SLOT;
  constexpr int KSLOT = 64;                                   // k per slot
  constexpr int KINT = SINGLE ? 128 : 64;                     // k per barrier interval
  // pieces (1 KiB = 8 panel rows): A has BM / 8, B has 16.  The early group takes the A pieces its own waves read.
  constexpr int NA0 = AKM ? 8 * ((BM / 2 + 63) / 64) : BM / 16;
  constexpr int NA1 = BM / 8 - NA0;
  constexpr int NP0 = (NA0 + 16) / 4, NP0A = NA0 / 4;         // pieces per early wave (per slot); the first NP0A are A's
  constexpr int NP1 = NA1 / 4;                                // pieces per late wave (per slot)
  static_assert(NA0 % 4 == 0 && NA1 % 4 == 0 && NP1 >= 1, "piece split");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const bool late = wave >= 4;
  const int w4 = wave & 3;
#ifdef VILCO_LAB
  const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (wave & 3) == 0;
#endif
  STAMPX(0);

  int bid = blockIdx.x;
  {
    const int nwg = g.ntiles, q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tm = bid / g.tiles_n + g.tm0, tn = bid % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int zo = z / g.batch_inner, zi = z % g.batch_inner;
  const int ks = blockIdx.y;

  const __bf16* pa = g.a.p + ((long)(g.a.has_o ? zo : 0) * g.a.nbi + (g.a.has_i ? zi : 0)) * g.a.batch_stride;
  const __bf16* pb = g.b.p + ((long)(g.b.has_o ? zo : 0) * g.b.nbi + (g.b.has_i ? zi : 0)) * g.b.batch_stride;
  __amdgpu_buffer_rsrc_t rsA[2], rsB[2];
  {
    const unsigned na = g.a.bytes > 0xffffffffL ? 0xffffffffu : (unsigned)g.a.bytes;
    const unsigned nb = g.b.bytes > 0xffffffffL ? 0xffffffffu : (unsigned)g.b.bytes;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      rsA[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(pa + (SINGLE ? 0 : q * g.a.plane_stride)), 0, na, 0x00020000);
      rsB[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(pb + (SINGLE ? 0 : q * g.b.plane_stride)), 0, nb, 0x00020000);
    }
  }
  STAMPX(5);
  // ---- this wave's DMA pieces: source byte offset of this lane, LDS offset of the piece inside a slot
  unsigned voff[NP0];
  int ldso[NP0];
  {
    const int sr = lane >> 3, cp = lane & 7;                  // panel row inside the piece, chunk POSITION in the 128-B row
#pragma unroll
    for (int j = 0; j < NP0; ++j) {
      const bool isB = !late && j >= NP0A;
      int p = late ? NA0 + w4 + 4 * j : (isB ? w4 + 4 * j - NA0 : w4 + 4 * j);
      if (late && j >= NP1) p = NA0 + w4;                     // (unused slot)
      const PlaneOp& o = isB ? g.b : g.a;
      const bool km = isB ? BKM : AKM;
      const int t0 = isB ? n0 : m0;
      long off;
      if (km) {
        const int panel = p >> 3, k = 8 * (p & 7) + sr;
        const int f = ((k >> 1) & 1) | (((k >> 3) & 1) << 1);
        const int c = (((cp >> 1) ^ f) << 1) | (cp & 1);      // source chunk of the row that belongs at position cp
        int col = t0 + panel * 64 + c * 8;
        if (col > o.cols - 8) col = o.cols - 8;
        off = (long)k * o.row_stride + col;
      } else {
        const int row = 8 * p + sr;
        const int c = cp ^ ((row >> 1) & 7);
        off = row_off(o, t0 + row) + c * 8;
      }
      voff[j] = (unsigned)(off * 2);
      ldso[j] = (isB ? A_BYTES : 0) + p * 1024;
    }
  }
  STAMPX(6);
  const int kstepA = AKM ? (int)(2 * KSLOT * g.a.row_stride) : 2 * KSLOT;     // bytes per 64 k
  const int kstepB = BKM ? (int)(2 * KSLOT * g.b.row_stride) : 2 * KSLOT;

  // ---- fragment addresses (bytes inside a slot; k32 half h adds hoff, MFMA block i / j its own term)
  int faddr[MI], fbddr[4];
  int fa_h1, fb_h1;                                           // what the second k32 half adds (k-major) or XORs (k-contiguous)
  {
    const int grp = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int fsw = ((q4 >> 1) & 1) | ((grp & 1) << 1);
    const int ksw = (lane >> 1) & 7;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (AKM) {
        const int c0 = wm * (16 * MI) + 16 * i;
        faddr[i] = (c0 >> 6) * 8192 + (8 * grp + q4) * 128 + ((((c0 >> 4) & 3) ^ fsw) << 5) + 8 * p4;
      } else {
        faddr[i] = (wm * (16 * MI) + 16 * i + (lane & 15)) * 128 + (((lane >> 4) ^ ksw) << 4);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (BKM) fbddr[j] = A_BYTES + wn * 8192 + (8 * grp + q4) * 128 + ((j ^ fsw) << 5) + 8 * p4;
      else fbddr[j] = A_BYTES + (wn * 64 + 16 * j + (lane & 15)) * 128 + (((lane >> 4) ^ ksw) << 4);
    }
    fa_h1 = AKM ? 4096 : 64;
    fb_h1 = BKM ? 4096 : 64;
  }

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  STAMPX(7);
  // ---- K range of this workgroup, in intervals of KINT
  const int nk32 = g.Kp / BK;
  constexpr int SUB = KINT / 32;                              // k32 sub-steps per interval
  const int nint_all = (nk32 + SUB - 1) / SUB;
  int c0 = ks * g.kchunk, c1 = c0 + g.kchunk;
  if (c1 > nint_all) c1 = nint_all;
  int sub_end = nk32;                                         // first k32 sub-step that is not computed
  if (g.band) {
    const int T = g.bandT;
    if (g.band == 1) {
      if (n0 >= 2 * T - m0 || n0 + BN - 1 < T - (m0 + BM - 1)) return;
    } else {
      int lo = T - (m0 + BM - 1), hi = 2 * T - m0;
      if (g.band == 3 && hi > T) hi = T;
      if (lo < 0) lo = 0;
      const int b0 = lo / KINT, b1 = (hi + KINT - 1) / KINT;
      if (b0 > c0) c0 = b0;
      if (b1 < c1) c1 = b1;
      if (c1 < c0) c1 = c0;
    }
  }
  const int nc = c1 - c0;
  const int last_sub = sub_end - (c1 - 1) * SUB;              // valid sub-steps of the last interval (>= SUB: all)

  auto issue = [&](int ci, int st) {                          // this wave's pieces of interval ci -> stage st
    unsigned char* sbase = smem_raw + st * STAGE;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kc64 = SINGLE ? 2 * ci + u : ci;              // which 64-k chunk goes into slot u
      const int sa = kc64 * kstepA, sb = kc64 * kstepB;
      if (!late) {
#pragma unroll
        for (int j = 0; j < NP0; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(j >= NP0A ? rsB[u] : rsA[u],
              (__attribute__((address_space(3))) void*)(sbase + u * SLOT + ldso[j]), 16, (int)voff[j], j >= NP0A ? sb : sa, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < NP1; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA[u], (__attribute__((address_space(3))) void*)(sbase + u * SLOT + ldso[j]),
                                                  16, (int)voff[j], sa, 0, 0);
      }
    }
  };

  bf16x8 fa[2][2][MI], fb[2][2][4];                           // [slot][k32 half][block]
  auto reads = [&](int st) {
    const unsigned char* sbase = smem_raw + st * STAGE;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if (AKM) fa[u][h][i] = tr_frag(reinterpret_cast<const __bf16*>(sbase + u * SLOT + faddr[i] + h * fa_h1), 4 * 64);
          else fa[u][h][i] = *reinterpret_cast<const bf16x8*>(sbase + u * SLOT + (faddr[i] ^ (h * fa_h1)));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (BKM) fb[u][h][j] = tr_frag(reinterpret_cast<const __bf16*>(sbase + u * SLOT + fbddr[j] + h * fb_h1), 4 * 64);
          else fb[u][h][j] = *reinterpret_cast<const bf16x8*>(sbase + u * SLOT + (fbddr[j] ^ (h * fb_h1)));
        }
      }
  };
  // nsub: valid k32 sub-steps of this interval (sub-step s = slot u, half h: SINGLE s = 2 u + h; two parts: s = h)
  auto mfmas = [&](int nsub) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < (SINGLE ? 2 : 1); ++u)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if ((SINGLE ? 2 * u + h : h) < nsub) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mma<true>(fa[u][h][i], fb[u][h][j], acc[i][j]);
          if constexpr (!SINGLE) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                acc[i][j] = mma<true>(fa[0][h][i], fb[1][h][j], acc[i][j]);
                acc[i][j] = mma<true>(fa[1][h][i], fb[0][h][j], acc[i][j]);
              }
          }
        }
      }
    __builtin_amdgcn_s_setprio(0);
  };
#define GL_BARRIER()                                  \
  do {                                                \
    __builtin_amdgcn_sched_barrier(0);                \
    __builtin_amdgcn_s_barrier();                     \
    __builtin_amdgcn_sched_barrier(0);                \
  } while (0)
#define GL_WAIT_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define GL_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

  if (nc > 0) {
    issue(c0, 0);
    STAMPX(4);
    GL_WAIT_DMA();
    GL_BARRIER();
    STAMPX(1);
    if (!late) {
      for (int t = 0; t < nc; ++t) {
        STAMP(0);
        reads(t & 1);
        __builtin_amdgcn_sched_barrier(0);
#ifdef VILCO_LAB_FINE
        STAMP(1);
#endif
        if (t + 1 < nc) issue(c0 + t + 1, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(2);
        GL_WAIT_LDS();
        GL_BARRIER();                                         // X
        STAMP(4);
        mfmas(t + 1 < nc ? SUB : last_sub);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5);
        GL_WAIT_DMA();
        STAMP(6);
        GL_BARRIER();                                         // Y
        STAMP(7);
      }
    } else {
      for (int t = 0; t < nc; ++t) {
        STAMP(0);
        if (t > 0) mfmas(SUB);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(1);
        GL_WAIT_DMA();
        STAMP(2);
        GL_BARRIER();                                         // X
        STAMP(3);
        reads(t & 1);
        __builtin_amdgcn_sched_barrier(0);
#ifdef VILCO_LAB_FINE
        STAMP(4);
#endif
        if (t + 1 < nc) issue(c0 + t + 1, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5);
        GL_WAIT_LDS();
        GL_BARRIER();                                         // Y
        STAMP(7);
      }
      mfmas(last_sub);
    }
  }
#undef GL_BARRIER
#undef GL_WAIT_DMA
#undef GL_WAIT_LDS
  STAMPX(2);
  gemm_epilogue<MI, true, false>(g, acc, smem_raw, tid, lane, wave, wm, wn, m0, n0, bid, ks, zo, zi, z);
  STAMPX(3);
}

template <int BM, bool AKM, bool BKM, bool SINGLE>
__global__ __launch_bounds__(512) void gemm_gl_kernel(GArgs g) {
  gemm_gl_body<BM, AKM, BKM, SINGLE>(g, (int)blockIdx.z);
}

// Grouped launch (round 5): up to four INDEPENDENT products of one shape -- the q / k / v projections of an attention block,
// forward and dX -- as one grid, blockIdx.z = the product.  Each of them alone is 192 tiles on 256 CUs (a 75 %-full round,
// plus a launch and an end-of-kernel write-back of its own); together the dispatcher packs 3 x 192 tiles into 2.25 rounds.
struct GArgsN { GArgs g[4]; };
template <int BM, bool AKM, bool BKM>
__global__ __launch_bounds__(512) void gemm_gl_group_kernel(GArgsN gg) {
  gemm_gl_body<BM, AKM, BKM, false>(gg.g[blockIdx.z], 0);
}


// out = epilogue(alpha * sum_s part[s]).  VEC (16-byte aligned rows, N % 4 == 0: the layout the MFMA kernel's own
// 16-byte partial stores require): a thread finishes four consecutive columns, and the partials of up to four splits
// are requested before the first one is added -- a scalar loop over the splits is one memory round trip per split.
// The sum keeps the split order either way (bitwise the same result).
// ------------------------------------------------------------------------------------------ few-row products
// gemm_skinny_kernel (round 6): y = x W^T with FEW token rows (M <= 640: the pyramid levels at T' <= 288, the 77-token text
// branch, every level of cfg1) -- shapes on which the tiled kernels above are all latency: 24-96 workgroups whose 4-chunk K
// loops expose one DMA round trip per chunk (in-kernel stamps, 288 x 1024 x 1024: 3.5 k cycles of set-up, 13.7 k for FOUR
// chunks, 3.9 k of epilogue = 11.4 us) and then need a second launch to sum the split-K slabs (4.1 us + the node gap): ~17 us
// for 0.6 GFLOP.  Here ONE launch, no LDS staging, no barrier in the K loop: a workgroup owns BMS x 64 outputs, its eight waves
// split K between them (contiguous eighths), every lane fetches its MFMA fragments straight from the operand planes (16 bytes =
// 8 consecutive k of one row: two consecutive k32 steps of a wave use the two halves of the same 128-byte lines), four steps
// per memory round trip, and the eight partial tiles are summed in wave order through LDS by the
// epilogue, which is the tiled kernels' own (store_out4: bias, activation, pre-activation, row masks, dropout, residual, beta,
// max|C|).  Both operands k-contiguous (NT), precision 3 (two fp16 parts, three MFMAs per fragment pair), unbatched.
constexpr int SK_LD = 68;                                       // floats per row of a wave's partial tile in LDS (EPI_LD's reasons)
template <int BMS>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(GArgs g) {
  constexpr int MB = BMS / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* part = reinterpret_cast<float*>(smem_raw);             // [8 waves][BMS][SK_LD]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.N + 63) >> 6;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * BMS, n0 = tn * 64;
  const int nk32 = g.Kp / BK;
  const int per = (nk32 + 7) >> 3;
  const int s0 = wave * per;
  int s1 = s0 + per;
  if (s1 > nk32) s1 = nk32;
  const int r16 = lane & 15, kq = (lane >> 4) * 8;
  const __bf16* ap[MB];
  const __bf16* bp[4];
#pragma unroll
  for (int i = 0; i < MB; ++i) ap[i] = g.a.p + row_off(g.a, m0 + 16 * i + r16) + kq;
#pragma unroll
  for (int j = 0; j < 4; ++j) bp[j] = g.b.p + row_off(g.b, n0 + 16 * j + r16) + kq;
  const long pa1 = g.a.plane_stride, pb1 = g.b.plane_stride;

  f32x4 acc[MB][4];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // up to FOUR k32 steps are requested at once (all of a wave's share at K <= 1024): one memory round trip per group of four, not
  // one per step -- with one step in flight a 4-step share took 4 exposed latencies (144 x 1024 x 1024: 11.9 us against 12.3 tiled)
  constexpr int NB = 4;
  bf16x8 fa[NB][2][MB], fb[NB][2][4];                           // [buffer][part][block]
  for (int st = s0; st < s1; st += NB) {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      if (st + u < s1) {
        const int k = (st + u) * BK;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          fa[u][0][i] = *reinterpret_cast<const bf16x8*>(ap[i] + k);
          fa[u][1][i] = *reinterpret_cast<const bf16x8*>(ap[i] + pa1 + k);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          fb[u][0][j] = *reinterpret_cast<const bf16x8*>(bp[j] + k);
          fb[u][1][j] = *reinterpret_cast<const bf16x8*>(bp[j] + pb1 + k);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      if (st + u < s1) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mma<true>(fa[u][0][i], fb[u][0][j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mma<true>(fa[u][0][i], fb[u][1][j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mma<true>(fa[u][1][i], fb[u][0][j], acc[i][j]);
      }
    }
  }

  // ---- the eight partial tiles -> LDS (row-major, one region per wave), summed in wave order, finished like any other tile
  float* mine = part + wave * (BMS * SK_LD);
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) mine[(16 * i + (lane >> 4) * 4 + rr) * SK_LD + 16 * j + r16] = acc[i][j][rr];
  __syncthreads();
  const float inv = g.inv_a[0] * g.inv_b[0];
  float am = 0.f;
#pragma unroll 1
  for (int q = tid; q < BMS * 16; q += 512) {
    const int row = q >> 4, c4 = (q & 15) * 4;
    const float* src = part + row * SK_LD + c4;
    f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
    for (int w = 1; w < 8; ++w) v += *reinterpret_cast<const f32x4*>(src + w * (BMS * SK_LD));
    v *= inv;
    const int m = m0 + row, n = n0 + c4;
    if (m >= g.M || n >= g.N) continue;
    const long idx = (long)m * g.ldc + n;
    bool valid = g.e.row_len ? (m % g.e.rowT) < g.e.row_len[m / g.e.rowT] : true;
    if (g.e.row_mask) valid = valid && g.e.row_mask[m] != 0.f;
    if (g.vec_out) am = fmaxf(am, store_out4(g, idx, n, v, valid));
    else
#pragma unroll 1
      for (int e = 0; e < 4; ++e) if (n + e < g.N) am = fmaxf(am, store_out(g, idx + e, n + e, v[e], valid));
  }
  if (g.amax_out) {
    am = wave_max(am);
    __syncthreads();
    if (lane == 0) part[wave] = am;
    __syncthreads();
    if (tid == 0) {
      float m = part[0];
#pragma unroll
      for (int w = 1; w < 8; ++w) m = fmaxf(m, part[w]);
      g.amax_out[blockIdx.x] = m;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GArgs g, int nz) {
  constexpr int V = VEC ? 4 : 1;
  const int NV = g.N / V;
  const long per = (long)g.M * NV;
  const long total = per * nz;
  float am = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int z = (int)(i / per);
    const long mn = i % per;
    const int m = (int)(mn / NV), n = (int)(mn % NV) * V;
    const int zo = z / g.batch_inner, zi = z % g.batch_inner;
    const long idx = zo * g.sCo + zi * g.sCi + (long)m * g.ldc + n;
    bool valid = true;
    if (g.e.row_len) valid = (m % g.e.rowT) < g.e.row_len[m / g.e.rowT];
    if (g.e.row_mask) valid = valid && g.e.row_mask[m] != 0.f;
    if (VEC) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      int k = 0;
      for (; k + 4 <= g.ksplit; k += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(g.c + (long)(k + u) * g.split_stride + idx);
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u];
      }
      if (k + 2 <= g.ksplit) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(g.c + (long)k * g.split_stride + idx);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(g.c + (long)(k + 1) * g.split_stride + idx);
        s += v0; s += v1;
        k += 2;
      }
      if (k < g.ksplit) s += *reinterpret_cast<const f32x4*>(g.c + (long)k * g.split_stride + idx);
      am = fmaxf(am, store_out4(g, idx, n, s, valid));
    } else {
      float s = 0.f;
      for (int k = 0; k < g.ksplit; ++k) s += g.c[(long)k * g.split_stride + idx];
      am = fmaxf(am, store_out(g, idx, n, s, valid));
    }
  }
  if (g.amax_out) {
    __shared__ float red[4];
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) g.amax_out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
}

// ------------------------------------------------------------------------------------------ host side
inline long align_up(long x, long a) { return (x + a - 1) / a * a; }
inline bool use_km() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("VILCO_GEMM_KM"); v = e ? atoi(e) : 1; }
  return v != 0;
}
constexpr long SCALE_BYTES = 2 * AMAX_MAX_BLOCKS * 4 + 256;   // fp16 x2 format: amax partials of A and B, then {1/sA, sA, 1/sB, sB}
constexpr long PACK_HDR = VILCO_PACK_HDR;                      // vilco_pack buffers: amax partials, {1/s, s}, then the planes

inline bool k2_enabled() {
  static const bool on = [] { const char* e = getenv("VILCO_GEMM_K2"); return !(e && e[0] == '0'); }();
  return on;
}
struct Tune { int bm, ks; };
inline Tune& tune() {       // env read once per process (the plan is made twice per launch); vilco_gemm_force changes it
  static Tune t = [] {
    const char* b = getenv("VILCO_GEMM_BM"); const char* k = getenv("VILCO_GEMM_KS");
    return Tune{b ? atoi(b) : 0, k ? atoi(k) : 0};
  }();
  return t;
}

struct Plan {
  int NP, Kp, BM, ksplit, kchunk;
  bool a_tr, b_tr;
  bool a_km, b_km;                // operand consumed k-major from natural-layout planes (ping-pong kernel only)
  int a_tap, b_tap;               // pack tap mode
  int a_out_rows, b_out_rows;     // plane rows written by the kc pack
  long a_plane, b_plane;          // elements per part (all batches)
  long a_batch, b_batch;          // elements per batch inside a part
  int a_nbo, a_nbi, b_nbo, b_nbi;
  long a_bytes, b_bytes, part_bytes;
  long split_stride;
  bool tapkm;                     // k=3 conv weight gradient as a k-major x k-major product over zero-padded token rows (make_plan)
  bool k2;                        // precision 4: a barrier interval of the kernel covers two K-steps (64 elements of K)
  bool fixup;                     // split-K finished inside the launch (tile counters) instead of by splitk_reduce_kernel
  bool gl;                        // gemm_gl_kernel (LDS-DMA staging, 64-element chunks): the fp16 x2 products
  bool skinny;                    // gemm_skinny_kernel (few rows, NT, precision 3): BM = its rows per workgroup (16 | 32)
};

// Arrival counters of the split-K fix-up: persistent, zero at load, reset by each tile's last arriver.  One region per
// stream (launches of one stream are ordered; different streams may run GEMMs at the same time).
constexpr int FIX_TILES = 4096, FIX_STREAMS = 32;
__device__ unsigned vilco_tile_counters[FIX_STREAMS * FIX_TILES];
// Default OFF.  Measured on MI355X (r03, same box, hipGraph replay of the whole step; tools/fixup_ab.sh): config P
// 27.49 ms with the fix-up against 27.59 ms with the reduce launch (noise), the T = 256 configuration 10.97 ms against
// 10.43 ms.  The chain sc1 stores -> vmcnt(0) -> atomic round trip -> sc1 loads costs the last arriver ~6 us, a dependent
// second launch ~5 us: device-scope hand-offs go through the memory side of the 8 XCDs (DESIGN_LOG.md 3.6 found the same for
// grid barriers).  Kept (tested bit-exact against the reduce kernel) behind VILCO_GEMM_FIXUP=1 / vilco_gemm_set_fixup.
inline bool& skinny_enabled() {
  static bool on = [] { const char* e = getenv("VILCO_GEMM_SKINNY"); return !(e && e[0] == '0'); }();
  return on;
}
inline int skinny_max_rows() {
  static const int v = [] { const char* e = getenv("VILCO_GEMM_SKINNY_M"); const int x = e ? atoi(e) : 640; return x > 0 ? x : 640; }();
  return v;
}
// largest M that takes 16-row workgroups (more workgroups, more re-reads of B).  Inside the planned range (K <= 768, N <= 1024) 16 rows
// are never slower (profiles/r06_skinny_bm16.txt: 154 x 1024 x 768 9.8 us against 11.4 at 32 rows, 512 x 512 x 512 7.6 against 8.6).
inline int skinny_bm16_rows() {
  static const int v = [] { const char* e = getenv("VILCO_GEMM_SKINNY_BM16"); const int x = e ? atoi(e) : 640; return x >= 0 ? x : 640; }();
  return v;
}
inline int skinny_max_k() {
  static const int v = [] { const char* e = getenv("VILCO_GEMM_SKINNY_K"); const int x = e ? atoi(e) : 768; return x > 0 ? x : 768; }();
  return v;
}
inline bool& fixup_enabled() {
  static bool on = [] { const char* e = getenv("VILCO_GEMM_FIXUP"); return e && e[0] == '1'; }();
  return on;
}
unsigned* tile_counters(hipStream_t s) {
  static std::mutex mu;
  static hipStream_t streams[FIX_STREAMS];
  static int n = 0;
  static unsigned* base = nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!base) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(vilco_tile_counters)) != hipSuccess || !q) return nullptr;
    base = reinterpret_cast<unsigned*>(q);
  }
  for (int i = 0; i < n; ++i)
    if (streams[i] == s) return base + (long)i * FIX_TILES;
  if (n >= FIX_STREAMS) return nullptr;
  streams[n] = s;
  return base + (long)(n++) * FIX_TILES;
}

inline bool& gl_enabled() {      // VILCO_GEMM_GL=0 / vilco_gemm_set_gl(0): the fp16 x2 products stay on gemm_pp_kernel (rounds 1-4)
  static bool v = [] { const char* e = getenv("VILCO_GEMM_GL"); return !(e && e[0] == '0'); }();
  return v;
}

inline bool& tail128_enabled() {      // VILCO_GEMM_TAIL128=1 / vilco_gemm_set_tail128(1): see gemm_impl (off by default: measured, no gain)
  static bool v = [] { const char* e = getenv("VILCO_GEMM_TAIL128"); return e && e[0] == '1'; }();
  return v;
}

inline bool tapkm_enabled() {
  static const bool v = [] { const char* e = getenv("VILCO_CONV_DW_KM"); return !(e && e[0] == '0'); }();
  return v;
}

void make_plan(const vilco_gemm_desc* d, Plan& p) {
  p.NP = d->precision == 1 ? 1 : ((d->precision == 0 || d->precision == 3 || d->precision == 4) ? 2 : 3);
  p.Kp = (int)align_up(d->K > 0 ? d->K : 1, 32);
  p.a_tr = !d->a_kcontig;
  p.b_tr = !d->b_kcontig;
  p.a_tap = p.b_tap = 0;
  p.tapkm = false;
  p.a_out_rows = d->M;
  p.b_out_rows = d->N;
  long a_batch = (long)d->M * p.Kp, b_batch = (long)d->N * p.Kp;
  // k-major consumption instead of a transposing pack: plain (no tap, unbatched) operands, 16-byte aligned rows
  const bool km_ok = use_km() && d->tap_operand == VILCO_TAP_NONE;
  p.a_km = km_ok && p.a_tr;
  p.b_km = km_ok && p.b_tr;
  if (p.a_km) { p.a_tr = false; p.a_out_rows = p.Kp; a_batch = (long)p.Kp * align_up(d->M, 32); }
  if (p.b_km) { p.b_tr = false; p.b_out_rows = p.Kp; b_batch = (long)p.Kp * align_up(d->N, 32); }
  if (d->tap_operand == VILCO_TAP_A) {
    if ((d->tapC % 8) == 0) {
      p.a_tap = 1;
      const long nseq = d->M / d->tapT;
      const long slack = (p.Kp + d->tapC - 1) / d->tapC;       // zero rows covering the last spans' over-read
      p.a_out_rows = (int)(nseq * (d->tapT + 2) + slack);
      a_batch = (long)p.a_out_rows * d->tapC;
    } else {
      p.a_tap = 2;
    }
  } else if (d->tap_operand == VILCO_TAP_B) {
    // dW[co][j*Cin + c] = sum_tok dZ[tok][co] X[tok + j - 1][c] (zero across sequence ends).  Round 4: no transposing pack.
    // Both operands are packed the way the forward conv packs its input -- token rows in their natural layout, one zero row
    // before and after every sequence (tap mode 1) -- and the product contracts over those PADDED rows r: with dZ' starting
    // at padded row 1, row r of dZ' meets rows r, r+1, r+2 of X' for the taps j = 0, 1, 2; X' has row stride Cin, so
    // "row r, columns j*Cin + c" of the virtual [rows][3 Cin] operand is the plain address r*Cin + j*Cin + c: the kernel's
    // ordinary k-major tile read (the overlapped-row trick of the forward conv, turned on its side).  The padding rows of
    // dZ' are zero, so whatever X' holds opposite them (the next sequence's first token) contributes nothing.
    if (use_km() && tapkm_enabled() && (d->tapC % 8) == 0 && (d->M % 8) == 0 && d->tapT > 0 && (d->K % d->tapT) == 0 &&
        d->batch_outer == 1 && d->batch_inner == 1) {
      const long nseq = d->K / d->tapT;
      p.tapkm = true;
      p.Kp = (int)align_up(nseq * (d->tapT + 2) - 1, 32);
      p.a_tap = p.b_tap = 1;
      p.a_tr = p.b_tr = false;
      p.a_km = p.b_km = true;
      p.a_out_rows = p.Kp + 1 + 32;          // + the skipped first row + one K-step of over-read
      p.b_out_rows = p.Kp + 3 + 32;
      a_batch = (long)p.a_out_rows * d->M;
      b_batch = (long)p.b_out_rows * d->tapC;
    } else {
      p.b_tap = 3;
    }
  }
  p.a_nbo = (d->sAo && d->batch_outer > 1) ? d->batch_outer : 1;
  p.a_nbi = (d->sAi && d->batch_inner > 1) ? d->batch_inner : 1;
  p.b_nbo = (d->sBo && d->batch_outer > 1) ? d->batch_outer : 1;
  p.b_nbi = (d->sBi && d->batch_inner > 1) ? d->batch_inner : 1;
  p.a_batch = align_up(a_batch, 8);
  p.b_batch = align_up(b_batch, 8);
  p.a_plane = p.a_batch * p.a_nbo * p.a_nbi;
  p.b_plane = p.b_batch * p.b_nbo * p.b_nbi;
  p.a_bytes = align_up(p.a_plane * p.NP * 2, 256);
  p.b_bytes = align_up(p.b_plane * p.NP * 2, 256);

  const long nbatch = (long)d->batch_outer * d->batch_inner;
  const long tn = (d->N + BN - 1) / BN;
  const long tiles128 = ((d->M + 127) / 128) * tn * nbatch;
  const long tiles256 = ((d->M + 255) / 256) * tn * nbatch;
  // tile / split-K choice, tuned on MI355X with tools/gemm_tune.py (profiles/r01_gemm_tune.txt).  The 128-row tile
  // (two blocks per CU) wins whenever all its tiles fit one per CU, or M is small; the 256-row tile for the many-tile
  // problems.  With >= 128 tiles only long K is worth splitting (>= 32 K-steps per split); with fewer, filling the
  // 256 CUs comes first, but never below 8 K-steps per split.
  // Tile height by wave quantisation: the grid runs in rounds of one tile per CU (256 CUs), so the cost of a tile
  // height BM is ceil(tiles / 256) * BM; the commonest shape, M = 4608 x N = 1024, is 144 tiles at 256 rows (44 % of
  // the CUs idle), 288 at 128 (two rounds) and 192 at 192 rows (one round, 75 % busy).  Ties go to the taller tile
  // (fewer B re-reads).  The 192-row tile exists for the default precision only.
  const long tiles192 = ((d->M + 191) / 192) * tn * nbatch;
  p.k2 = d->precision == 4 && d->band == 0 && k2_enabled();
  // round 5: the fp16 x2 products run on gemm_gl_kernel (128- or 192-row tiles: two stages of a 64-element chunk of a
  // 256-row tile do not fit the LDS)
  // (single-part products: measured in the step, the weight-gradient shapes run 5-15 % slower on it than on gemm_pp_kernel's K2
  // loop -- 1024 x 3072 x 9082: 103 -> 119 us -- their intervals hold a third of the MFMAs per byte and the late group's DMAs
  // do not land inside one MFMA phase; VILCO_GEMM_GL_SINGLE=1 selects it anyway)
  static const bool gl_single = [] { const char* e = getenv("VILCO_GEMM_GL_SINGLE"); return e && e[0] == '1'; }();
  p.gl = gl_enabled() && (d->precision == 3 || (d->precision == 4 && p.k2 && gl_single));
  if (tiles128 <= 256 || d->M < 2048) {
    p.BM = 128;
  } else {
    const long c128 = ((tiles128 + 255) / 256) * 128, c192 = ((tiles192 + 255) / 256) * 192, c256 = ((tiles256 + 255) / 256) * 256;
    p.BM = 256;
    long best = c256;
    if (p.gl) {
      // gemm_gl_kernel: a round of 128-row tiles takes 0.74 of a round of 192-row tiles, not 2/3 (measured, round 5:
      // 4608 x 4096 x 1024 as 5 rounds of 128 rows 131 us, as 3 rounds of 192 rows 107 us; 8192^3 3.33 vs 2.74 ms)
      p.BM = 192; best = c192;
      if (((tiles128 + 255) / 256) * 150 < best) { p.BM = 128; best = 0; }      // (142 measured; near-ties go to the taller tile: 4608 x 3072 x 1024 92.8 vs 96.6 us)
    }
    else if ((d->precision == 3 || d->precision == 4) && c192 < best) { p.BM = 192; best = c192; }
    if (!p.gl && c128 < best) { p.BM = 128; best = c128; }
  }
  // Round 6 (tools/lab/fwd_sweep.py, profiles/r06_fwd_sweep.txt): two sub-round cases the rules above sent to 128 rows.
  // (a) M < 2048 with more than one round of 128-row tiles but at most one of 192-row tiles (1152 x 4096 x 1024: 288 vs 192
  //     tiles): 47.4 -> 31.7 us (NT), 51.7 -> 33.0 (NN);
  // (b) 128..175 tiles of 128 rows and a long K (2304 x 1024 x 4096: 144 tiles x 3 splits = 432 workgroups, 1.7 rounds): 96 tiles
  //     of 192 rows x 2 splits fill ONE round with half the slab traffic: 68.1 -> 59.9 us (NT), 74.5 -> 62.7 (NN).
  static const bool sub192 = [] { const char* e = getenv("VILCO_GEMM_SUB192"); return !(e && e[0] == '0'); }();
  int sub192_ks = 0;
  if (sub192 && p.gl && d->precision == 3 && p.BM == 128 && nbatch == 1) {
    const int nk64 = (p.Kp / BK + 1) / 2;
    if (tiles128 > 256 && tiles192 <= 256 && tiles192 >= 176) p.BM = 192;
    else if (tiles128 >= 128 && tiles128 < 176 && nk64 >= 48 && tiles192 * 2 <= 256 && tiles192 * 2 >= 176) { p.BM = 192; sub192_ks = 2; }
  }
  const long tiles = p.BM == 256 ? tiles256 : (p.BM == 192 ? tiles192 : tiles128);
  const int nk32p = p.Kp / BK;
  // barrier intervals of the K loop: gemm_gl_kernel 64 k (two parts) / 128 k (single part), gemm_pp_kernel 32 k (K2: 64)
  const int nk = p.gl ? (d->precision == 4 ? (nk32p + 3) / 4 : (nk32p + 1) / 2) : (p.k2 ? (nk32p + 1) / 2 : nk32p);
  int ks = 1;
  // Split-K, re-measured in round 2 with tools/lab/ks_try*.sh (kernel + finish, operands packed): what matters is the
  // number of rounds -- a grid of exactly <= 256 workgroups beats a slightly larger one by 15-20 % (1024 x 1024 x 4608:
  // 4 splits = 256 workgroups 42.7 us, 5 splits = 320 workgroups 50.8 us), and a tile costs ~15-20 K-steps of prologue +
  // epilogue on top of its K loop, so a 75 %-full single round is better left alone (4608 x 1024 x 4096, 192 tiles:
  // unsplit 148 us, 3 splits 172 us) while a 56 %-full one still gains from splitting (2304 x 1024 x 4096: 97 -> 80 us).
  if (tiles < 256) {
    if (tiles >= 176) {
      ks = 1;
    } else if (tiles >= 128) {
      ks = nk / (p.gl ? 16 : 32);
      if (ks > 3) ks = 3;
    } else {
      ks = (int)(256 / tiles);
      if (ks > nk / (p.gl ? 4 : 8)) ks = nk / (p.gl ? 4 : 8);      // (the same 256 elements of K per split either way)
      if (ks > 16) ks = 16;
    }
    if (ks < 1) ks = 1;
    if (sub192_ks) ks = sub192_ks;
  }
  // Round 6: the single-part products on gemm_pp_kernel's K2 loop (the weight gradients: M = Cout = 1024 rows, long K) were always
  // planned at 128 rows (M < 2048).  A 128 x 128 tile of ONE part is bound by its staging traffic (a 64-k interval moves 32 KB for
  // 128 MFMAs per CU); a 256-row tile does 2x the MFMAs for 1.5x the bytes, and the lost tile count is made up by split-K.  Measured
  // per 64-k interval and round (tools/lab/dw_sweep.py, profiles/r06_dw_sweep_pp.txt): 0.70 us at 128 rows, 0.95 at 192, 1.02 at 256,
  // + ~9 us per launch, + the slab traffic of a split.  The plan takes the cheapest of {128 as before, 192, 256 with the split count
  // that fills one round}, and leaves 128 unless the model says >= 8 % (1024 x 3072 x 9082: 109.9 -> 81.5 us, 1024 x 6912 x 4608:
  // 101.1 -> 78.3, 3072 x 1024 x 4608: 51.4 -> 46.0; the 1024 x 1024 and x 4096 shapes keep their plan).
  static const bool dw_tall = [] { const char* e = getenv("VILCO_GEMM_DW_TALL"); return !(e && e[0] == '0'); }();
  if (dw_tall && d->precision == 4 && p.k2 && !p.gl && nbatch == 1 && p.BM == 128) {
    auto model = [&](int bm, int k, double c) {
      const long t = ((d->M + bm - 1) / bm) * tn;
      const long rounds = (t * k + 255) / 256;
      const double split = k > 1 ? 3.0 + (double)d->M * d->N * 4.0 * (k + 1) / 5e6 : 0.0;
      return 9.0 + (double)rounds * ((nk + k - 1) / k) * c + split;
    };
    const double base = model(128, ks, 0.70);
    double best_t = base; int best_bm = 128, best_ks = ks;
    const int bms[2] = {192, 256}; const double cs[2] = {0.95, 1.02};
    for (int q = 0; q < 2; ++q) {
      const long t = ((d->M + bms[q] - 1) / bms[q]) * tn;
      int k = t >= 256 ? 1 : (int)(256 / t);
      if (k > nk / 8) k = nk / 8;
      if (k > 16) k = 16;
      if (k < 1) k = 1;
      const double m = model(bms[q], k, cs[q]);
      if (m < best_t) { best_t = m; best_bm = bms[q]; best_ks = k; }
    }
    if (best_bm != 128 && best_t <= 0.92 * base) { p.BM = best_bm; ks = best_ks; }
  }
  // tuning overrides (tools/gemm_tune.py): VILCO_GEMM_BM = 128|256, VILCO_GEMM_KS = forced split count
  // (read once per process: the plan is made twice per launch)
  const int force_bm = tune().bm, force_ks = tune().ks;
  if (force_bm == 128 || (force_bm == 256 && !p.gl) || (force_bm == 192 && d->precision >= 3)) p.BM = force_bm;
  if (force_ks >= 1 && force_ks <= nk) ks = force_ks;
  // few-row NT products of the default precision: one launch of gemm_skinny_kernel instead of a split-K plan + its reduce launch
  p.skinny = false;
  if (skinny_enabled() && force_bm == 0 && force_ks < 1 && d->precision == 3 && gl_enabled() && !p.a_tr && !p.b_tr && !p.a_km && !p.b_km &&
      d->tap_operand == VILCO_TAP_NONE && d->band == 0 && nbatch == 1 && d->M <= skinny_max_rows() && d->K >= 64 && d->K <= skinny_max_k() &&
      d->N <= 1024) {
    // (tools/lab/skinny_ab.py, hipGraph replays: 16..512 x 512 x 512 7.3-8.8 us against 10.7-12.2 tiled, 154 x 1024 x 768 9.8 against
    // 12.5; from K = 1024 or N = 2048 on the 16..32-row tiles' re-reads of B cost more than the second launch: 288 x 1024 x 1024 14.3
    // against 13.3, 576 x 1024 x 4096 91 against 28)
    p.skinny = true;
    p.gl = false;
    p.BM = d->M <= skinny_bm16_rows() ? 16 : 32;       // (64 rows per workgroup spill at two waves per SIMD)
    ks = 1;
  }
  p.kchunk = (nk + ks - 1) / ks;
  p.ksplit = (nk + p.kchunk - 1) / p.kchunk;
  long out_span = 0;
  if (p.ksplit > 1) {   // a split slab covers the whole (strided) output footprint
    out_span = (long)(d->batch_outer - 1) * d->sCo + (long)(d->batch_inner - 1) * d->sCi + (long)(d->M - 1) * d->ldc + d->N;
    out_span = align_up(out_span, 64);
  }
  p.split_stride = out_span;
  p.part_bytes = p.ksplit > 1 ? align_up(out_span * p.ksplit * 4, 256) : 0;
  p.fixup = false;
  if (p.ksplit > 1 && fixup_enabled() && !p.gl) {
    const long ntiles = tn * ((d->M + p.BM - 1) / p.BM) * nbatch;
    if (ntiles <= FIX_TILES) {      // slots [tile][split] of BM x 128 floats in the accumulators' register layout
      p.fixup = true;
      p.part_bytes = align_up(ntiles * p.ksplit * (long)p.BM * BN * 4, 256);
    }
  }
}

template <int BM, int NP, bool F16, bool AKM, bool BKM, bool K2 = false>
void launch_pp_km(const GArgs& g, dim3 grid, hipStream_t s) {
  constexpr size_t a_el = AKM ? 32 * ((BM == 192 ? 256 : BM) + 16) : BM * 32, b_el = BKM ? 32 * (BN + 16) : BN * 32;
  constexpr size_t pipe = (size_t)2 * NP * (a_el + b_el) * sizeof(__bf16), epi = (size_t)8 * 16 * (BM / 64) * EPI_LD * 4;
  constexpr size_t lds = pipe > epi ? pipe : epi;
  static const bool once = [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<BM, NP, F16, AKM, BKM, K2>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    return true;
  }();
  (void)once;
  hipLaunchKernelGGL((gemm_pp_kernel<BM, NP, F16, AKM, BKM, K2>), grid, dim3(512), lds, s, g);
}

template <int BM, bool AKM, bool BKM, bool SINGLE>
void launch_gl_km(const GArgs& g, dim3 grid, hipStream_t s) {
  constexpr size_t lds = (size_t)4 * (BM + BN) * 128;          // two stages x two slots x (A + B) x 128 B
  static const bool once = [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_gl_kernel<BM, AKM, BKM, SINGLE>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    return true;
  }();
  (void)once;
  hipLaunchKernelGGL((gemm_gl_kernel<BM, AKM, BKM, SINGLE>), grid, dim3(512), lds, s, g);
}
template <int BM, bool SINGLE>
void launch_gl(const GArgs& g, dim3 grid, hipStream_t s, bool akm, bool bkm) {
  if (akm && bkm) launch_gl_km<BM, true, true, SINGLE>(g, grid, s);
  else if (bkm) launch_gl_km<BM, false, true, SINGLE>(g, grid, s);
  else launch_gl_km<BM, false, false, SINGLE>(g, grid, s);
}

// operand orientations in use: (kc,kc) forward / convs, (kc,km) dX = dY W, (km,km) dW = dY^T X
template <int BM, int NP, bool F16 = false, bool K2 = false>
void launch_pp(const GArgs& g, dim3 grid, hipStream_t s, bool akm = false, bool bkm = false) {
  if (akm && bkm) launch_pp_km<BM, NP, F16, true, true, K2>(g, grid, s);
  else if (bkm) launch_pp_km<BM, NP, F16, false, true, K2>(g, grid, s);
  else launch_pp_km<BM, NP, F16, false, false, K2>(g, grid, s);
}

}  // namespace

#ifdef VILCO_LAB
extern "C" int vilco_lab_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vilco_lab_stamps), sizeof(unsigned long long) * 2 * 64 * 8);
}
#endif

// ---- optional timing of the MFMA kernel alone (bench.py's roofline line): HIP events on the caller's stream
namespace {
template <int BMS>
static void launch_skinny(const GArgs& g, hipStream_t s) {
  constexpr size_t lds = (size_t)8 * BMS * SK_LD * 4;
  static const bool once = [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<BMS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    return true;
  }();
  (void)once;
  const int tiles = ((g.N + 63) / 64) * ((g.M + BMS - 1) / BMS);
  hipLaunchKernelGGL((gemm_skinny_kernel<BMS>), dim3(tiles), dim3(512), lds, s, g);
}

struct ProfRec { int64_t v[10]; };   // M, N, K, batch, BM, ksplit, precision, a_km, b_km, tap
struct ProfState {
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  std::vector<ProfRec> rec;          // one per event pair
  std::vector<double> ms;            // filled by profile_end, read by profile_records
};
ProfState& prof() { static ProfState p; return p; }
}  // namespace

// number of max|C| partials a vilco_gemm call with this descriptor writes to desc->amax_out (0: too many -- do not ask)
static long amax_out_parts(const vilco_gemm_desc* d, const Plan& p) {
  const long nz = (long)d->batch_outer * d->batch_inner;
  if (p.ksplit > 1 && !p.fixup) {
    long blocks = ((long)d->M * d->N * nz + 255) / 256;
    return blocks > 2048 ? 2048 : blocks;
  }
  if (p.skinny) return (long)((d->N + 63) / 64) * ((d->M + p.BM - 1) / p.BM);
  const long n = (long)((d->N + BN - 1) / BN) * ((d->M + p.BM - 1) / p.BM) * nz;
  return n <= 8192 ? n : 0;
}

extern "C" int32_t vilco_gemm_amax_parts(const vilco_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->band == 1) return 0;      // band 1 leaves tiles unwritten
  Plan p;
  make_plan(d, p);
  return (int32_t)amax_out_parts(d, p);
}

extern "C" int vilco_gemm_profile_begin(void) {
  ProfState& p = prof();
  for (auto& e : p.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  p.ev.clear();
  p.rec.clear();
  p.ms.clear();
  p.on = true;
  return VILCO_OK;
}

extern "C" int vilco_gemm_profile_end(double* kernel_ms, int64_t* launches) {
  ProfState& p = prof();
  p.on = false;
  double total = 0.0;
  for (auto& e : p.ev) {
    if (hipEventSynchronize(e.second) != hipSuccess) return VILCO_ERR_LAUNCH;
    float ms = 0.f;
    hipEventElapsedTime(&ms, e.first, e.second);
    total += ms;
    p.ms.push_back(ms);
  }
  if (kernel_ms) *kernel_ms = total;
  if (launches) *launches = (int64_t)p.ev.size();
  for (auto& e : p.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  p.ev.clear();
  return VILCO_OK;
}

// per-launch records of the last begin/end bracket: desc[i*10 ..] = M, N, K, batch, BM, ksplit, precision, a_km, b_km, tap
extern "C" int64_t vilco_gemm_profile_records(int64_t* desc, double* ms, int64_t cap) {
  ProfState& p = prof();
  const int64_t n = (int64_t)std::min(p.rec.size(), p.ms.size());
  for (int64_t i = 0; i < n && i < cap; ++i) {
    if (desc) for (int j = 0; j < 10; ++j) desc[i * 10 + j] = p.rec[i].v[j];
    if (ms) ms[i] = p.ms[i];
  }
  return n;
}

static inline int np_of_precision(int precision) { return precision == 1 ? 1 : ((precision == 0 || precision == 3) ? 2 : 3); }

extern "C" size_t vilco_pack_bytes(int64_t rows, int64_t cols, int32_t precision) {
  if (rows < 0 || cols < 0) return 0;
  return (size_t)(PACK_HDR + align_up(rows > 0 ? rows : 1, 32) * align_up(cols > 0 ? cols : 1, 32) * 2 * np_of_precision(precision));
}

static inline long item_cols(const vilco_pack_item& it) { return it.relshift ? it.rows + it.cols : it.cols; }

// rows of a zero-padded per-sequence plane image (vilco_pack_item.seq_len): nseq * (T + 2) padded rows + enough zero rows for
// both readers -- the forward / dX conv's overlapped spans (make_plan: a_out_rows) and the weight-gradient product's
// contraction over the padded rows (Kp + 3 + one K-step)
static inline long tap_plane_rows(long nseq, long T) { return vilco_tap_plane_rows(nseq, T); }

extern "C" size_t vilco_pack_item_bytes(const vilco_pack_item* it, int32_t precision) {
  if (!it || it->rows <= 0 || it->cols <= 0) return 0;
  if (it->seq_len > 0)
    return (size_t)(PACK_HDR + align_up(tap_plane_rows(it->rows / it->seq_len, it->seq_len) * it->cols, 8) * 2 * np_of_precision(precision));
  const long nb = it->nbatch > 1 ? it->nbatch : 1;
  return (size_t)(PACK_HDR + nb * align_up(it->rows, 32) * align_up(item_cols(*it), 32) * 2 * np_of_precision(precision));
}

extern "C" int vilco_pack_many(const vilco_pack_item* items, int32_t n, int32_t precision, void* stream) {
  if (!items || n < 1 || n > 4 || precision < 0 || precision > 3) return VILCO_ERR_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int NP = np_of_precision(precision);
  PackArgs4 pk;
  AmaxArgs am;
  bool have_amax[4] = {false, false, false, false};
  const int nbatch = items[0].nbatch > 1 ? items[0].nbatch : 1;
  for (int i = 0; i < n; ++i) {
    const vilco_pack_item& it = items[i];
    if (!it.src || !it.planes || it.rows <= 0 || it.cols <= 0 || it.ld < it.cols) return VILCO_ERR_BADARG;
    if (it.rows > 0x7fffffff || it.cols > 0x3fffffff || !vilco_aligned(it.planes, 256)) return VILCO_ERR_BADARG;
    if ((it.nbatch > 1 ? it.nbatch : 1) != nbatch) return VILCO_ERR_UNSUPPORTED;      // one launch = one batch count
    if (it.planes_bytes < vilco_pack_item_bytes(&it, precision)) return VILCO_ERR_WORKSPACE;
    const long cols = item_cols(it);
    const long rows32 = align_up(it.rows, 32), cols32 = align_up(cols, 32);
    float* hdr = reinterpret_cast<float*>(it.planes);
    PackArgs& pa = pk.a[i];
    pa.src = it.src; pa.dst = reinterpret_cast<__bf16*>(reinterpret_cast<unsigned char*>(it.planes) + PACK_HDR);
    pa.ld = it.ld; pa.rows = (int)it.rows; pa.K = (int)cols; pa.Kp = (int)cols32;
    pa.plane_stride = rows32 * cols32 * nbatch; pa.batch_stride = rows32 * cols32; pa.nbi = nbatch;
    pa.so = 0; pa.si = nbatch > 1 ? it.batch_stride : 0;
    pa.tap = it.relshift ? 4 : 0; pa.tapC = it.relshift ? (int)it.cols : 1; pa.tapT = 1; pa.out_rows = (int)rows32;
    pa.vec = vilco_aligned(it.src, 16) && (it.ld % 4) == 0 && (it.batch_stride % 4) == 0;
    if (it.seq_len > 0) {          // the k=3 convs' image: [nseq * (T + 2) + zero rows][cols], first and last row of a sequence zero
      if (it.relshift || nbatch != 1 || (it.cols % 8) != 0 || (it.rows % it.seq_len) != 0) return VILCO_ERR_UNSUPPORTED;
      const long tr = tap_plane_rows(it.rows / it.seq_len, it.seq_len);
      pa.tap = 1; pa.tapC = (int)it.cols; pa.tapT = it.seq_len; pa.out_rows = (int)tr;
      pa.plane_stride = align_up(tr * it.cols, 8); pa.batch_stride = pa.plane_stride;
    }
    pa.amax = nullptr; pa.namax = 0; pa.inv_scale = hdr + AMAX_MAX_BLOCKS;
    if (precision == 3) {
      PackArgs src_view = pa;                     // amax runs over the source matrix itself ([rows][cols])
      src_view.K = (int)it.cols; src_view.tap = 0;
      am.op[i] = amax_view(src_view, false, 1, hdr);
      pa.amax = hdr; pa.namax = am.op[i].nblocks;
      if (it.amax && it.namax > 0) { pa.amax = it.amax; pa.namax = it.namax; have_amax[i] = true; }   // left by the producer
    }
  }
  if (precision == 3) {
    int todo = 0;
    for (int i = 0; i < n; ++i) todo += have_amax[i] ? 0 : 1;
    if (todo == n && dispatch_pack_fused(NP, pk.a, am.op, n, nbatch, s, VILCO_SITE_PACK)) return vilco_launch_status();   // amax + pack: 1 launch
    if (todo) {
      AmaxArgs am2;
      int k = 0;
      for (int i = 0; i < n; ++i)
        if (!have_amax[i]) am2.op[k++] = am.op[i];
      launch_amax(am2, k, s);
    }
  }
  if (n == 1 && nbatch == 1) dispatch_pack(NP, pk.a[0], false, 1, s);
  else dispatch_pack_multi(NP, pk, n, s, nbatch);
  return vilco_launch_status();
}

extern "C" int vilco_pack(const float* src, int64_t rows, int64_t cols, int64_t ld, int32_t precision, void* planes,
                          size_t planes_bytes, void* stream) {
  const vilco_pack_item it = {src, rows, cols, ld, planes, planes_bytes, 1, 0, 0, nullptr, 0};
  return vilco_pack_many(&it, 1, precision, stream);
}

extern "C" size_t vilco_gemm_workspace(const vilco_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K < 0) return 512;
  Plan p;
  make_plan(d, p);
  return (size_t)(p.a_bytes + p.b_bytes + p.part_bytes + 512 + SCALE_BYTES);
}

// gout / pout non-null: validate, plan and fill the kernel arguments of a product whose operands are already packed, WITHOUT
// launching anything (vilco_gemm_group collects the members of a grouped launch this way)
static int gemm_impl(const vilco_gemm_desc* d, void* stream, GArgs* gout, Plan* pout) {
  if (!d || !d->C || (!d->A && !d->a_planes) || (!d->B && !d->b_planes)) return VILCO_ERR_BADARG;
  if (gout && !(d->a_planes && d->b_planes)) return VILCO_ERR_UNSUPPORTED;
  if (d->a_planes || d->b_planes) {
    // pre-packed operands: untapped problems, or the plain (weight) operand B of a k=3 conv whose taps are on A
    // (k=3 convs: the tapped operand's planes must be the zero-padded per-sequence image, vilco_pack_item.seq_len; checked
    //  against the plan below)
    const bool tap_ok = d->tap_operand == VILCO_TAP_NONE || (d->tap_operand == VILCO_TAP_A && d->b_kcontig) ||
                        d->tap_operand == VILCO_TAP_B;
    if (!tap_ok || !use_km()) return VILCO_ERR_UNSUPPORTED;
    if (!vilco_aligned(d->a_planes, 256) || !vilco_aligned(d->b_planes, 256)) return VILCO_ERR_BADARG;
  }
  if (d->M < 0 || d->N < 0 || d->K < 0 || d->batch_outer < 1 || d->batch_inner < 1) return VILCO_ERR_BADARG;
  if (d->M == 0 || d->N == 0) return VILCO_OK;
  if (d->precision < 0 || d->precision > 4) return VILCO_ERR_BADARG;
  if (d->act < 0 || d->act > 2) return VILCO_ERR_BADARG;
  if (d->band < 0 || d->band > 3 || (d->band && d->bandT <= 0)) return VILCO_ERR_BADARG;
  if (!(d->drop_p >= 0.f) || d->drop_p >= 1.f) return VILCO_ERR_BADARG;
  if (d->drop_p > 0.f && (d->ldc != d->N || d->batch_outer != 1 || d->batch_inner != 1)) return VILCO_ERR_UNSUPPORTED;
  if (d->row_len && d->rowT <= 0) return VILCO_ERR_BADARG;
  if (d->row_mask && (d->batch_outer != 1 || d->batch_inner != 1)) return VILCO_ERR_UNSUPPORTED;
  if (d->a_kcontig == 0 && d->b_kcontig == 1) return VILCO_ERR_UNSUPPORTED;  // "TT" is never needed
  if (d->tap_operand != VILCO_TAP_NONE) {
    if (d->tapC <= 0 || d->tapT <= 0) return VILCO_ERR_BADARG;
    if (d->batch_outer != 1 || d->batch_inner != 1) return VILCO_ERR_UNSUPPORTED;
    if (d->tap_operand == VILCO_TAP_A) {
      if (!(d->a_kcontig == 1 && d->K == 3 * d->tapC && d->lda == d->tapC && (d->M % d->tapT) == 0)) return VILCO_ERR_BADARG;
    } else if (d->tap_operand == VILCO_TAP_B) {
      if (!(d->b_kcontig == 0 && d->N == 3 * d->tapC && d->ldb == d->tapC && (d->K % d->tapT) == 0)) return VILCO_ERR_BADARG;
    } else {
      return VILCO_ERR_BADARG;
    }
  }
  Plan p;
  make_plan(d, p);
  // operand planes must come in the layout the product reads: natural for plain operands, the per-sequence image for tapped ones
  {
    const bool a_seq = d->a_planes && d->a_planes_seq, b_seq = d->b_planes && d->b_planes_seq;
    const bool a_want = d->a_planes && ((d->tap_operand == VILCO_TAP_A && p.a_tap == 1) || p.tapkm);
    const bool b_want = d->b_planes && p.tapkm;
    if (a_seq != a_want || b_seq != b_want) return VILCO_ERR_UNSUPPORTED;
    if (d->tap_operand == VILCO_TAP_A && d->a_planes && p.a_tap != 1) return VILCO_ERR_UNSUPPORTED;   // tapC % 8 != 0: expanded rows
    if (d->tap_operand == VILCO_TAP_B && (d->a_planes || d->b_planes) && !(p.tapkm && d->a_planes && d->b_planes)) return VILCO_ERR_UNSUPPORTED;
  }
  // the kernel's staging loads address one operand matrix (one batch element of one part) with 32-bit byte offsets
  // (buffer loads: lane offset + K-step offset, gemm_pp_kernel); the last K-step may over-read one tile
  {
    const long lim = (1L << 31) - (1L << 20);
    const long a_el = p.a_batch > (long)align_up(d->M, 32) * p.Kp ? p.a_batch : (long)align_up(d->M, 32) * p.Kp;
    const long b_el = p.b_batch > (long)align_up(d->N, 32) * p.Kp ? p.b_batch : (long)align_up(d->N, 32) * p.Kp;
    if (a_el * 2 >= lim || b_el * 2 >= lim) return VILCO_ERR_UNSUPPORTED;
  }
  const size_t need = (size_t)(p.a_bytes + p.b_bytes + p.part_bytes + 512 + SCALE_BYTES);
  if (!d->workspace || d->workspace_bytes < need) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);

  unsigned char* ws = reinterpret_cast<unsigned char*>(align_up((long)reinterpret_cast<uintptr_t>(d->workspace), 256));
  float* scales = reinterpret_cast<float*>(ws);            // [A partials | B partials | 1/sA, 1/sB]
  ws += SCALE_BYTES;
  __bf16* planesA = reinterpret_cast<__bf16*>(ws);
  __bf16* planesB = reinterpret_cast<__bf16*>(ws + p.a_bytes);
  float* parts = reinterpret_cast<float*>(ws + p.a_bytes + p.b_bytes);
  const bool f16 = d->precision == 3 || d->precision == 4;      // 4: fp16 x2 planes, the product uses their first parts only

  // ---- pack A and B into bf16 planes
  PackArgs pa;
  pa.src = d->A; pa.dst = planesA; pa.ld = d->lda; pa.rows = d->M; pa.K = d->K; pa.Kp = p.Kp;
  pa.plane_stride = p.a_plane; pa.batch_stride = p.a_batch; pa.nbi = p.a_nbi;
  pa.so = p.a_nbo > 1 ? d->sAo : 0; pa.si = p.a_nbi > 1 ? d->sAi : 0;
  pa.tap = p.a_tap; pa.tapC = d->tapC > 0 ? d->tapC : 1; pa.tapT = d->tapT > 0 ? d->tapT : 1;
  pa.out_rows = p.a_out_rows;
  if (p.a_km) { pa.rows = d->K; pa.K = d->M; pa.Kp = (int)align_up(d->M, 32); }   // natural [K][M] view
  if (p.tapkm) { pa.tapC = d->M; pa.K = d->M; }                                     // dZ rows: Cout wide, zero-padded per sequence
  pa.vec = vilco_aligned(d->A, 16) && (d->lda % 4) == 0 && (d->sAo % 4) == 0 && (d->sAi % 4) == 0;
  pa.amax = f16 ? scales : nullptr; pa.namax = 0; pa.inv_scale = scales + 2 * AMAX_MAX_BLOCKS;
  const bool packA = d->a_planes == nullptr, packB = d->b_planes == nullptr;

  PackArgs pb;
  pb.src = d->B; pb.dst = planesB; pb.ld = d->ldb; pb.rows = d->N; pb.K = d->K; pb.Kp = p.Kp;
  pb.plane_stride = p.b_plane; pb.batch_stride = p.b_batch; pb.nbi = p.b_nbi;
  pb.so = p.b_nbo > 1 ? d->sBo : 0; pb.si = p.b_nbi > 1 ? d->sBi : 0;
  pb.tap = p.b_tap; pb.tapC = d->tapC > 0 ? d->tapC : 1; pb.tapT = d->tapT > 0 ? d->tapT : 1;
  pb.out_rows = p.b_out_rows;
  if (p.b_km) { pb.rows = d->K; pb.K = d->N; pb.Kp = (int)align_up(d->N, 32); }   // natural [K][N] view
  if (p.tapkm) { pb.K = d->tapC; }                                                  // X rows: Cin wide
  pb.vec = vilco_aligned(d->B, 16) && (d->ldb % 4) == 0 && (d->sBo % 4) == 0 && (d->sBi % 4) == 0;
  pb.amax = f16 ? scales + AMAX_MAX_BLOCKS : nullptr; pb.namax = 0; pb.inv_scale = scales + 2 * AMAX_MAX_BLOCKS + 2;
  bool doneA = !packA, doneB = !packB;
  if (f16 && (packA || packB)) {
    // k-contiguous operands: amax + pack in one launch each (grid barrier, pack.h); transposing packs keep the amax launch
    AmaxOp ma = amax_view(pa, p.a_tr, p.a_nbo, scales), mb = amax_view(pb, p.b_tr, p.b_nbo, scales + AMAX_MAX_BLOCKS);
    if (packA && !p.a_tr && !d->a_amax) doneA = dispatch_pack_fused(p.NP, &pa, &ma, 1, p.a_nbo * p.a_nbi, s, VILCO_SITE_GEMMPACK);
    if (packB && !p.b_tr && !d->b_amax) doneB = dispatch_pack_fused(p.NP, &pb, &mb, 1, p.b_nbo * p.b_nbi, s, VILCO_SITE_GEMMPACK);
    AmaxArgs am;
    int nops = 0;
    // partials left by the producer of A / B (vilco_gemm_desc.a_amax / b_amax) replace the amax pass over that operand
    if (!doneA) {
      if (d->a_amax && d->a_namax > 0) { pa.amax = d->a_amax; pa.namax = d->a_namax; }
      else { am.op[nops++] = ma; pa.namax = ma.nblocks; }
    }
    if (!doneB) {
      if (d->b_amax && d->b_namax > 0) { pb.amax = d->b_amax; pb.namax = d->b_namax; }
      else { am.op[nops++] = mb; pb.namax = mb.nblocks; }
    }
    if (nops) launch_amax(am, nops, s);
  }
  if (!doneA) dispatch_pack(p.NP, pa, p.a_tr, p.a_nbo * p.a_nbi, s);
  if (!doneB) dispatch_pack(p.NP, pb, p.b_tr, p.b_nbo * p.b_nbi, s);
  const float* inv_a = pa.inv_scale;
  const float* inv_b = pb.inv_scale;
  if (!packA) {    // planes from vilco_pack: [part][rows32][cols32] behind the header
    const unsigned char* u = reinterpret_cast<const unsigned char*>(d->a_planes);
    planesA = reinterpret_cast<__bf16*>(const_cast<unsigned char*>(u) + PACK_HDR);
    inv_a = reinterpret_cast<const float*>(u) + AMAX_MAX_BLOCKS;
    p.a_batch = p.a_km ? (long)p.Kp * align_up(d->M, 32) : align_up(d->M, 32) * (long)p.Kp;
    if (p.a_tap == 1)                                // the convs' zero-padded image (vilco_pack_item.seq_len)
      p.a_batch = p.tapkm ? align_up(tap_plane_rows(d->K / d->tapT, d->tapT) * d->M, 8)
                          : align_up(tap_plane_rows(d->M / d->tapT, d->tapT) * d->tapC, 8);
    p.a_plane = p.a_batch * p.a_nbo * p.a_nbi;       // batched planes: [part][batch][rows32][cols32]
  }
  if (!packB) {
    const unsigned char* u = reinterpret_cast<const unsigned char*>(d->b_planes);
    planesB = reinterpret_cast<__bf16*>(const_cast<unsigned char*>(u) + PACK_HDR);
    inv_b = reinterpret_cast<const float*>(u) + AMAX_MAX_BLOCKS;
    p.b_batch = p.b_km ? (long)p.Kp * align_up(d->N, 32) : align_up(d->N, 32) * (long)p.Kp;
    if (p.tapkm) p.b_batch = align_up(tap_plane_rows(d->K / d->tapT, d->tapT) * d->tapC, 8);
    p.b_plane = p.b_batch * p.b_nbo * p.b_nbi;
  }

  // ---- MFMA kernel
  GArgs g;
  g.a.p = planesA; g.a.plane_stride = p.a_plane; g.a.batch_stride = p.a_batch; g.a.nbi = p.a_nbi;
  g.a.has_o = p.a_nbo > 1; g.a.has_i = p.a_nbi > 1; g.a.rows = d->M;
  if (p.a_tap == 1) { g.a.seqT = d->tapT; g.a.seq_stride = (long)(d->tapT + 2) * d->tapC; g.a.row_stride = d->tapC; }
  else { g.a.seqT = 0x7fffffff; g.a.seq_stride = 0; g.a.row_stride = p.Kp; }
  g.a.cols = 0;
  if (p.a_km) { g.a.row_stride = align_up(d->M, 32); g.a.cols = (int)align_up(d->M, 32); }
  if (p.tapkm) { g.a.p = planesA + d->M; g.a.row_stride = d->M; g.a.cols = d->M; }      // dZ' = padded rows 1 ...
  g.b.p = planesB; g.b.plane_stride = p.b_plane; g.b.batch_stride = p.b_batch; g.b.nbi = p.b_nbi;
  g.b.has_o = p.b_nbo > 1; g.b.has_i = p.b_nbi > 1; g.b.rows = d->N;
  g.b.seqT = 0x7fffffff; g.b.seq_stride = 0; g.b.row_stride = p.Kp;
  g.b.cols = 0;
  if (p.b_km) { g.b.row_stride = align_up(d->N, 32); g.b.cols = (int)align_up(d->N, 32); }
  if (p.tapkm) { g.b.row_stride = d->tapC; g.b.cols = d->N; }                            // X' seen as [rows][3 Cin], row stride Cin
  g.a.bytes = 2 * (p.a_batch - (g.a.p - planesA)); g.b.bytes = 2 * p.b_batch;      // one part of one batch (gemm_gl_kernel's buffer range)
  g.ldc = d->ldc; g.M = d->M; g.N = d->N; g.Kp = p.Kp;
  g.batch_inner = d->batch_inner; g.sCo = d->sCo; g.sCi = d->sCi;
  g.tiles_n = (d->N + BN - 1) / BN;
  g.ntiles = g.tiles_n * ((d->M + p.BM - 1) / p.BM);
  g.tm0 = 0;
  g.ksplit = p.ksplit; g.kchunk = p.kchunk; g.split_stride = p.split_stride;
  g.inv_a = inv_a; g.inv_b = inv_b;
  g.band = d->band; g.bandT = d->bandT;
  g.drop_thresh = vilco_drop_threshold_host(d->drop_p); g.drop_seed = d->drop_seed; g.drop_inv_keep = 1.f / (1.f - d->drop_p); g.seed_word = vilco_seed_word_dev();
  g.vec_out = (d->N % 4) == 0 && (d->ldc % 4) == 0 && (d->sCo % 4) == 0 && (d->sCi % 4) == 0 && vilco_aligned(d->C, 16) &&
              vilco_aligned(d->bias, 16) && vilco_aligned(d->preact, 16) && vilco_aligned(d->colscale, 16) &&
              vilco_aligned(d->residual, 16);
  g.c = p.ksplit > 1 ? parts : d->C;
  g.cfinal = d->C;
  g.tile_ctr = nullptr;
  if (p.fixup) {
    g.tile_ctr = tile_counters(s);
    if (!g.tile_ctr) return VILCO_ERR_UNSUPPORTED;      // more than FIX_STREAMS streams in one process
  }
  g.amax_out = d->band == 1 ? nullptr : d->amax_out;
  g.e = Epi{d->alpha, d->beta, d->bias, d->preact, d->act, d->row_len, d->rowT, d->colscale, d->residual,
            d->res_masked, d->row_mask};
  const int nz = d->batch_outer * d->batch_inner;
  if (gout) { *gout = g; *pout = p; return VILCO_OK; }
  dim3 grid(g.ntiles, p.ksplit, nz);
  // Between one and two rounds of 192-row tiles (384 tiles on 256 CUs: the k = 3 head convs at 9082 rows, 4608 x 2048, 2304 x 4096 ...)
  // the second round is half empty and costs a whole one.  Two launches instead: ONE full round of 192-row tiles over the first
  // rows, the remaining rows as one round of 128-row tiles (a round of those costs 0.74 of a 192-row round, see make_plan):
  // 1.74 instead of 2 round-units on paper.  Same tiles' arithmetic as either height alone (a tile's K order does not depend on
  // its height), row ranges disjoint, bit-identical results (tests/test_ops_gpu.py).  MEASURED (tools/lab/tail128_ab.sh, operands
  // packed): 9082 x 1024 x 3072 184.9 -> 178.4 us, 4608 x 2048 x 1024 / 2304 x 4096 x 1024 / 9216 x 1024 x 1024 unchanged (69 us), the
  // P step 20.49 -> 20.63 ms: the second launch starts only when the first has drained, and a lone round of 128-row tiles costs
  // more than the 0.74 it costs inside a multi-round launch.  OFF by default (VILCO_GEMM_TAIL128=1 / vilco_gemm_set_tail128(1)).
  if (tail128_enabled() && p.gl && d->precision == 3 && p.BM == 192 && p.ksplit == 1 && nz == 1 && !g.amax_out && d->band == 0 &&
      g.ntiles > 256 && g.ntiles <= 512 && g.tiles_n <= 128 && (256 % g.tiles_n) == 0) {
    const int rows1 = (256 / g.tiles_n) * 192;                        // one full round of 192-row tiles (a multiple of 128 rows)
    const long rem_tiles = (long)((d->M - rows1 + 127) / 128) * g.tiles_n;
    if (rows1 < d->M && rem_tiles <= 256) {
      const bool ak = p.a_km, bk = p.b_km;
      GArgs g1 = g, g2 = g;
      g1.ntiles = 256; g1.tm0 = 0;
      g2.ntiles = (int)rem_tiles; g2.tm0 = rows1 / 128;
      hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;      // (every profile pair owns its two events)
      if (prof().on) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3); hipEventRecord(e0, s); }
      launch_gl<192, false>(g1, dim3(g1.ntiles, 1, 1), s, ak, bk);
      if (e0) { hipEventRecord(e1, s); hipEventRecord(e2, s); }
      launch_gl<128, false>(g2, dim3(g2.ntiles, 1, 1), s, ak, bk);
      if (e0) {
        hipEventRecord(e3, s);
        prof().ev.emplace_back(e0, e1);
        prof().rec.push_back(ProfRec{{rows1, d->N, d->K, nz, 192, 1, d->precision, p.a_km, p.b_km, p.a_tap | (p.b_tap << 4)}});
        prof().ev.emplace_back(e2, e3);
        prof().rec.push_back(ProfRec{{d->M - rows1, d->N, d->K, nz, 128, 1, d->precision, p.a_km, p.b_km, p.a_tap | (p.b_tap << 4)}});
      }
      return vilco_launch_status();
    }
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (prof().on) { hipEventCreate(&ev0); hipEventCreate(&ev1); hipEventRecord(ev0, s); }
  {
    const bool ak = p.a_km, bk = p.b_km;
    if (p.skinny) {
      if (p.BM == 16) launch_skinny<16>(g, s); else launch_skinny<32>(g, s);
    } else if (p.gl) {
      const bool single = d->precision == 4;
      if (p.BM == 192) { if (single) launch_gl<192, true>(g, grid, s, ak, bk); else launch_gl<192, false>(g, grid, s, ak, bk); }
      else { if (single) launch_gl<128, true>(g, grid, s, ak, bk); else launch_gl<128, false>(g, grid, s, ak, bk); }
    } else if (d->precision == 4 && p.k2) {  // one MFMA per product on the leading fp16 parts, two K-steps per barrier interval
      if (p.BM == 256) launch_pp<256, 2, true, true>(g, grid, s, ak, bk);
      else if (p.BM == 192) launch_pp<192, 2, true, true>(g, grid, s, ak, bk);
      else launch_pp<128, 2, true, true>(g, grid, s, ak, bk);
    } else if (d->precision == 4) {   // (band-limited products: one K-step per interval; plane strides unchanged)
      if (p.BM == 256) launch_pp<256, 1, true>(g, grid, s, ak, bk);
      else if (p.BM == 192) launch_pp<192, 1, true>(g, grid, s, ak, bk);
      else launch_pp<128, 1, true>(g, grid, s, ak, bk);
    } else if (f16) {
      if (p.BM == 256) launch_pp<256, 2, true>(g, grid, s, ak, bk);
      else if (p.BM == 192) launch_pp<192, 2, true>(g, grid, s, ak, bk);
      else launch_pp<128, 2, true>(g, grid, s, ak, bk);
    }
    else if (p.BM == 256) {
      if (p.NP == 1) launch_pp<256, 1>(g, grid, s, ak, bk);
      else if (p.NP == 2) launch_pp<256, 2>(g, grid, s, ak, bk);
      else launch_pp<256, 3>(g, grid, s, ak, bk);
    } else {
      if (p.NP == 1) launch_pp<128, 1>(g, grid, s, ak, bk);
      else if (p.NP == 2) launch_pp<128, 2>(g, grid, s, ak, bk);
      else launch_pp<128, 3>(g, grid, s, ak, bk);
    }
  }
  if (ev0) {
    hipEventRecord(ev1, s);
    prof().ev.emplace_back(ev0, ev1);
    prof().rec.push_back(ProfRec{{d->M, d->N, d->K, nz, p.BM, p.ksplit, d->precision, p.a_km, p.b_km, p.a_tap | (p.b_tap << 4)}});
  }
  if (p.ksplit > 1 && !p.fixup && vilco_defer_active() && nz == 1 && d->alpha == 1.f && d->beta == 0.f && !d->bias &&
      !d->preact && d->act == VILCO_ACT_NONE && !d->row_len && !d->row_mask && !d->colscale && !d->residual && d->drop_p == 0.f &&
      !d->amax_out) {
    // a plain sum of the split slabs whose result nothing reads before the end of backward (a weight gradient): recorded
    vilco_defer_push_sk(g.c, g.cfinal, g.split_stride, g.ldc, d->M, d->N, p.ksplit);
  } else if (p.ksplit > 1 && !p.fixup) {
    long blocks = ((long)d->M * d->N * nz + 255) / 256;       // == amax_out_parts(): one max|C| partial per block, either form
    if (blocks > 2048) blocks = 2048;
    if (g.vec_out) hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3((int)blocks), dim3(256), 0, s, g, nz);
    else hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3((int)blocks), dim3(256), 0, s, g, nz);
  }
  return vilco_launch_status();
}

extern "C" int vilco_gemm(const vilco_gemm_desc* d, void* stream) { return gemm_impl(d, stream, nullptr, nullptr); }

template <int BM>
static void launch_gl_group(const GArgsN& gg, int n, int ntiles, hipStream_t s, bool akm, bool bkm) {
  constexpr size_t lds = (size_t)4 * (BM + BN) * 128;
  const dim3 grid(ntiles, 1, n);
#define VILCO_GROUP_LAUNCH(A_, B_)                                                                                      \
  do {                                                                                                                  \
    static const bool once = [] {                                                                                       \
      hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_gl_group_kernel<BM, A_, B_>),                             \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                        \
      (void)hipGetLastError();                                                                                          \
      return true;                                                                                                      \
    }();                                                                                                                \
    (void)once;                                                                                                         \
    hipLaunchKernelGGL((gemm_gl_group_kernel<BM, A_, B_>), grid, dim3(512), lds, s, gg);                                \
  } while (0)
  if (akm && bkm) VILCO_GROUP_LAUNCH(true, true);
  else if (bkm) VILCO_GROUP_LAUNCH(false, true);
  else VILCO_GROUP_LAUNCH(false, false);
#undef VILCO_GROUP_LAUNCH
}

// n (2..4) independent products of ONE shape, orientation and format in one launch (gemm_gl_group_kernel): operands packed
// beforehand, precision 3, unbatched, untapped, no band; whatever does not qualify -- a split-K plan, the old kernel, unequal
// shapes -- runs as n ordinary vilco_gemm calls, in order (same results either way: the kernel body is the same).
extern "C" int vilco_gemm_group(const vilco_gemm_desc* descs, int32_t n, void* stream) {
  if (!descs || n < 1 || n > 4) return VILCO_ERR_BADARG;
  static const bool enabled = [] { const char* e = getenv("VILCO_GEMM_GROUP"); return !(e && e[0] == '0'); }();
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  GArgsN gg;
  Plan p0;
  bool ok = enabled && n >= 2;
  for (int i = 0; ok && i < n; ++i) {
    const vilco_gemm_desc& d = descs[i];
    Plan p;
    ok = d.precision == 3 && d.batch_outer == 1 && d.batch_inner == 1 && d.tap_operand == VILCO_TAP_NONE && d.band == 0 &&
         d.M == descs[0].M && d.N == descs[0].N && d.K == descs[0].K && d.a_kcontig == descs[0].a_kcontig &&
         d.b_kcontig == descs[0].b_kcontig && d.M > 0 && d.N > 0 && gemm_impl(&d, stream, &gg.g[i], &p) == VILCO_OK &&
         p.gl && p.ksplit == 1 && (i == 0 || (p.BM == p0.BM && p.a_km == p0.a_km && p.b_km == p0.b_km));
    if (i == 0) p0 = p;
  }
  if (!ok) {
    for (int i = 0; i < n; ++i) {
      const int rc = vilco_gemm(&descs[i], stream);
      if (rc != VILCO_OK) return rc;
    }
    return VILCO_OK;
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (prof().on) { hipEventCreate(&ev0); hipEventCreate(&ev1); hipEventRecord(ev0, s); }
  if (p0.BM == 192) launch_gl_group<192>(gg, n, gg.g[0].ntiles, s, p0.a_km, p0.b_km);
  else launch_gl_group<128>(gg, n, gg.g[0].ntiles, s, p0.a_km, p0.b_km);
  if (ev0) {
    hipEventRecord(ev1, s);
    prof().ev.emplace_back(ev0, ev1);
    prof().rec.push_back(ProfRec{{descs[0].M, descs[0].N, descs[0].K, n, p0.BM, 1, 3, p0.a_km, p0.b_km, 0x100}});    // (batch field = group size)
  }
  return vilco_launch_status();
}

// tuning override (tools/gemm_tune.py): force the tile height (128 | 192 | 256; 0 = cost model) and the split-K count
// (0 = heuristic) of every following vilco_gemm of this process; the environment variables VILCO_GEMM_BM / VILCO_GEMM_KS
// give the initial values.
// split-K finish: 1 = inside the launch (tile counters), 0 = partial slabs + splitk_reduce_kernel (default; see
// fixup_enabled()).  Both sum the splits in the same order: the results are bitwise equal (tests/test_ops_gpu.py).
// every setter below bumps this generation: a captured hipGraph replays the plan it recorded, so vilco_amd/graph.py keys its
// graphs on the value (a vilco_gemm_set_* call between replays used to replay the old kernels silently)
static int64_t g_config_gen = 0;
extern "C" int64_t vilco_gemm_config_gen(void) { return g_config_gen; }

extern "C" int vilco_gemm_set_fixup(int32_t on) {
  fixup_enabled() = on != 0;
  ++g_config_gen;
  return VILCO_OK;
}

extern "C" int vilco_gemm_set_gl(int32_t on) {
  gl_enabled() = on != 0;
  ++g_config_gen;
  return VILCO_OK;
}

extern "C" int vilco_gemm_set_skinny(int32_t on) {
  skinny_enabled() = on != 0;
  ++g_config_gen;
  return VILCO_OK;
}

extern "C" int vilco_gemm_set_tail128(int32_t on) {
  tail128_enabled() = on != 0;
  ++g_config_gen;
  return VILCO_OK;
}

extern "C" int vilco_gemm_force(int32_t bm, int32_t ks) {
  if (bm < 0 || ks < 0) return VILCO_ERR_BADARG;
  tune().bm = bm; tune().ks = ks;
  ++g_config_gen;
  return VILCO_OK;
}
