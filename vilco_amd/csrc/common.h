// Shared device/host helpers for libvilco_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vilco_hip.h"

#define VILCO_WAVE 64

static inline int vilco_launch_status() {
  return hipGetLastError() == hipSuccess ? VILCO_OK : VILCO_ERR_LAUNCH;
}

static inline bool vilco_aligned(const void* p, size_t a) {
  return (reinterpret_cast<uintptr_t>(p) % a) == 0;
}

// Wavefront reductions, result in every lane.  Inside a row of 16 lanes the partners come through DPP operand modifiers
// (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror: one VALU instruction per step, no LDS); the four row totals are
// read out with v_readlane and added as scalars operands.  11 instructions and no s_waitcnt, against 6 x (ds_bpermute +
// wait + add) for the __shfl_xor butterfly -- the kernels whose waves reduce many small statistics (LayerNorm, the fused
// q/k/v pre-projection, the amax partials) were issue- and LDS-latency-bound on those.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float vilco_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float vilco_lane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// op(x, x from lane ^ 16) and op(x, x from lane ^ 32) for commutative ops, through gfx950's v_permlane{16,32}_swap
// (VALU, no LDS round trip as with ds_bpermute): swapping the odd rows / upper half of one copy of x with the even rows /
// lower half of another leaves the two partners of every lane in the two results.
__device__ __forceinline__ void vilco_pair16(float x, float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void vilco_pair32(float x, float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
// reductions over the four lanes {l, l^16, l^32, l^48} (the MFMA accumulator layout spreads a row over them)
__device__ __forceinline__ float quad16_sum(float x) {
  float a, b;
  vilco_pair16(x, a, b); x = a + b;
  vilco_pair32(x, a, b); return a + b;
}
__device__ __forceinline__ float quad16_max(float x) {
  float a, b;
  vilco_pair16(x, a, b); x = fmaxf(a, b);
  vilco_pair32(x, a, b); return fmaxf(a, b);
}

__device__ __forceinline__ float wave_sum(float v) {
  v += vilco_dpp<0xB1>(v);          // quad_perm [1,0,3,2]
  v += vilco_dpp<0x4E>(v);          // quad_perm [2,3,0,1]
  v += vilco_dpp<0x141>(v);         // row_half_mirror: lanes i <-> 7 - i of every 8
  v += vilco_dpp<0x140>(v);         // row_mirror:      lanes i <-> 15 - i of every 16
  return (vilco_lane(v, 0) + vilco_lane(v, 16)) + (vilco_lane(v, 32) + vilco_lane(v, 48));
}

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, vilco_dpp<0xB1>(v));
  v = fmaxf(v, vilco_dpp<0x4E>(v));
  v = fmaxf(v, vilco_dpp<0x141>(v));
  v = fmaxf(v, vilco_dpp<0x140>(v));
  return fmaxf(fmaxf(vilco_lane(v, 0), vilco_lane(v, 16)), fmaxf(vilco_lane(v, 32), vilco_lane(v, 48)));
}

// Dropout keep decision of element `idx` of the stream `seed`: counter-based (no state), so forward and backward -- and,
// for attention probabilities, three different kernels -- regenerate the same mask.  32-bit murmur3 finalizer over the
// folded 64-bit index; keep iff hash >= p * 2^32.
__device__ __forceinline__ uint32_t vilco_drop_hash(uint32_t seed, uint64_t idx) {
  uint32_t h = (uint32_t)idx ^ (seed * 0x9E3779B1u) ^ ((uint32_t)(idx >> 32) * 0x85EBCA77u);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  h += seed; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
  return h;
}
// Attention-probability dropout (the [B*H, Tq, Tk] masks of vilco_attn_*; round 5).  One strong hash per ROW of the probability
// matrix (vilco_drop_hash over the row index bh * Tq + i), then a murmur3-style finalizer over row key + j * golden ratio per
// element: 2 of the quarter-rate 32-bit multiplies per element instead of 4 -- the full hash per element was 28 % of the XLNet
// forward kernel's cycles (tools/lab/attn_stamps_xl.py).  The kernels keep a lane's row keys in registers (forward / dQ: two
// rows for the whole kernel) or stage them through LDS with the Q tile (dK-dV).  Same mask in all three kernels and in
// vilco_attn_dropout_mask; adjacent / strided / cross-row / cross-seed correlations of the drop indicator are at the noise
// level of 2e7 samples and row / column drop rates have binomial spread (checked on the host with numpy when it was chosen).
__device__ __forceinline__ uint32_t vilco_attn_drop_row(uint32_t seed, uint64_t row) { return vilco_drop_hash(seed, row); }
constexpr uint32_t VILCO_ATTN_DROP_W = 0x9E3779B1u;
// `jw` = j * VILCO_ATTN_DROP_W (callers form it from a per-tile base plus compile-time constants)
__device__ __forceinline__ bool vilco_attn_drop_keep_w(uint32_t rowkey, uint32_t jw, uint32_t thresh) {
  uint32_t h = rowkey + jw;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u;
  return h >= thresh;
}
__device__ __forceinline__ bool vilco_attn_drop_keep(uint32_t rowkey, uint32_t j, uint32_t thresh) {
  return vilco_attn_drop_keep_w(rowkey, j * VILCO_ATTN_DROP_W, thresh);
}
// The seed a kernel actually uses: the launch's seed plus the device-resident step word (sync.hip).  A step captured as a
// hipGraph replays its kernel arguments, so what must change from replay to replay lives in memory: the graph's first node
// bumps the word (vilco_seed_word_bump) and every mask of the replay moves with it.  Eager launches run with word 0,
// where the effective seed IS the launch's seed.
__device__ __forceinline__ uint32_t vilco_step_seed(uint32_t seed, const uint32_t* word) {
  return seed + __builtin_nontemporal_load(word) * 0x9E3779B1u;
}
const uint32_t* vilco_seed_word_dev();     // sync.hip: device address of the step word

static inline uint32_t vilco_drop_threshold_host(float p) {     // host side: the kernels receive the threshold
  if (p <= 0.f) return 0u;
  const double t = (double)p * 4294967296.0;
  return t > 4294967040.0 ? 4294967040u : (uint32_t)t;
}

// erf-GELU, as torch.nn.GELU() default (blocks.py:480, XLNet "gelu").
__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// ------------------------------------------------------------------------------------------------------------------
// In-launch grid barrier + "finish the column reduction in the same launch".
// Two-stage reductions (per-block partial sums -> final sums; per-block partial maxima -> scale -> pack) used to be two
// launches each; ~5 us of launch latency plus ~1.5 us of dependency gap per extra kernel made them cost more than the
// work.  Here every block arrives at a counter, the last one flips a generation word, everybody leaves: ~2-3 us.
// Requirements, enforced by the host code that sizes the grids: all blocks of the launch are co-resident
// (<= VILCO_SYNC_MAX_BLOCKS blocks of <= 256 threads: 4 per CU), and one launch at a time per counter pair -- counters
// are per (stream slot, call site), streams get their slot from vilco_sync_slot() (sync.hip).
// The spin is bounded: a barrier that does not complete counts a timeout (vilco_sync_timeouts_read) instead of hanging.
constexpr int VILCO_SYNC_MAX_BLOCKS = 1024;
constexpr int VILCO_SYNC_SITES = 16;        // barrier states per stream slot
constexpr int VILCO_SYNC_SLOTS = 8;
constexpr int VILCO_SYNC_GROUPS = 32;       // arrival groups (one 64-byte line each) below the top-level counter
constexpr int VILCO_SYNC_WORDS = 16 * (1 + VILCO_SYNC_GROUPS);   // words per barrier state
enum { VILCO_SITE_PACK = 0, VILCO_SITE_LN = 1, VILCO_SITE_COLSUM = 2, VILCO_SITE_DWCONV = 3, VILCO_SITE_GEMMPACK = 4,
       VILCO_SITE_ATTNPACK = 5 };

// device address of the site's barrier state (sync.hip), or null: use the two-launch form
unsigned* vilco_sync_counter(hipStream_t s, int site);

// Device-scope atomics on MI355X execute at the memory side (the per-XCD L2s are not coherent with each other), ~40 ns
// apiece on one address: 1024 blocks arriving at ONE counter cost ~40 us (measured, r02).  So arrival is two-level:
// block b arrives at group b % G (G lines in different channels, in parallel), the last arriver of a group arrives at the
// top line, the last of those flips every group's generation word; blocks poll only their own group's line.
// state layout (words): line 0 = {top arrive, -, timeouts}; line 1 + g = {arrive, generation}.
//
// Memory: an agent-scope release / acquire fence on this multi-XCD part writes back / invalidates the XCD's whole L2
// (measured: ~20 us per barrier after a kernel has dirtied megabytes).  The barrier therefore carries NO fence: the few
// words that cross it (partial sums, partial maxima) must be written with vilco_st_agent and read with vilco_ld_agent
// (write-through / cache-bypassing accesses at device scope); each thread waits for its own stores to be acknowledged
// before the block arrives.
__device__ __forceinline__ void vilco_st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float vilco_ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void vilco_grid_barrier(unsigned* ctr, unsigned nblocks, unsigned bid) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // my write-through stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    const unsigned G = nblocks < (unsigned)VILCO_SYNC_GROUPS ? nblocks : (unsigned)VILCO_SYNC_GROUPS;
    const unsigned grp = bid % G;
    const unsigned gsize = (nblocks - grp + G - 1) / G;
    unsigned* g = ctr + 16 * (1 + grp);
    const unsigned gen = __hip_atomic_load(&g[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // before arriving
    if (__hip_atomic_fetch_add(&g[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
      __hip_atomic_store(&g[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(&ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1) {
        __hip_atomic_store(&ctr[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the counter resets land before anybody is released
        for (unsigned j = 0; j < G; ++j)
          __hip_atomic_fetch_add(&ctr[16 * (1 + j) + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    long spins = 0;
    while (__hip_atomic_load(&g[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1L << 21)) {            // ~seconds: co-residency assumption violated
        atomicAdd(&ctr[2], 1u);
        break;
      }
    }
  }
  __syncthreads();
}

// out[j] = sum_r ws[r][j] (j < ncols; j >= split goes to out1[j - split] when out1 != null), by ALL blocks of the launch
// after a grid barrier: block `bid` of `nblocks` takes 64-column groups bid, bid + nblocks, ...; 4 row slices per group
// combined through LDS.  blockDim must be 256 (1-D).
__device__ __forceinline__ void vilco_finish_colsum(const float* __restrict__ ws, float* __restrict__ out0,
                                                    float* __restrict__ out1, int nrows, int ncols, int split,
                                                    unsigned* ctr, unsigned bid, unsigned nblocks) {
  __shared__ float vilco_fc_part[4][64];
  vilco_grid_barrier(ctr, nblocks, bid);
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  for (int g = (int)bid; g * 64 < ncols; g += (int)nblocks) {
    const int j = g * 64 + lane;
    float s = 0.f;
    if (j < ncols)
      for (int r = slice; r < nrows; r += 4) s += vilco_ld_agent(ws + (long)r * ncols + j);
    __syncthreads();
    vilco_fc_part[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && j < ncols) {
      s = (vilco_fc_part[0][lane] + vilco_fc_part[1][lane]) + (vilco_fc_part[2][lane] + vilco_fc_part[3][lane]);
      if (out1 && j >= split) out1[j - split] = s;
      else out0[j] = s;
    }
  }
}

// deferred second stages of two-stage reductions (defer.hip)
bool vilco_defer_active();

// vilco_pack buffers (pack.h, gemm.hip): [VILCO_AMAX_MAX_BLOCKS floats of amax partials | {1/s, s} | pad] = VILCO_PACK_HDR bytes,
// then the 16-bit planes [part][rows32][cols32].  Producers that write the planes of their own output (eltwise.hip, norm.hip)
// use the same layout.
constexpr int VILCO_AMAX_MAX_BLOCKS = 1024;
constexpr long VILCO_PACK_HDR = VILCO_AMAX_MAX_BLOCKS * 4 + 512;
// rows of a zero-padded per-sequence plane image (vilco_pack_item.seq_len): nseq * (T + 2) padded rows + enough zero rows for
// both readers -- the forward / dX conv's overlapped spans and the weight-gradient product's contraction over the padded rows
static inline long vilco_tap_plane_rows(long nseq, long T) { return (nseq * (T + 2) + 31) / 32 * 32 + 64; }
void vilco_defer_push_rr(const float* ws, float* out0, float* out1, int nrows, int ncols, int split);
void vilco_defer_push_sk(const float* part, float* out, long split_stride, long ldc, int M, int N, int ksplit);
