// Candidate decode of PtTransformer.inference_single_video (MQ/libs/modeling/meta_archs.py:1594-1692;
// NLQ/libs/modeling/meta_archs.py:1253-1338) for one clip, all pyramid levels in one launch:
//   per level:  prob = sigmoid(logit) * valid;  keep prob > pre_nms_thresh;  the pre_nms_topk highest of them;
//               pt = idx / C, cls = idx % C;  seg = (t - off_l * stride, t + off_r * stride);  keep seg length > duration_thresh
// The reference does this with ~12 tensor ops and two boolean-index host round trips per level.  Here one workgroup per
// level thresholds, selects the exact top-k (radix select over the fp32 bit patterns: probabilities are positive, so
// their bits order like the values), decodes and compacts in index order; a second tiny launch concatenates the levels.
// Candidate ORDER inside a level differs from the reference (index order instead of score order): every consumer
// (batched_nms: per-class NMS, final top-k by score) orders by score itself, so only exact score ties could tell.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "common.h"
#include "../../include/vilco_hip.h"

namespace {

constexpr int DEC_THREADS = 1024;

struct DecArgs {
  const float* logits;       // [R][C]
  const float* offsets;      // [R][2]   relu(Scale_l(x)) of the regression head
  const float* points;       // [R][4] = (t, reg_lo, reg_hi, stride)
  const int* level_row0;     // [L] first row of level l in the row layout
  const int* level_len;      // [L] valid positions of level l (rows level_row0[l] .. + level_len[l] - 1)
  int C, L, topk;
  float thresh, dur_thresh;
  float* segs;               // [L][topk][2]
  float* scores;             // [L][topk]
  long long* labels;         // [L][topk]
  int* counts;               // [L]
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// block-wide exclusive scan of one int per thread (DEC_THREADS threads); returns the exclusive prefix, *total = sum
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  __syncthreads();
  if (lane == 63) sm[wave] = x;
  __syncthreads();
  if (threadIdx.x < 64) {
    int w = threadIdx.x < DEC_THREADS / 64 ? sm[threadIdx.x] : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(w, d, 64);
      if ((int)threadIdx.x >= d) w += y;
    }
    if (threadIdx.x < DEC_THREADS / 64) sm[threadIdx.x] = w;       // inclusive over waves
  }
  __syncthreads();
  const int base = wave ? sm[wave - 1] : 0;
  *total = sm[DEC_THREADS / 64 - 1];
  return base + x - v;
}

__global__ __launch_bounds__(DEC_THREADS) void decode_level_kernel(DecArgs a) {
  __shared__ int hist[256];
  __shared__ int sm[64];
  __shared__ unsigned s_prefix, s_need;
  const int l = blockIdx.x, tid = threadIdx.x;
  const int row0 = a.level_row0[l], len = a.level_len[l];
  const long n = (long)len * a.C;
  const float* lg = a.logits + (long)row0 * a.C;

  // ---- how many candidates pass the threshold
  int cnt = 0;
  for (long e = tid; e < n; e += DEC_THREADS) cnt += sigmoidf_(lg[e]) > a.thresh ? 1 : 0;
  int total;
  block_excl_scan(cnt, &total, sm);

  // ---- exact top-k by radix select when more pass than the cap: after the loop `s_prefix` is the bit pattern of the
  // k-th largest probability and `s_need` how many candidates EQUAL to it are still admitted (in index order)
  unsigned cut = 0u, need_eq = 0xffffffffu;      // default: admit everything above the threshold
  if (total > a.topk) {
    if (tid == 0) { s_prefix = 0u; s_need = (unsigned)a.topk; }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const unsigned prefix = s_prefix, mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
      for (long e = tid; e < n; e += DEC_THREADS) {
        const float p = sigmoidf_(lg[e]);
        if (!(p > a.thresh)) continue;
        const unsigned b = __float_as_uint(p);
        if ((b & mask) == (prefix & mask)) atomicAdd(&hist[(b >> shift) & 255u], 1);
      }
      __syncthreads();
      if (tid == 0) {                          // walk the 256 bins from the top: the bin holding the k-th largest
        unsigned need = s_need;
        int bin = 255;
        for (; bin > 0; --bin) {
          if ((unsigned)hist[bin] >= need) break;
          need -= (unsigned)hist[bin];
        }
        s_prefix = prefix | ((unsigned)bin << shift);
        s_need = need;
      }
      __syncthreads();
    }
    cut = s_prefix;
    need_eq = s_need;
  }

  // ---- decode + duration filter + compaction in index order.  Pass A marks, pass B writes (two scans: the equal-to-cut
  // candidates are rationed in index order, then the survivors are numbered)
  const long per = (n + DEC_THREADS - 1) / DEC_THREADS;          // a contiguous index range per thread: index order = thread order
  const long e0 = (long)tid * per, e1 = e0 + per < n ? e0 + per : n;
  int eq_before = 0;
  {
    int eq = 0;
    if (total > a.topk)
      for (long e = e0; e < e1; ++e) {
        const float p = sigmoidf_(lg[e]);
        if (p > a.thresh && __float_as_uint(p) == cut) ++eq;
      }
    int dummy;
    eq_before = block_excl_scan(eq, &dummy, sm);
  }
  auto admitted = [&](float p, int& eq_seen) {
    if (!(p > a.thresh)) return false;
    if (total <= a.topk) return true;
    const unsigned b = __float_as_uint(p);
    if (b > cut) return true;
    if (b < cut) return false;
    return (unsigned)(eq_seen++) < need_eq;
  };
  auto decode = [&](long e, float& left, float& right) {
    const long r = row0 + e / a.C;
    const float t = a.points[r * 4], stride = a.points[r * 4 + 3];
    left = t - a.offsets[r * 2] * stride;
    right = t + a.offsets[r * 2 + 1] * stride;
    return (right - left) > a.dur_thresh;
  };
  int keep = 0;
  {
    int eq_seen = eq_before;
    for (long e = e0; e < e1; ++e) {
      float lft, rgt;
      if (admitted(sigmoidf_(lg[e]), eq_seen) && decode(e, lft, rgt)) ++keep;
    }
  }
  int kept_total;
  int pos = block_excl_scan(keep, &kept_total, sm);
  {
    int eq_seen = eq_before;
    for (long e = e0; e < e1; ++e) {
      const float p = sigmoidf_(lg[e]);
      float lft, rgt;
      if (admitted(p, eq_seen) && decode(e, lft, rgt)) {
        const long o = (long)l * a.topk + pos++;
        a.segs[o * 2] = lft; a.segs[o * 2 + 1] = rgt;
        a.scores[o] = p;
        a.labels[o] = (long long)(e % a.C);
      }
    }
  }
  if (tid == 0) a.counts[l] = kept_total;
}

// concatenate the per-level slabs: out rows [sum_{l' < l} counts[l'] ...); total -> out_total[0]
__global__ __launch_bounds__(256) void decode_concat_kernel(DecArgs a, float* osegs, float* oscores, long long* olabels, int* out_total) {
  __shared__ int base[65];
  if (threadIdx.x == 0) {
    int s = 0;
    for (int l = 0; l < a.L; ++l) { base[l] = s; s += a.counts[l]; }
    base[a.L] = s;
    if (blockIdx.x == 0) out_total[0] = s;
  }
  __syncthreads();
  const int l = blockIdx.y, c = a.counts[l];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c; i += gridDim.x * blockDim.x) {
    const long src = (long)l * a.topk + i, dst = base[l] + i;
    osegs[dst * 2] = a.segs[src * 2]; osegs[dst * 2 + 1] = a.segs[src * 2 + 1];
    oscores[dst] = a.scores[src];
    olabels[dst] = a.labels[src];
  }
}

inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

extern "C" size_t vilco_decode_workspace(int32_t L, int32_t topk) {
  if (L <= 0 || L > 64 || topk <= 0) return 0;
  const size_t n = (size_t)L * topk;
  return up256(n * 8) + up256(n * 4) + up256(n * 8) + up256((size_t)L * 4) + 256;
}

extern "C" int vilco_decode(const float* logits, const float* offsets, const float* points, const int32_t* level_row0,
                            const int32_t* level_len, int32_t C, int32_t L, int32_t topk, float pre_nms_thresh,
                            float duration_thresh, float* out_segs, float* out_scores, int64_t* out_labels,
                            int32_t* out_total, void* workspace, size_t workspace_bytes, void* stream) {
  if (!logits || !offsets || !points || !level_row0 || !level_len || !out_segs || !out_scores || !out_labels || !out_total)
    return VILCO_ERR_BADARG;
  if (C <= 0 || L <= 0 || L > 64 || topk <= 0) return VILCO_ERR_BADARG;
  if (!workspace || workspace_bytes < vilco_decode_workspace(L, topk)) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t n = (size_t)L * topk;
  unsigned char* w = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(workspace) + 255) / 256 * 256);
  DecArgs a;
  a.logits = logits; a.offsets = offsets; a.points = points; a.level_row0 = level_row0; a.level_len = level_len;
  a.C = C; a.L = L; a.topk = topk; a.thresh = pre_nms_thresh; a.dur_thresh = duration_thresh;
  a.segs = reinterpret_cast<float*>(w); w += up256(n * 8);
  a.scores = reinterpret_cast<float*>(w); w += up256(n * 4);
  a.labels = reinterpret_cast<long long*>(w); w += up256(n * 8);
  a.counts = reinterpret_cast<int*>(w);
  hipLaunchKernelGGL(decode_level_kernel, dim3(L), dim3(DEC_THREADS), 0, s, a);
  hipLaunchKernelGGL(decode_concat_kernel, dim3(8, L), dim3(256), 0, s, a, out_segs, out_scores,
                     reinterpret_cast<long long*>(out_labels), out_total);
  return vilco_launch_status();
}
