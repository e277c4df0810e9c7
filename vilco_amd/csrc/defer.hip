// Deferred finishing of two-stage reductions (round 4).
//
// Every column sum of the backward pass (LayerNorm d-gamma / d-beta, bias gradients, AffineDropPath scale gradients, the
// depthwise-tap gradients) and every split-K weight-gradient product ends in a small second launch that folds per-block
// partials: ~350 launches per step of config P, 4-6 us each because each is a dependent node, not because of its bytes
// (profiles/r03_z_bench_kernel_stats.csv: splitk_reduce 148, reduce_rows 156, colsum 46).  Their results are PARAMETER
// gradients: nothing in backward reads them.  While vilco_defer_set(1) is in force those second
// stages are RECORDED instead of launched; vilco_defer_flush issues them as a handful of batched launches (the item table
// travels in the kernel arguments).  Each item is finished by the same arithmetic in the same order as its own launch
// would have used, so results are bitwise the same.  The caller keeps the partial buffers alive until the flush
// (vilco_amd/ops.py: _defer).
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct RRItem {           // out[j] = sum_r ws[r][j] (reduce_rows_kernel, norm.hip)
  const float* ws; float* out0; float* out1;
  int nrows, ncols, split, blk0;
};
constexpr int RR_MAX = 80;
struct RRBatch { RRItem it[RR_MAX]; int n; };

struct SKItem {           // out[m][n] = sum_s part[s][m][n] in split order (splitk_reduce_kernel, gemm.hip; plain epilogue)
  const float* part; float* out;
  long split_stride, ldc;
  int M, N, ksplit, blk0, nblk;
};
constexpr int SK_MAX = 64;
struct SKBatch { SKItem it[SK_MAX]; int n; };

// ONE recording per process, not per thread: autograd runs the backward functions of device tensors on its worker thread
// and the end-of-backward callback that flushes on the thread that called backward().  (Recording is a mode of the one
// training loop of the process; entry points used from other threads meanwhile would be recorded too.)
struct State {
  std::atomic<bool> on{false};      // written by the thread that drives backward, read by autograd's worker threads
  std::vector<RRItem> rr;
  std::vector<SKItem> sk;
  std::mutex mu;
};
State& st() { static State s; return s; }

template <class B>
__device__ __forceinline__ int find_item(const B& b, int blk) {
  int lo = 0, hi = b.n - 1;             // last item with blk0 <= blk
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (b.it[mid].blk0 <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void reduce_rows_many_kernel(RRBatch b) {
  __shared__ float part[4][64];
  const RRItem& it = b.it[find_item(b, (int)blockIdx.x)];
  const float* __restrict__ ws = it.ws;
  const int nrows = it.nrows, ncols = it.ncols;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int j = ((int)blockIdx.x - it.blk0) * 64 + lane;
  float s = 0.f;
  if (j < ncols) {
    int r = slice;
    for (; r + 28 < nrows; r += 32) {       // the loop of reduce_rows_kernel: the sum keeps the row order
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
    if (it.out1 && j >= it.split) it.out1[j - it.split] = s;
    else it.out0[j] = s;
  }
}

__global__ __launch_bounds__(256) void splitk_sum_many_kernel(SKBatch b) {
  const SKItem& it = b.it[find_item(b, (int)blockIdx.x)];
  const float* __restrict__ part = it.part;
  // 16-byte accesses need 16-byte aligned bases too: dW written into a GradReducer bucket slot sits at an arbitrary element
  // offset of the bucket (the undeferred path's vec_out checks the same, gemm.hip)
  const bool vec = (it.N % 4) == 0 && (it.ldc % 4) == 0 && (it.split_stride % 4) == 0 &&
                   ((reinterpret_cast<uintptr_t>(it.out) | reinterpret_cast<uintptr_t>(it.part)) & 15) == 0;
  const int V = vec ? 4 : 1;
  const int NV = it.N / V;
  const long total = (long)it.M * NV;
  const long stride = (long)it.nblk * 256;
  for (long i = (long)((int)blockIdx.x - it.blk0) * 256 + threadIdx.x; i < total; i += stride) {
    const int m = (int)(i / NV), n = (int)(i % NV) * V;
    const long idx = (long)m * it.ldc + n;
    if (vec) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      int k = 0;
      for (; k + 4 <= it.ksplit; k += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(part + (long)(k + u) * it.split_stride + idx);
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u];
      }
      for (; k < it.ksplit; ++k) s += *reinterpret_cast<const f32x4*>(part + (long)k * it.split_stride + idx);
      *reinterpret_cast<f32x4*>(it.out + idx) = s * 1.0f;      // (alpha = 1, as splitk_reduce_kernel's store_out4 applies it)
    } else {
      float s = 0.f;
      for (int k = 0; k < it.ksplit; ++k) s += part[(long)k * it.split_stride + idx];
      it.out[idx] = 1.0f * s;
    }
  }
}

}  // namespace

bool vilco_defer_active() { return st().on; }

void vilco_defer_push_rr(const float* ws, float* out0, float* out1, int nrows, int ncols, int split) {
  std::lock_guard<std::mutex> g(st().mu);
  st().rr.push_back(RRItem{ws, out0, out1, nrows, ncols, split, 0});
}

void vilco_defer_push_sk(const float* part, float* out, long split_stride, long ldc, int M, int N, int ksplit) {
  std::lock_guard<std::mutex> g(st().mu);
  st().sk.push_back(SKItem{part, out, split_stride, ldc, M, N, ksplit, 0, 0});
}

extern "C" int vilco_defer_set(int32_t on) {
  st().on = on != 0;
  return VILCO_OK;
}

extern "C" int64_t vilco_defer_pending(void) {
  std::lock_guard<std::mutex> g(st().mu);
  return (int64_t)(st().rr.size() + st().sk.size());
}

extern "C" int vilco_defer_flush(void* stream) {
  State& s = st();
  std::lock_guard<std::mutex> g(s.mu);
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
  for (size_t i0 = 0; i0 < s.rr.size(); i0 += RR_MAX) {
    RRBatch b;
    b.n = (int)((s.rr.size() - i0) < (size_t)RR_MAX ? s.rr.size() - i0 : RR_MAX);
    int blk = 0;
    for (int i = 0; i < b.n; ++i) {
      b.it[i] = s.rr[i0 + i];
      b.it[i].blk0 = blk;
      blk += (b.it[i].ncols + 63) / 64;
    }
    hipLaunchKernelGGL(reduce_rows_many_kernel, dim3(blk), dim3(256), 0, hs, b);
  }
  for (size_t i0 = 0; i0 < s.sk.size(); i0 += SK_MAX) {
    SKBatch b;
    b.n = (int)((s.sk.size() - i0) < (size_t)SK_MAX ? s.sk.size() - i0 : SK_MAX);
    int blk = 0;
    for (int i = 0; i < b.n; ++i) {
      b.it[i] = s.sk[i0 + i];
      long nb = ((long)b.it[i].M * b.it[i].N / 4 + 255) / 256;
      if (nb > 256) nb = 256;
      if (nb < 1) nb = 1;
      b.it[i].blk0 = blk;
      b.it[i].nblk = (int)nb;
      blk += (int)nb;
    }
    hipLaunchKernelGGL(splitk_sum_many_kernel, dim3(blk), dim3(256), 0, hs, b);
  }
  s.rr.clear();
  s.sk.clear();
  s.on = false;
  return vilco_launch_status();
}
