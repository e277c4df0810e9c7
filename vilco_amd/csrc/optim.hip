// Step glue of the training loop (MQ/libs/utils/train_utils.py:343-351): global-L2-norm gradient clipping
// (torch.nn.utils.clip_grad_norm_) and the AdamW / SGD-momentum update of torch.optim, as multi-tensor kernels:
// ONE launch walks every parameter tensor through a chunk table, so the 358 gradient-bearing tensors of config P
// (210.7 M parameters, 28 B/param of HBM traffic for AdamW) cost 3 launches instead of ~1500.
// The clip coefficient stays on the device (no host sync) and is applied to the gradient inside the update.
#include "common.h"

namespace {

constexpr int OPT_THREADS = 256;

struct MultiArgs {
  const long* ptrs;          // [4][n]: param, grad, exp_avg (m), exp_avg_sq (v) device pointers
  const long* numel;         // [n]
  const int* chunk_tensor;   // [nchunks]
  const long* chunk_off;     // [nchunks] element offset inside the tensor
  const int* group;          // [n] parameter-group index
  int n, chunk;
};

__global__ __launch_bounds__(OPT_THREADS) void sqnorm_kernel(MultiArgs a, float* __restrict__ partial) {
  __shared__ float red[OPT_THREADS / 64];
  const int t = a.chunk_tensor[blockIdx.x];
  const long off = a.chunk_off[blockIdx.x];
  const float* g = reinterpret_cast<const float*>(a.ptrs[(long)a.n + t]) + off;
  long cnt = a.numel[t] - off;
  if (cnt > a.chunk) cnt = a.chunk;
  float s = 0.f;
  for (long i = threadIdx.x; i < cnt; i += OPT_THREADS) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))   (max_norm <= 0: coef 1)
__global__ __launch_bounds__(OPT_THREADS) void clip_coef_kernel(const float* __restrict__ partial, int n,
                                                                float max_norm, float* __restrict__ out) {
  __shared__ double red[OPT_THREADS];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += OPT_THREADS) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = OPT_THREADS / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    out[0] = norm;
    float c = 1.f;
    if (max_norm > 0.f) {
      c = max_norm / (norm + 1e-6f);
      if (c > 1.f) c = 1.f;
    }
    out[1] = c;
  }
}

struct Hyper {
  float lr[8], wd[8];
  float beta1, beta2, eps;
  float momentum;                          // SGD
  const float* tstep;                      // [n] per-tensor step count (after this update), torch keeps it per param
  const float* lr_dev;                     // optional [ngroups] learning rates in device memory (override lr[]): a captured
                                           // step replays its kernel arguments, the schedule's value must come from memory
};

// torch.optim.AdamW (decoupled weight decay): p *= 1 - lr*wd ; m,v EMA ; p -= lr/bias1 * m / (sqrt(v)/sqrt(bias2) + eps)
// chunk_amax (optional): max|p| of the UPDATED parameter over each chunk -- the optimizer is the producer of next step's
// weights, so the fp16 x2 weight packs take their per-tensor scale from these partials instead of re-reading the weight
__device__ __forceinline__ void emit_chunk_amax(float mx, float* __restrict__ chunk_amax) {
  __shared__ float red[OPT_THREADS / 64];
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) chunk_amax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(MultiArgs a, Hyper h, const float* __restrict__ coef,
                                                            float* __restrict__ chunk_amax) {
  const int t = a.chunk_tensor[blockIdx.x];
  const long off = a.chunk_off[blockIdx.x];
  float* p = reinterpret_cast<float*>(a.ptrs[t]) + off;
  const float* g = reinterpret_cast<const float*>(a.ptrs[(long)a.n + t]) + off;
  float* m = reinterpret_cast<float*>(a.ptrs[2L * a.n + t]) + off;
  float* v = reinterpret_cast<float*>(a.ptrs[3L * a.n + t]) + off;
  long cnt = a.numel[t] - off;
  if (cnt > a.chunk) cnt = a.chunk;
  const int gi = a.group[t];
  const float lr = h.lr_dev ? h.lr_dev[gi] : h.lr[gi], wd = h.wd[gi];
  const float c = coef ? coef[1] : 1.f;
  const float stp = h.tstep[t];
  const float step_size = lr / (1.f - powf(h.beta1, stp));      // bias corrections 1 - beta^step
  const float rs2 = sqrtf(1.f - powf(h.beta2, stp));
  float mx = 0.f;
  for (long i = threadIdx.x; i < cnt; i += OPT_THREADS) {
    const float gr = g[i] * c;
    float pv = p[i] * (1.f - lr * wd);
    const float mv = h.beta1 * m[i] + (1.f - h.beta1) * gr;
    const float vv = h.beta2 * v[i] + (1.f - h.beta2) * gr * gr;
    m[i] = mv;
    v[i] = vv;
    pv -= step_size * (mv / (sqrtf(vv) / rs2 + h.eps));
    p[i] = pv;
    mx = fmaxf(mx, fabsf(pv));
  }
  if (chunk_amax) emit_chunk_amax(mx, chunk_amax);
}

// torch.optim.SGD with momentum (no dampening / nesterov): g += wd*p ; buf = first ? g : mom*buf + g ; p -= lr*buf
__global__ __launch_bounds__(OPT_THREADS) void sgd_kernel(MultiArgs a, Hyper h, const float* __restrict__ coef,
                                                          float* __restrict__ chunk_amax) {
  const int t = a.chunk_tensor[blockIdx.x];
  const long off = a.chunk_off[blockIdx.x];
  float* p = reinterpret_cast<float*>(a.ptrs[t]) + off;
  const float* g = reinterpret_cast<const float*>(a.ptrs[(long)a.n + t]) + off;
  float* m = reinterpret_cast<float*>(a.ptrs[2L * a.n + t]) + off;
  long cnt = a.numel[t] - off;
  if (cnt > a.chunk) cnt = a.chunk;
  const int gi = a.group[t];
  const float lr = h.lr_dev ? h.lr_dev[gi] : h.lr[gi], wd = h.wd[gi];
  const float c = coef ? coef[1] : 1.f;
  const bool first = h.tstep[t] <= 1.f;          // momentum buffer starts as the first gradient
  float mx = 0.f;
  for (long i = threadIdx.x; i < cnt; i += OPT_THREADS) {
    const float pv0 = p[i];
    float gr = g[i] * c + wd * pv0;
    const float b = first ? gr : h.momentum * m[i] + gr;
    m[i] = b;
    const float pv = pv0 - lr * b;
    p[i] = pv;
    mx = fmaxf(mx, fabsf(pv));
  }
  if (chunk_amax) emit_chunk_amax(mx, chunk_amax);
}

// EWC / MAS penalty (MQ/libs/cl_methods/EWC.py:6-22, MAS.py:5-21) over a chunk table of (parameter, importance,
// consolidated value) triples: partial[chunk] = sum F (opt - p)^2 ; grad += -2 lambda F (opt - p).
// ptrs = [4][n]: param, grad, importance, optpar; numel[t] = optpar's element count (a prefix of the parameter when the
// class head has grown since the task was consolidated).
__global__ __launch_bounds__(OPT_THREADS) void cl_penalty_kernel(MultiArgs a, float lambda, float* __restrict__ partial) {
  __shared__ float red[OPT_THREADS / 64];
  const int t = a.chunk_tensor[blockIdx.x];
  const long off = a.chunk_off[blockIdx.x];
  const float* p = reinterpret_cast<const float*>(a.ptrs[t]) + off;
  float* g = reinterpret_cast<float*>(a.ptrs[(long)a.n + t]) + off;
  const float* f = reinterpret_cast<const float*>(a.ptrs[2L * a.n + t]) + off;
  const float* o = reinterpret_cast<const float*>(a.ptrs[3L * a.n + t]) + off;
  long cnt = a.numel[t] - off;
  if (cnt > a.chunk) cnt = a.chunk;
  float s = 0.f;
  for (long i = threadIdx.x; i < cnt; i += OPT_THREADS) {
    const float d = o[i] - p[i];
    const float fd = f[i] * d;
    s += fd * d;
    g[i] -= 2.f * lambda * fd;      // every parameter occurs once: plain read-modify-write
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// the same, for a parameter that appears in more than one task's dictionary: gradient accumulated with atomics
__global__ __launch_bounds__(OPT_THREADS) void cl_penalty_atomic_kernel(MultiArgs a, float lambda,
                                                                        float* __restrict__ partial) {
  __shared__ float red[OPT_THREADS / 64];
  const int t = a.chunk_tensor[blockIdx.x];
  const long off = a.chunk_off[blockIdx.x];
  const float* p = reinterpret_cast<const float*>(a.ptrs[t]) + off;
  float* g = reinterpret_cast<float*>(a.ptrs[(long)a.n + t]) + off;
  const float* f = reinterpret_cast<const float*>(a.ptrs[2L * a.n + t]) + off;
  const float* o = reinterpret_cast<const float*>(a.ptrs[3L * a.n + t]) + off;
  long cnt = a.numel[t] - off;
  if (cnt > a.chunk) cnt = a.chunk;
  float s = 0.f;
  for (long i = threadIdx.x; i < cnt; i += OPT_THREADS) {
    const float d = o[i] - p[i];
    const float fd = f[i] * d;
    s += fd * d;
    atomicAdd(&g[i], -2.f * lambda * fd);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(OPT_THREADS) void scaled_sum_kernel(const float* __restrict__ partial, int n, float scale,
                                                                 float* __restrict__ out) {
  __shared__ double red[OPT_THREADS];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += OPT_THREADS) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = OPT_THREADS / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] * (double)scale);
}

struct F16Vals { float v[16]; };
__global__ void store_f32_kernel(float* __restrict__ dst, F16Vals vals, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}

}  // namespace

// dst[0..n) = vals[0..n) (n <= 16 host floats, carried as kernel arguments: no staging buffer whose lifetime the caller
// would have to manage, stream-ordered with the replays that read dst).  The learning rates of a captured step.
extern "C" int vilco_store_f32(float* dst, const float* vals, int32_t n, void* stream) {
  if (!dst || !vals || n < 0 || n > 16) return VILCO_ERR_BADARG;
  if (n == 0) return VILCO_OK;
  F16Vals v;
  for (int i = 0; i < 16; ++i) v.v[i] = i < n ? vals[i] : 0.f;
  hipLaunchKernelGGL(store_f32_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), dst, v, n);
  return vilco_launch_status();
}

extern "C" int vilco_cl_penalty(const int64_t* ptrs, const int64_t* numel, const int32_t* chunk_tensor,
                                const int64_t* chunk_off, int32_t n, int32_t nchunks, int32_t chunk, float lambda,
                                int32_t shared_params, float* partial, float* out, void* stream) {
  if (!ptrs || !numel || !chunk_tensor || !chunk_off || !partial || !out || n < 0 || nchunks < 0 || chunk <= 0)
    return VILCO_ERR_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  MultiArgs a{reinterpret_cast<const long*>(ptrs), reinterpret_cast<const long*>(numel), chunk_tensor,
              reinterpret_cast<const long*>(chunk_off), nullptr, n, chunk};
  if (nchunks > 0) {
    if (shared_params)
      hipLaunchKernelGGL(cl_penalty_atomic_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, s, a, lambda, partial);
    else
      hipLaunchKernelGGL(cl_penalty_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, s, a, lambda, partial);
  }
  hipLaunchKernelGGL(scaled_sum_kernel, dim3(1), dim3(OPT_THREADS), 0, s, partial, nchunks, lambda, out);
  return vilco_launch_status();
}

extern "C" int vilco_grad_norm(const int64_t* ptrs, const int64_t* numel, const int32_t* chunk_tensor,
                               const int64_t* chunk_off, int32_t n, int32_t nchunks, int32_t chunk, float max_norm,
                               float* partial, float* norm_coef, void* stream) {
  if (!ptrs || !numel || !chunk_tensor || !chunk_off || !partial || !norm_coef || n < 0 || nchunks < 0 || chunk <= 0)
    return VILCO_ERR_BADARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  MultiArgs a{reinterpret_cast<const long*>(ptrs), reinterpret_cast<const long*>(numel), chunk_tensor,
              reinterpret_cast<const long*>(chunk_off), nullptr, n, chunk};
  if (nchunks > 0) hipLaunchKernelGGL(sqnorm_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, s, a, partial);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(OPT_THREADS), 0, s, partial, nchunks, max_norm, norm_coef);
  return vilco_launch_status();
}

extern "C" int vilco_optim_step_dev(int32_t kind, const int64_t* ptrs, const int64_t* numel, const int32_t* chunk_tensor,
                                    const int64_t* chunk_off, const int32_t* group, int32_t n, int32_t nchunks,
                                    int32_t chunk, const float* lr, const float* wd, int32_t ngroups, float beta1,
                                    float beta2, float eps, float momentum, const float* tensor_step,
                                    const float* norm_coef, float* chunk_amax, const float* lr_dev, void* stream) {
  if (!ptrs || !numel || !chunk_tensor || !chunk_off || !group || !lr || !wd || n < 0 || nchunks < 0 || chunk <= 0)
    return VILCO_ERR_BADARG;
  if (ngroups < 1 || ngroups > 8 || !tensor_step || (kind != 0 && kind != 1)) return VILCO_ERR_BADARG;
  if (nchunks == 0) return VILCO_OK;
  Hyper h;
  for (int i = 0; i < 8; ++i) { h.lr[i] = i < ngroups ? lr[i] : 0.f; h.wd[i] = i < ngroups ? wd[i] : 0.f; }
  h.beta1 = beta1; h.beta2 = beta2; h.eps = eps;
  h.momentum = momentum;
  h.tstep = tensor_step;
  h.lr_dev = lr_dev;
  MultiArgs a{reinterpret_cast<const long*>(ptrs), reinterpret_cast<const long*>(numel), chunk_tensor,
              reinterpret_cast<const long*>(chunk_off), group, n, chunk};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (kind == 0) hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, s, a, h, norm_coef, chunk_amax);
  else hipLaunchKernelGGL(sgd_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, s, a, h, norm_coef, chunk_amax);
  return vilco_launch_status();
}

extern "C" int vilco_optim_step_amax(int32_t kind, const int64_t* ptrs, const int64_t* numel, const int32_t* chunk_tensor,
                                     const int64_t* chunk_off, const int32_t* group, int32_t n, int32_t nchunks,
                                     int32_t chunk, const float* lr, const float* wd, int32_t ngroups, float beta1,
                                     float beta2, float eps, float momentum, const float* tensor_step,
                                     const float* norm_coef, float* chunk_amax, void* stream) {
  return vilco_optim_step_dev(kind, ptrs, numel, chunk_tensor, chunk_off, group, n, nchunks, chunk, lr, wd, ngroups, beta1,
                              beta2, eps, momentum, tensor_step, norm_coef, chunk_amax, nullptr, stream);
}

extern "C" int vilco_optim_step(int32_t kind, const int64_t* ptrs, const int64_t* numel, const int32_t* chunk_tensor,
                                const int64_t* chunk_off, const int32_t* group, int32_t n, int32_t nchunks,
                                int32_t chunk, const float* lr, const float* wd, int32_t ngroups, float beta1,
                                float beta2, float eps, float momentum, const float* tensor_step,
                                const float* norm_coef, void* stream) {
  return vilco_optim_step_amax(kind, ptrs, numel, chunk_tensor, chunk_off, group, n, nchunks, chunk, lr, wd, ngroups, beta1,
                               beta2, eps, momentum, tensor_step, norm_coef, nullptr, stream);
}
