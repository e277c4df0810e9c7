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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
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
