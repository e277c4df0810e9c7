#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float4* src, float4* out) {
  __shared__ float4 buf[256];
  const int tid = threadIdx.x;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + tid),
                                   (__attribute__((address_space(3))) void*)(buf + (tid & ~63)), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[tid] = buf[tid ^ 1];
}
int main() {
  float4 h[256], o[256]; for (int i = 0; i < 256; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
  float4 *d, *e; hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 256>>>(d, e); hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) if (o[i].x != h[i ^ 1].x || o[i].w != h[i ^ 1].w) ++bad;
  printf("bad %d  o[0]=%g,%g o[65]=%g\n", bad, o[0].x, o[0].w, o[65].x);
  return 0;
}
