// prints the wave -> SIMD placement of one 512-thread workgroup (HW_REG_HW_ID bits [5:4] = simd_id on gfx9)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = v;
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 8 * 4);
  k<<<64, 512>>>(d);
  unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 6; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 8; ++w) printf(" w%d:simd%u/cu%u/wv%u", w, (h[b*8+w] >> 4) & 3, (h[b*8+w] >> 8) & 15, h[b*8+w] & 15);
    printf("\n");
  }
  return 0;
}
