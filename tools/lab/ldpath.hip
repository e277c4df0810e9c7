// Lab (round 5): what does the per-CU vector-memory path deliver for GEMM-staging-shaped loads out of L2?
// One workgroup per CU streams "tile rows" of an L2-resident matrix [R][ROWB bytes] the way a K loop does: per step every
// wave issues 1-KiB load instructions that cover  SEG bytes of 1024/SEG consecutive rows (SEG = 64: the BK = 32 planes of
// rounds 1-4, half a 128-B line per row;  SEG = 128: whole lines;  SEG = 1024: one contiguous KiB), then moves SEG bytes
// along the rows.  DST = 0: buffer_load_dwordx4 to VGPRs, DST = 1: global_load_lds_dwordx4 (LDS-DMA).
// Prints bytes / cycle / CU (s_memtime) and GB/s per CU from the wall clock.   hipcc --offload-arch=gfx950 -O3 ldpath.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SEG, int DST, int NW, int U>
__global__ __launch_bounds__(NW * 64) void k(const unsigned char* src, long rowb, int rows_total, int steps,
                                             unsigned* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int RPI = 1024 / SEG;              // rows per instruction
  constexpr int LPR = SEG / 16;                // lanes per row
  constexpr int ROWS = NW * U * RPI;           // rows this workgroup touches per step
  const int r_in = lane / LPR, c_in = (lane % LPR) * 16;
  const int row0 = (int)(((long)blockIdx.x * 977) % (rows_total - ROWS));
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, -1, 0x00020000);
  unsigned voff[U];
#pragma unroll
  for (int j = 0; j < U; ++j) voff[j] = (unsigned)((long)(row0 + (j * NW + wave) * RPI + r_in) * rowb + c_in);
  u32x4 acc = {0, 0, 0, 0};
  u32x4 va[U], vb[U];
  const int kmask = (int)rowb - 1;             // rowb is a power of two
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (DST == 0) {
#pragma unroll
    for (int j = 0; j < U; ++j) va[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[j], 0, 0);
    for (int s = 1; s + 1 < steps; s += 2) {
      const int k1 = (s * SEG) & kmask, k2 = ((s + 1) * SEG) & kmask;
#pragma unroll
      for (int j = 0; j < U; ++j) vb[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[j], k1, 0);
#pragma unroll
      for (int j = 0; j < U; ++j) acc ^= va[j];
#pragma unroll
      for (int j = 0; j < U; ++j) va[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[j], k2, 0);
#pragma unroll
      for (int j = 0; j < U; ++j) acc ^= vb[j];
    }
#pragma unroll
    for (int j = 0; j < U; ++j) acc ^= va[j];
  } else {
    for (int s = 0; s < steps; ++s) {
      const int koff = (s * SEG) & kmask;
#pragma unroll
      for (int j = 0; j < U; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + voff[j] + koff),
                                         (__attribute__((address_space(3))) void*)(lds + (((s & 1) * U + j) * NW + wave) * 1024), 16, 0, 0);
      if (U == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (U == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (DST == 1) { __syncthreads(); acc[0] = *reinterpret_cast<unsigned*>(lds + tid * 4); }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[tid] = 1;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SEG, int DST, int NW, int U>
void run(const unsigned char* d, long rowb, int rows_total, int steps, unsigned* out, unsigned long long* cyc, int nwg) {
  const int rows_per_wg = NW * U * (1024 / SEG);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t ldsb = DST ? 2 * U * NW * 1024 : 0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SEG, DST, NW, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SEG, DST, NW, U>), dim3(nwg), dim3(NW * 64), ldsb, 0, d, rowb, rows_total, steps, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg);
  hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto c : h) avg += (double)c; avg /= nwg;
  const double bytes = (double)rows_per_wg * SEG * steps;
  // s_memtime ticks at 100 MHz-derived constant clock on some parts; report both
  printf("SEG %4d DST %d NW %d U %d rows/wg %3d : %.3f ms  %.1f GB/s/CU  %.2f TB/s chip   bytes/memtime-tick/CU %.2f\n", SEG, DST, NW, U,
         rows_per_wg, ms, bytes / (ms * 1e-3) / 1e9, bytes * nwg / (ms * 1e-3) / 1e12, bytes / avg);
}

int main(int argc, char** argv) {
  const long rowb = 2048;                 // Kp = 1024 16-bit elements
  const int rows_total = argc > 1 ? atoi(argv[1]) : 1536;     // 3 MB: inside every XCD's 4-MB L2
  const int nwg = 256;
  unsigned char* d; unsigned* out; unsigned long long* cyc;
  hipMalloc(&d, (size_t)rows_total * rowb + 4096);
  hipMemset(d, 1, (size_t)rows_total * rowb + 4096);
  hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, nwg * 8);
  printf("matrix %d rows x %ld B = %.1f MB\n", rows_total, rowb, rows_total * rowb / 1048576.0);
  const int S = 4096;
  run<64, 0, 4, 8>(d, rowb, rows_total, S, out, cyc, nwg);
  run<128, 0, 4, 8>(d, rowb, rows_total, S, out, cyc, nwg);
  run<256, 0, 4, 8>(d, rowb, rows_total, S, out, cyc, nwg);
  run<1024, 0, 4, 8>(d, rowb, rows_total, S, out, cyc, nwg);
  run<64, 1, 4, 8>(d, rowb, rows_total, S, out, cyc, nwg);
  run<128, 1, 4, 8>(d, rowb, rows_total, S, out, cyc, nwg);
  run<256, 1, 4, 8>(d, rowb, rows_total, S, out, cyc, nwg);
  run<1024, 1, 4, 8>(d, rowb, rows_total, S, out, cyc, nwg);
  run<64, 0, 8, 4>(d, rowb, rows_total, S, out, cyc, nwg);
  run<128, 0, 8, 4>(d, rowb, rows_total, S, out, cyc, nwg);
  run<64, 1, 8, 4>(d, rowb, rows_total, S, out, cyc, nwg);
  run<128, 1, 8, 4>(d, rowb, rows_total, S, out, cyc, nwg);
  run<128, 1, 8, 8>(d, rowb, rows_total, S, out, cyc, nwg);
  run<128, 1, 4, 4>(d, rowb, rows_total, S, out, cyc, nwg);
  return 0;
}
