// write-bandwidth microbenchmark: does the ADDRESS PATTERN of the ring kernel's stores (256 workgroups, each walking its own
// run of 9 KB rows in three tensors) reach the rate of a linear fill?   hipcc --offload-arch=gfx950 -O3 wpattern.hip -o wpattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// mode 0: linear sweep, block b writes rows b, b + grid, ...   (row = C floats)
// mode 1: block b owns rows [b*seg, (b+1)*seg) of each of the 3 tensors, writes them in order
// mode 2: like 1 with runs of R rows dealt round-robin: block b handles runs b, b + grid, ...
__global__ __launch_bounds__(576) void wr(float* y0, float* y1, float* y2, int rows, int C, int seg, int mode, int R, int spin) {
  const int tid = threadIdx.x;            // 576 threads x 16 B = 9216 B = one row of C = 2304
  float* ys[3] = {y0, y1, y2};
  const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  auto put = [&](long row) {
    for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(ys[j] + row * C + tid * 4) = v;
    if (spin) { for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(16); __syncthreads(); }
  };
  if (mode == 0) { for (long r = blockIdx.x; r < rows; r += gridDim.x) put(r); }
  else if (mode == 1) { for (int i = 0; i < seg; ++i) { long r = (long)blockIdx.x * seg + i; if (r < rows) put(r); } }
  else { for (long run = blockIdx.x; run * R < rows; run += gridDim.x) for (int i = 0; i < R; ++i) { long r = run * R + i; if (r < rows) put(r); } }
}
int main() {
  const int C = 2304, rows = 8 * 2304; const size_t n = (size_t)rows * C;
  float *y[3]; for (int j = 0; j < 3; ++j) hipMalloc(&y[j], n * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, int grid, int seg, int mode, int R, int spin) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(wr, dim3(grid), dim3(576), 0, 0, y[0], y[1], y[2], rows, C, seg, mode, R, spin);
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(wr, dim3(grid), dim3(576), 0, 0, y[0], y[1], y[2], rows, C, seg, mode, R, spin);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-44s %7.1f us  %5.2f TB/s\n", name, ms * 1e3, 3.0 * n * 4 / ms / 1e9);
  };
  run("linear sweep, 256 blocks", 256, 0, 0, 0, 0);
  run("linear sweep, 1024 blocks", 1024, 0, 0, 0, 0);
  run("own run of 72 rows, 256 blocks", 256, 72, 1, 0, 0);
  run("own run of 18 rows, 1024 blocks", 1024, 18, 1, 0, 0);
  run("runs of 8 round-robin, 256 blocks", 256, 0, 2, 8, 0);
  run("runs of 4 round-robin, 256 blocks", 256, 0, 2, 4, 0);
  run("runs of 1 round-robin (=linear), 256", 256, 0, 2, 1, 0);
  for (int spin : {4, 8, 16, 32}) {
    char nm[64]; snprintf(nm, 64, "own run of 72, sleep %d + barrier per row", spin);
    run(nm, 256, 72, 1, 0, spin);
    snprintf(nm, 64, "linear, sleep %d + barrier per row", spin);
    run(nm, 256, 0, 0, 0, spin);
  }
  return 0;
}
