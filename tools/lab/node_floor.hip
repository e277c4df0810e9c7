// Per-node cost of a chain of dependent tiny kernels on this runtime: launched one by one on a stream (host loop in C) against the
// same chain captured into a hipGraph and replayed.  tools/lab: hipcc --offload-arch=gfx950 -O2 node_floor.hip -o node_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void bump(float* x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += 1.f;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int chain = argc > 1 ? atoi(argv[1]) : 1000, reps = 20;
  const int grids[3] = {1, 256, 2048};
  float* x; CK(hipMalloc(&x, 2048 * 256 * 4)); CK(hipMemset(x, 0, 2048 * 256 * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int gi = 0; gi < 3; ++gi) {
    const int g = grids[gi], n = g * 256;
    for (int i = 0; i < 100; ++i) bump<<<g, 256, 0, s>>>(x, n);
    CK(hipStreamSynchronize(s));
    double t0 = now();
    for (int r = 0; r < reps; ++r) for (int i = 0; i < chain; ++i) bump<<<g, 256, 0, s>>>(x, n);
    double t_host = now() - t0;
    CK(hipStreamSynchronize(s));
    double t_stream = now() - t0;
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < chain; ++i) bump<<<g, 256, 0, s>>>(x, n);
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(exec, s));
    CK(hipStreamSynchronize(s));
    t0 = now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(exec, s));
    CK(hipStreamSynchronize(s));
    double t_graph = now() - t0;
    printf("grid %4d x 256: stream launches %.2f us per kernel (host enqueue alone %.2f), hipGraph replay %.2f us per node (chain of %d)\n",
           g, t_stream / reps / chain * 1e6, t_host / reps / chain * 1e6, t_graph / reps / chain * 1e6, chain);
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
  }
  return 0;
}
