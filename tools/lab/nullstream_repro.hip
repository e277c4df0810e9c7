// Round 6: is "graph launch -> small eager kernel -> graph launch" ordered on the NULL stream when the host
// runs far ahead of the device?  Plain HIP, no torch, no RCCL.  The captured graph forks a side branch the way the captured
// training step does (ops.fork_enabled); a device counter is bumped by every launch and both branches write it into a small
// buffer after spinning; the eager copy behind the launch must read the value of THAT launch.
//   build: hipcc --offload-arch=gfx950 -O2 -o nullstream_repro nullstream_repro.hip
//   run:   ./nullstream_repro <stream: 0 null | 1 created | 2 created non-blocking> <pattern 0..3> [iters] [spin cycles] [chain]
// chain (default 1): kernel nodes per branch -- the captured step's graphs have hundreds of nodes in a few branches
// patterns: 0 copy kernels on the launch stream; 1 hipMemcpyAsync D2D on the launch stream; 2 the copies on a SECOND stream ordered
// by events both ways (what a collective's stream does); 3 as 0 with two graphs (stages) per iteration and a copy after each;
// 4 the graph holds what the stale gradients had in common (loss.hip:402 / eltwise.hip colsum): a MEMSET node, then 64 workgroups
// adding the counter into the cleared buffer with device-scope float atomics; the eager copy must read 64 x (launch number);
// 5 / 6 the data-parallel replay's own sequence (vilco_amd/graph.py: _replay_staged + dist.GradReducer): graph 1, gather copy on the
// launch stream, an in-place "collective" on a SECOND stream behind an event (5: that stream blocking, 6: NON-BLOCKING, as torch's and
// RCCL's streams are), graph 2 launched on the launch stream while the collective runs, then the wait for the collective, the copy
// back into the gradient's own memory and the check;
// 7 / 8 / 9 what the torch-level bisect left (tools/lab/dp_staged_dbg2.py with VILCO_DP_DEBUG_NO_COLLECTIVE=2 fails on the null stream
// WITHOUT any RCCL call): as 6 but NOTHING runs on the second stream -- it only waits for the launch stream's event and records the end
// event the launch stream then waits for; the second stream is high-priority non-blocking (torch's pool); 8: the two events are created
// and destroyed every iteration, as torch.cuda.Event objects are; 9: as 7 with the end event queried from a second host thread while
// the loop runs (what a collective's watchdog does);
// 10: as 8, and the graph produces its value the way ATen's multi-block reductions do (the stale tensors of the real step -- XLNet's
// r_w_bias / r_r_bias gradients -- are outputs of `aten::sum`): a MEMSET node clears a semaphore, 64 workgroups add to it, and only the
// workgroup that sees itself last writes the result -- a semaphore that is not zero when the kernel starts leaves the OLD result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <atomic>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void bump(unsigned* ctr) { if (threadIdx.x == 0) *ctr = *ctr + 1u; }
__global__ void spin_write(float* g, const unsigned* ctr, int n, long spin) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long)(__builtin_amdgcn_s_memtime() - t0) < spin) { }
  if ((int)threadIdx.x < n) g[threadIdx.x] = (float)(*ctr);
}
__global__ void atomic_acc(float* acc, const unsigned* ctr, int n, long spin) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long)(__builtin_amdgcn_s_memtime() - t0) < spin) { }
  if ((int)threadIdx.x < n) atomicAdd(acc + threadIdx.x, (float)(*ctr));
}
__global__ void slow_inplace(float* b, int n, long spin) {      // stands in for the all-reduce (one rank: identity), takes a while
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const float v = (int)threadIdx.x < n ? b[threadIdx.x] : 0.f;
  while ((long)(__builtin_amdgcn_s_memtime() - t0) < spin) { }
  if ((int)threadIdx.x < n) b[threadIdx.x] = v;
}
__global__ void last_block_writes(float* g, unsigned* sem, const unsigned* ctr, int n, long spin) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long)(__builtin_amdgcn_s_memtime() - t0) < spin) { }
  __shared__ unsigned prev;
  if (threadIdx.x == 0) prev = atomicAdd(sem, 1u);
  __syncthreads();
  if (prev == gridDim.x - 1 && (int)threadIdx.x < n) g[threadIdx.x] = (float)(*ctr);
}
__global__ void copy_k(const float* __restrict__ s, float* __restrict__ d, int n) { if ((int)threadIdx.x < n) d[threadIdx.x] = s[threadIdx.x]; }

int main(int argc, char** argv) {
  const int smode = argc > 1 ? atoi(argv[1]) : 0, pat = argc > 2 ? atoi(argv[2]) : 0;
  const int iters = argc > 3 ? atoi(argv[3]) : 3000;
  const long spin = argc > 4 ? atol(argv[4]) : 20000;       // 100 MHz counter: 20000 = 200 us per branch
  const int chain = argc > 5 ? atoi(argv[5]) : 1;
  const int n = 64;
  unsigned* ctr; float *gA, *gB, *gC, *out;
  CK(hipMalloc(&ctr, 4)); CK(hipMemset(ctr, 0, 4));
  CK(hipMalloc(&gA, n * 4)); CK(hipMalloc(&gB, n * 4)); CK(hipMalloc(&gC, n * 4));
  CK(hipMalloc(&out, (size_t)iters * 3 * n * 4)); CK(hipMemset(out, 0, (size_t)iters * 3 * n * 4));
  hipStream_t cap, side, S = nullptr, C2;
  CK(hipStreamCreate(&cap)); CK(hipStreamCreate(&side));
  if (pat >= 7) CK(hipStreamCreateWithPriority(&C2, hipStreamNonBlocking, -1));
  else if (pat == 6) CK(hipStreamCreateWithFlags(&C2, hipStreamNonBlocking)); else CK(hipStreamCreate(&C2));
  std::atomic<hipEvent_t> watched{nullptr}; std::atomic<bool> stop{false};
  std::thread watchdog;
  if (pat == 9) watchdog = std::thread([&] { while (!stop.load()) { hipEvent_t e = watched.load(); if (e) (void)hipEventQuery(e); } });
  float* bucket; CK(hipMalloc(&bucket, n * 4)); CK(hipMemset(bucket, 0, n * 4));
  unsigned* sems; CK(hipMalloc(&sems, 8)); CK(hipMemset(sems, 0, 8));
  if (smode == 1) CK(hipStreamCreate(&S));
  if (smode == 2) CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  hipEvent_t f, j, e1, e2;
  CK(hipEventCreateWithFlags(&f, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  // graph 1: bump -> { main: spin_write(gA) ; side: spin_write(gB), shorter } -> join.   graph 2 (pattern 3): spin_write(gC)
  hipGraph_t g1, g2; hipGraphExec_t x1, x2;
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, cap, ctr);
  CK(hipEventRecord(f, cap)); CK(hipStreamWaitEvent(side, f, 0));
  if (pat == 10) {
    CK(hipMemsetAsync(sems, 0, 4, cap)); CK(hipMemsetAsync(sems + 1, 0, 4, side));
    hipLaunchKernelGGL(last_block_writes, dim3(64), dim3(64), 0, side, gB, sems + 1, ctr, n, spin / 2);
    hipLaunchKernelGGL(last_block_writes, dim3(64), dim3(64), 0, cap, gA, sems, ctr, n, spin);
  } else if (pat == 4) {
    CK(hipMemsetAsync(gA, 0, n * 4, cap)); CK(hipMemsetAsync(gB, 0, n * 4, side));
    for (int c = 0; c < chain; ++c) {
      hipLaunchKernelGGL(atomic_acc, dim3(64 / chain > 0 ? 64 / chain : 1), dim3(64), 0, side, gB, ctr, n, spin / 2 / chain);
      hipLaunchKernelGGL(atomic_acc, dim3(64 / chain > 0 ? 64 / chain : 1), dim3(64), 0, cap, gA, ctr, n, spin / chain);
    }
  } else
  for (int c = 0; c < chain; ++c) {
    hipLaunchKernelGGL(spin_write, dim3(1), dim3(64), 0, side, gB, ctr, n, spin / 2 / chain);
    hipLaunchKernelGGL(spin_write, dim3(1), dim3(64), 0, cap, gA, ctr, n, spin / chain);
  }
  CK(hipEventRecord(j, side)); CK(hipStreamWaitEvent(cap, j, 0));
  CK(hipStreamEndCapture(cap, &g1)); CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0));
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(spin_write, dim3(1), dim3(64), 0, cap, gC, ctr, n, spin / 4 / chain);
  CK(hipStreamEndCapture(cap, &g2)); CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
  for (int it = 0; it < iters; ++it) {                       // no host wait anywhere in here
    float* o = out + (size_t)it * 3 * n;
    CK(hipGraphLaunch(x1, S));
    if (pat >= 7) {
      hipEvent_t a = e1, b = e2;
      if (pat == 8 || pat == 10) { CK(hipEventCreateWithFlags(&a, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&b, hipEventDisableTiming)); }
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, S, gA, bucket, n);                 // gather
      CK(hipEventRecord(a, S)); CK(hipStreamWaitEvent(C2, a, 0));
      CK(hipEventRecord(b, C2));                                                          // (nothing runs on C2)
      if (pat == 9) watched.store(b);
      CK(hipGraphLaunch(x2, S));
      CK(hipStreamWaitEvent(S, b, 0));
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, S, bucket, gA, n);                 // copy back
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, S, gA, o, n);
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, S, gB, o + n, n);
      if (pat == 8 || pat == 10) { CK(hipEventDestroy(a)); CK(hipEventDestroy(b)); }
    } else if (pat == 5 || pat == 6) {
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, S, gA, bucket, n);                 // gather
      CK(hipEventRecord(e1, S)); CK(hipStreamWaitEvent(C2, e1, 0));
      hipLaunchKernelGGL(slow_inplace, dim3(1), dim3(64), 0, C2, bucket, n, spin / 2);   // the "collective"
      CK(hipEventRecord(e2, C2));
      CK(hipGraphLaunch(x2, S));                                                          // the next stage runs under it
      CK(hipStreamWaitEvent(S, e2, 0));
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, S, bucket, gA, n);                 // copy back into the gradient itself
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, S, gA, o, n);
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, S, gB, o + n, n);
    } else if (pat == 0 || pat == 3 || pat == 4) {
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, S, gA, o, n);
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, S, gB, o + n, n);
    } else if (pat == 1) {
      CK(hipMemcpyAsync(o, gA, n * 4, hipMemcpyDeviceToDevice, S));
      CK(hipMemcpyAsync(o + n, gB, n * 4, hipMemcpyDeviceToDevice, S));
    } else {
      CK(hipEventRecord(e1, S)); CK(hipStreamWaitEvent(C2, e1, 0));
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, C2, gA, o, n);
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, C2, gB, o + n, n);
      CK(hipEventRecord(e2, C2)); CK(hipStreamWaitEvent(S, e2, 0));
    }
    if (pat == 3) {
      CK(hipGraphLaunch(x2, S));
      hipLaunchKernelGGL(copy_k, dim3(1), dim3(64), 0, S, gC, o + 2 * n, n);
    }
  }
  CK(hipDeviceSynchronize());
  if (pat == 9) { stop.store(true); watchdog.join(); }
  std::vector<float> h((size_t)iters * 3 * n);
  CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
  long bad = 0; int first = -1;
  for (int it = 0; it < iters; ++it)
    for (int b = 0; b < (pat == 3 ? 3 : 2); ++b)
      for (int k = 0; k < n; ++k)
        if (h[((size_t)it * 3 + b) * n + k] != (float)(it + 1) * (pat == 4 ? (float)((64 / chain > 0 ? 64 / chain : 1) * chain) : 1.f)) { ++bad; if (first < 0) first = it; }
  int ver = 0; CK(hipRuntimeGetVersion(&ver));
  printf("stream %s pattern %d iters %d spin %ld chain %d: %ld wrong values%s (first at iteration %d)  [HIP runtime %d]\n",
         smode == 0 ? "NULL" : (smode == 1 ? "created" : "created-nonblocking"), pat, iters, spin, chain, bad, bad ? "  <-- ORDER VIOLATED" : "", first, ver);
  return bad ? 1 : 0;
}
