// State of the in-launch grid barriers (common.h): one two-level {arrive, generation} structure per (stream slot, call site).
// A stream gets a slot the first time it is seen (at most VILCO_SYNC_SLOTS distinct streams per process; later ones
// get null and the callers fall back to the two-launch form).  Zero-initialised at module load, self-resetting.
#include <mutex>
#include "common.h"

__device__ unsigned vilco_sync_words[VILCO_SYNC_SLOTS * VILCO_SYNC_SITES * VILCO_SYNC_WORDS];

namespace {
std::mutex g_mu;
hipStream_t g_streams[VILCO_SYNC_SLOTS];
int g_nstreams = 0;
unsigned* g_base = nullptr;
bool g_disabled = false;
}  // namespace

unsigned* vilco_sync_counter(hipStream_t s, int site) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_disabled) return nullptr;
  if (!g_base) {
    // default OFF: measured on MI355X (r02, same box A/B of the P step) the in-launch forms are 38.2 ms against 37.6 ms
    // for the two-launch forms -- a grid barrier costs 8-10 us here (device-scope atomics execute at the memory side of
    // the 8 XCDs, ~2 us per hop), about what the second launch costs.  VILCO_GRID_SYNC=1 switches them on (fewer
    // launches: the better trade when the host, not the GPU, is the bottleneck).
    const char* e = getenv("VILCO_GRID_SYNC");
    if (!e || e[0] != '1') { g_disabled = true; return nullptr; }
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(vilco_sync_words)) != hipSuccess || !p) { g_disabled = true; return nullptr; }
    g_base = reinterpret_cast<unsigned*>(p);
  }
  int slot = -1;
  for (int i = 0; i < g_nstreams; ++i)
    if (g_streams[i] == s) { slot = i; break; }
  if (slot < 0) {
    if (g_nstreams >= VILCO_SYNC_SLOTS) return nullptr;
    slot = g_nstreams;
    g_streams[g_nstreams++] = s;
  }
  return g_base + ((long)slot * VILCO_SYNC_SITES + site) * VILCO_SYNC_WORDS;
}

// number of grid barriers that gave up waiting since the library was loaded (0 in a healthy process); synchronises
extern "C" int vilco_sync_timeouts_read(void) {
  static unsigned w[VILCO_SYNC_SLOTS * VILCO_SYNC_SITES * VILCO_SYNC_WORDS];
  if (hipMemcpyFromSymbol(w, HIP_SYMBOL(vilco_sync_words), sizeof(w)) != hipSuccess) return -1;
  long v = 0;
  for (int i = 0; i < VILCO_SYNC_SLOTS * VILCO_SYNC_SITES; ++i) v += w[(long)i * VILCO_SYNC_WORDS + 2];
  return (int)v;
}

// ---- step word of the counter-based dropout masks (common.h: vilco_step_seed)
__device__ uint32_t vilco_seed_word_storage[16];      // word 0; its own 64-byte line

const uint32_t* vilco_seed_word_dev() {
  static const uint32_t* p = [] {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(vilco_seed_word_storage)) != hipSuccess) q = nullptr;
    return reinterpret_cast<const uint32_t*>(q);
  }();
  return p;
}

__global__ void seed_word_kernel(uint32_t* w, uint32_t v, int add) { *w = add ? *w + v : v; }

extern "C" int vilco_seed_word_set(uint32_t value, void* stream) {
  uint32_t* w = const_cast<uint32_t*>(vilco_seed_word_dev());
  if (!w) return VILCO_ERR_LAUNCH;
  hipLaunchKernelGGL(seed_word_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), w, value, 0);
  return vilco_launch_status();
}

extern "C" int vilco_seed_word_bump(void* stream) {
  uint32_t* w = const_cast<uint32_t*>(vilco_seed_word_dev());
  if (!w) return VILCO_ERR_LAUNCH;
  hipLaunchKernelGGL(seed_word_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), w, 1u, 1);
  return vilco_launch_status();
}

// current value (synchronises the device: tests / logging only)
extern "C" int vilco_seed_word_get(uint32_t* out) {
  if (!out) return VILCO_ERR_BADARG;
  if (hipDeviceSynchronize() != hipSuccess) return VILCO_ERR_LAUNCH;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(vilco_seed_word_storage), sizeof(uint32_t)) == hipSuccess ? VILCO_OK : VILCO_ERR_LAUNCH;
}
