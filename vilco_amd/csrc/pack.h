// bf16 "plane" packing shared by the GEMM and the attention kernels: an fp32 operand is split once into NP bf16
// parts (x = p0 + p1 + p2, p0 = bf16(x), p1 = bf16(x - p0), ...), laid out [part][batch][row][Kp] with k contiguous
// and zero padded to a multiple of 32; pack_tr transposes on the way through an LDS tile.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// fmt 1 ("f16x2"): two fp16 parts of x * s, s a per-tensor power of two that puts max|x| in [2^14, 2^15):
// x*s = h0 + h1 to 22 bits (h0 = fp16(x*s), h1 = fp16(x*s - h0); absolute floor 2^-25 = 2^-40 of the tensor max),
// so hi*hi + hi*lo + lo*hi = 3 MFMAs reach 2^-22 -- against 6 MFMAs for the three bf16 parts.  The planes are 16-bit
// either way and share every layout below; amax_kernel leaves per-block partial maxima in the GEMM workspace and
// every pack block folds them (<= 4 KB from L2) into the scale: no atomics, no memset, no extra launch.

namespace {

template <int NP>
__device__ __forceinline__ void splitN(const float (&v)[8], bf16x8 (&part)[3], float f16_scale = 0.f) {
  if (f16_scale != 0.f) {          // fmt 1: NP == 2 fp16 parts of the scaled value
    f16x8 h0, h1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = v[e] * f16_scale;   // exact (power of two)
      const _Float16 h = (_Float16)x;
      h0[e] = h;
      h1[e] = (_Float16)(x - (float)h);
    }
    part[0] = __builtin_bit_cast(bf16x8, h0);
    part[1] = __builtin_bit_cast(bf16x8, h1);
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)v[e];
    part[0][e] = h;
    if (NP >= 2) {
      const float r1 = v[e] - (float)h;      // exact in fp32
      const __bf16 m = (__bf16)r1;
      part[1][e] = m;
      if (NP >= 3) part[2][e] = (__bf16)(r1 - (float)m);
    }
  }
}

// fmt 1 scale from the amax partials: s = 2^(141 - e), e the biased exponent of amax, so amax * s is in [2^14, 2^15)
__device__ __forceinline__ float f16_scale_from(const float* parts, int nparts, float* out, bool writer) {
  __shared__ float red_[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 1024) {      // four partials in flight (clamped index: a repeat changes no maximum)
    const int last = nparts - 1;
    const float p0 = vilco_ld_agent(parts + i), p1 = vilco_ld_agent(parts + (i + 256 < last ? i + 256 : last)),   // may cross an in-launch barrier
                p2 = vilco_ld_agent(parts + (i + 512 < last ? i + 512 : last)), p3 = vilco_ld_agent(parts + (i + 768 < last ? i + 768 : last));
    m = fmaxf(fmaxf(m, fmaxf(p0, p1)), fmaxf(p2, p3));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red_[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red_[0], red_[1]), fmaxf(red_[2], red_[3]));
  int e = (int)((__float_as_uint(m) >> 23) & 0xff);
  if (e < 15) e = 15;                                     // tiny / zero tensors: any scale works
  if (e > 250) e = 250;                                   // inf input: the result is garbage either way
  const float sc = __uint_as_float((unsigned)(268 - e) << 23);
  if (writer) { out[0] = __uint_as_float((unsigned)(e - 14) << 23); out[1] = sc; }   // {1/s, s} for the consumer kernel
  return sc;
}

// ------------------------------------------------------------------------------------------ pack
// "kc" source: element (r,k) at src[r*ld + k];  "tr" source: element (r,k) at src[k*ld + r].
struct PackArgs {
  const float* src;
  __bf16* dst;         // part 0, batch 0
  long ld;
  int rows, K, Kp;
  long plane_stride;   // elements between parts
  long batch_stride;   // elements between batches inside one part
  int nbi;             // packed inner batch count
  long so, si;         // source batch strides
  // tap modes (k=3 conv):  0 none
  //  1 (kc, seqpad): rows = B*T source rows of width tapC; output has T+2 rows per sequence (first and last
  //                  zero) plus zero slack rows; the GEMM reads 3*tapC-wide overlapped spans from it
  //  2 (kc, expand): output row r = [x[t-1] | x[t] | x[t+1]] explicitly (K = 3*tapC), for tapC % 8 != 0
  //  3 (tr, taps):   output row (j*tapC + c), column tok = x[tok + j - 1][c], zero across sequence ends
  //  4 (kc, relshift): the logical matrix is XLNet's UNSHIFTED [rows][rows + tapC] view of a [rows][tapC] source:
  //                  element (i, p) = src[i][p - rows + i] when that column exists, else 0 (the adjoint of
  //                  rel_shift_bnij, modeling_xlnet_x.py:204-214): dS -> d(bd) without materialising it
  int tap, tapC, tapT;
  int out_rows;        // rows written by the kc kernel
  int vec;             // 16-byte aligned source rows
  const float* amax;   // fmt 1: per-block partial maxima of |src| from amax_kernel (null = bf16 parts)
  int namax;
  float* inv_scale;    // fmt 1: where block 0 leaves {1/s, s} for the consumer kernel
};

// workgroups of a kc pack: every block folds the amax partials before it can scale anything (a load round trip, a block
// reduction and a barrier), so more blocks than the chip holds at once only repeat that prologue: 768 = three per CU
// (r03, tools/lab/pack_time.py, [4608, 1024]: 2048 -> 14.3 us, 1024 -> 13.7, 768 -> 13.4, 512 -> 13.6, 256 -> 14.4)
inline long kc_cap() {
  static const long cap = [] { const char* e = getenv("VILCO_PACK_CAP"); const long v = e ? atol(e) : 0; return v > 0 ? v : 768L; }();
  return cap;
}

template <int NP>
__device__ __forceinline__ void pack_kc_body(const PackArgs& a, int z, int bx, int nbx) {
  const int zo = z / a.nbi, zi = z % a.nbi;
  const float* src = a.src + zo * a.so + zi * a.si;
  __bf16* dst = a.dst + (long)z * a.batch_stride;
  const int width = (a.tap == 1) ? a.tapC : a.Kp;       // elements per output row
  const int kmax = (a.tap == 1) ? a.tapC : a.K;         // valid source columns
  const int chunks = width >> 3;
  const long total = (long)a.out_rows * chunks;
  // the 8 source values of item i (row orow, columns k0 .. k0+7 of the packed row)
  auto load_item = [&](long i, float (&v)[8]) {
    const int c = (int)(i % chunks);
    const long orow = i / chunks;
    const int k0 = c * 8;
    long srow = orow;
    bool row_ok = orow < a.rows;
    if (a.tap == 1) {
      const long seq = orow / (a.tapT + 2);
      const int tt = (int)(orow % (a.tapT + 2)) - 1;
      srow = seq * a.tapT + tt;
      row_ok = tt >= 0 && tt < a.tapT && srow < a.rows;
    }
    if (a.tap == 4) {
      const long j0 = (long)k0 - a.rows + orow;       // source column of the chunk's first element
      if (row_ok && j0 >= 0 && j0 + 8 <= a.tapC) {
        struct __attribute__((packed, aligned(4))) f4u_ { float v[4]; };
        const f4u_ t0 = *reinterpret_cast<const f4u_*>(src + orow * a.ld + j0);
        const f4u_ t1 = *reinterpret_cast<const f4u_*>(src + orow * a.ld + j0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = t0.v[e]; v[4 + e] = t1.v[e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (row_ok && j0 + e >= 0 && j0 + e < a.tapC) ? src[orow * a.ld + j0 + e] : 0.f;
      }
    } else if (a.tap == 2) {
      const int t = (int)(orow % a.tapT);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        const int j = k / a.tapC, cc = k % a.tapC;
        const bool ok = row_ok && k < a.K && !((j == 0 && t == 0) || (j == 2 && t == a.tapT - 1));
        v[e] = ok ? src[(orow + j - 1) * a.ld + cc] : 0.f;
      }
    } else if (row_ok && a.vec && k0 + 8 <= kmax) {
      const float4 v0 = *reinterpret_cast<const float4*>(src + srow * a.ld + k0);
      const float4 v1 = *reinterpret_cast<const float4*>(src + srow * a.ld + k0 + 4);
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (row_ok && k0 + e < kmax) ? src[srow * a.ld + k0 + e] : 0.f;
    }
  };
  // The first item's source values are requested BEFORE the scale is folded out of the amax partials (a load, a block
  // reduction and a barrier of its own): the two round trips overlap instead of following each other.
  if (a.tap == 0 && a.vec && a.K == a.Kp && a.out_rows <= a.rows &&
      total < (1L << 31) - 2L * nbx * (long)blockDim.x) {      // i + 2 * st below stays inside an int
    // the plain case (an activation or a weight, whole rows): 32-bit item arithmetic, two items of a thread in flight
    const int tot = (int)total, st = nbx * (int)blockDim.x;
    int i = bx * (int)blockDim.x + (int)threadIdx.x;
    auto ld8 = [&](int it, float4& v0, float4& v1) {
      const int row = it / chunks, k0 = (it - row * chunks) * 8;
      const float* p = src + (long)row * a.ld + k0;
      v0 = *reinterpret_cast<const float4*>(p); v1 = *reinterpret_cast<const float4*>(p + 4);
    };
    float4 a0, a1, b0, b1;
    const int ic = i < tot ? i : tot - 1, jc = i + st < tot ? i + st : tot - 1;      // clamped: loads are unconditional
    if (tot > 0) { ld8(ic, a0, a1); ld8(jc, b0, b1); }
    const float fs = a.amax ? f16_scale_from(a.amax, a.namax, a.inv_scale, bx == 0 && z == 0 && threadIdx.x == 0) : 0.f;
    while (i < tot) {
      const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      bf16x8 part[3];
      splitN<NP>(v, part, fs);
      const long o = (long)i * 8;                                   // width == Kp == chunks * 8: item i starts at element 8 i
      a0 = b0; a1 = b1;
      const int nn = i + 2 * st < tot ? i + 2 * st : tot - 1;
      ld8(nn, b0, b1);
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8*>(dst + q * a.plane_stride + o) = part[q];
      i += st;
    }
    return;
  }
  const long step = (long)nbx * blockDim.x;
  long i = (long)bx * blockDim.x + threadIdx.x;
  float v[8];
  if (i < total) load_item(i, v);
  const float fs = a.amax ? f16_scale_from(a.amax, a.namax, a.inv_scale, bx == 0 && z == 0 && threadIdx.x == 0) : 0.f;
  while (i < total) {
    bf16x8 part[3];
    splitN<NP>(v, part, fs);
    const long o = (i / chunks) * (long)width + (int)(i % chunks) * 8;
    const long inext = i + step;
    if (inext < total) load_item(inext, v);
#pragma unroll
    for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8*>(dst + q * a.plane_stride + o) = part[q];
    i = inext;
  }
}

template <int NP>
__global__ __launch_bounds__(256) void pack_kc_kernel(PackArgs a) { pack_kc_body<NP>(a, blockIdx.z, blockIdx.x, gridDim.x); }

// up to four kc packs with the same batch count in one launch (blockIdx.y = operand): an activation and its
// layer's weight, or the natural planes of q / k / v / dO
struct PackArgs4 { PackArgs a[4]; };
template <int NP>
__global__ __launch_bounds__(256) void pack_kc_multi_kernel(PackArgs4 args) {
  pack_kc_body<NP>(args.a[blockIdx.y], blockIdx.z, blockIdx.x, gridDim.x);
}

// transposing pack: 64(r) x 64(k) tile through LDS.  grid = (ceil(nrows/64), ceil(Kp/64), batch * ntap)
template <int NP>
__device__ __forceinline__ void pack_tr_body(const PackArgs& a, int bz) {
  __shared__ float tile[64][65];
  const int ntap = (a.tap == 3) ? 3 : 1;
  const int nrows = (a.tap == 3) ? a.tapC : a.rows;
  if ((int)blockIdx.x * 64 >= nrows || (int)blockIdx.y * 64 >= a.Kp) return;     // multi launch: grid is the max
  const int z = bz / ntap, j = bz % ntap;
  const int zo = z / a.nbi, zi = z % a.nbi;
  const float* src = a.src + zo * a.so + zi * a.si;
  __bf16* dst = a.dst + (long)z * a.batch_stride;
  const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  // load 64 k-rows x 64 r, float4 along r: the four loads of a thread are issued together (full, aligned quads from a
  // clamped row; only edge tiles take the guarded element-wise path)
  float4 tv[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int id = tid + it * 256;
    const int kk = id >> 4, r4 = (id & 15) * 4;
    const int k = k0 + kk, r = r0 + r4;
    bool ok = k < a.K;
    long ksrc = k;
    if (a.tap == 3) {
      const int t = k % a.tapT;
      if ((j == 0 && t == 0) || (j == 2 && t == a.tapT - 1)) ok = false;
      ksrc = (long)k + j - 1;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.vec && r + 4 <= nrows) {
      const float4 q = *reinterpret_cast<const float4*>(src + (ok ? ksrc : 0) * a.ld + r);
      if (ok) v = q;
    } else if (ok && r < nrows) {
      const float* p = src + ksrc * a.ld + r;
      v.x = p[0];
      if (r + 1 < nrows) v.y = p[1];
      if (r + 2 < nrows) v.z = p[2];
      if (r + 3 < nrows) v.w = p[3];
    }
    tv[it] = v;
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int id = tid + it * 256;
    const int kk = id >> 4, r4 = (id & 15) * 4;
    tile[kk][r4] = tv[it].x; tile[kk][r4 + 1] = tv[it].y; tile[kk][r4 + 2] = tv[it].z; tile[kk][r4 + 3] = tv[it].w;
  }
  // the scale is folded out of the amax partials while the tile loads are in flight
  const float fs = a.amax ? f16_scale_from(a.amax, a.namax, a.inv_scale, (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && tid == 0) : 0.f;
  __syncthreads();
  const int r = tid & 63;                 // store: thread -> (row r, two 8-k chunks)
  if (r0 + r < nrows) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kc = (tid >> 6) + h * 4;
      if (k0 + kc * 8 >= a.Kp) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[kc * 8 + e][r];
      bf16x8 part[3];
      splitN<NP>(v, part, fs);
      const long o = ((long)j * a.tapC * (a.tap == 3) + r0 + r) * a.Kp + k0 + kc * 8;
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8*>(dst + q * a.plane_stride + o) = part[q];
    }
  }
}

// ------------------------------------------------------------------------------------------ amax (fmt 1)
// Every pack source is a strided matrix [R][W] (W contiguous) per batch; up to four operands per launch
// (blockIdx.y); block b leaves max|x| over its rows in parts[b].
constexpr int AMAX_MAX_BLOCKS = VILCO_AMAX_MAX_BLOCKS;

struct AmaxOp {
  const float* src;
  long ld, so, si;
  int R, W, nbo, nbi, vec;
  int tw;              // threads per row (power of two <= 256)
  int nblocks;
  float* parts;        // nblocks partial maxima
};
struct AmaxArgs { AmaxOp op[4]; };

__global__ __launch_bounds__(256) void amax_kernel(AmaxArgs args) {
  const AmaxOp& o = args.op[blockIdx.y];
  if ((int)blockIdx.x >= o.nblocks) return;
  __shared__ float red[4];
  float m = 0.f;
  // rows are dealt to blocks; inside a block `tw` threads (a power of two) walk one row, 256/tw rows at a time:
  // no per-element division
  const int total_rows = o.nbo * o.nbi * o.R;
  const int sub = threadIdx.x / o.tw, col = threadIdx.x & (o.tw - 1), rpi = 256 / o.tw;
  const bool vec = o.vec && (o.W & 3) == 0;
  const int wq = vec ? (o.W >> 2) : o.W;
  auto row_ptr = [&](int row) {
    const int z = row / o.R, r = row - z * o.R;
    return o.src + (long)(z / o.nbi) * o.so + (long)(z % o.nbi) * o.si + (long)r * o.ld;
  };
  const int rstep = o.nblocks * rpi;
  int row = blockIdx.x * rpi + sub;
  if (vec && wq <= o.tw) {
    // a row is one float4 per thread: four rows' loads in flight instead of one round trip per row (a clamped repeat of
    // the last row changes no maximum)
    if (col < wq)
      for (; row < total_rows; row += 4 * rstep) {
        const int lastrow = total_rows - 1;
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ru = row + u * rstep;
          v[u] = *reinterpret_cast<const float4*>(row_ptr(ru < total_rows ? ru : lastrow) + col * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
      }
  } else {
    for (; row < total_rows; row += rstep) {
      const float* p = row_ptr(row);
      if (vec) {
        for (int c = col; c < wq; c += o.tw) {
          const float4 v = *reinterpret_cast<const float4*>(p + c * 4);
          m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
      } else {
        for (int c = col; c < wq; c += o.tw) m = fmaxf(m, fabsf(p[c]));
      }
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) o.parts[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// the [R][W] view of a pack source
inline AmaxOp amax_view(const PackArgs& a, bool tr, int nbo, float* parts) {
  AmaxOp o;
  o.src = a.src; o.ld = a.ld; o.so = a.so; o.si = a.si; o.nbo = nbo; o.nbi = a.nbi; o.vec = a.vec; o.parts = parts;
  const bool tapped = a.tap != 0;
  if (!tr) { o.R = a.rows; o.W = tapped ? a.tapC : a.K; }
  else { o.R = a.K; o.W = tapped ? a.tapC : a.rows; }
  const int wq = (o.vec && (o.W & 3) == 0) ? o.W >> 2 : o.W;
  o.tw = 1;
  while (o.tw < 256 && o.tw < wq) o.tw <<= 1;
  const long elems = (long)o.R * o.W * nbo * a.nbi;
  long nb = (elems + 8191) / 8192;                        // ~8 float4 per thread
  o.nblocks = (int)(nb < 1 ? 1 : (nb > AMAX_MAX_BLOCKS ? AMAX_MAX_BLOCKS : nb));
  return o;
}

inline void launch_amax(AmaxArgs& am, int nops, hipStream_t s) {
  static const bool trace = getenv("VILCO_AMAX_TRACE") != nullptr;          // (diagnosis: which tensors still need a pass of their own)
  if (trace)
    for (int i = 0; i < nops; ++i)
      fprintf(stderr, "amax_launch %s R=%d W=%d nb=%d\n", VILCO_TU, am.op[i].R, am.op[i].W, am.op[i].nbo * am.op[i].nbi);
  int gx = 1;
  for (int i = 0; i < nops; ++i) gx = am.op[i].nblocks > gx ? am.op[i].nblocks : gx;
  hipLaunchKernelGGL(amax_kernel, dim3(gx, nops), dim3(256), 0, s, am);
}

template <int NP>
__global__ __launch_bounds__(256) void pack_tr_kernel(PackArgs a) { pack_tr_body<NP>(a, blockIdx.z); }

// untapped transposing packs with the same batch count in one launch: blockIdx.z = operand * nbatch + batch
template <int NP>
__global__ __launch_bounds__(256) void pack_tr_multi_kernel(PackArgs4 args, int nbatch) {
  pack_tr_body<NP>(args.a[blockIdx.z / nbatch], blockIdx.z % nbatch);
}

template <int NP>
void launch_pack(const PackArgs& a, bool tr, int nbatch, hipStream_t s) {
  if (!tr) {
    const int width = (a.tap == 1) ? a.tapC : a.Kp;
    long blocks = ((long)a.out_rows * (width / 8) + 255) / 256;
    if (blocks > kc_cap()) blocks = kc_cap();
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((pack_kc_kernel<NP>), dim3((int)blocks, 1, nbatch), dim3(256), 0, s, a);
  } else {
    const int ntap = a.tap == 3 ? 3 : 1;
    const int nrows = a.tap == 3 ? a.tapC : a.rows;
    dim3 grid((nrows + 63) / 64, (a.Kp + 63) / 64, nbatch * ntap);
    hipLaunchKernelGGL((pack_tr_kernel<NP>), grid, dim3(256), 0, s, a);
  }
}

inline int kc_blocks(const PackArgs& a) {
  const int width = (a.tap == 1) ? a.tapC : a.Kp;
  long blocks = ((long)a.out_rows * (width / 8) + 255) / 256;
  return (int)(blocks > kc_cap() ? kc_cap() : (blocks < 1 ? 1 : blocks));
}

// ------------------------------------------------------------------------------------------ amax + pack in ONE launch
// fmt 1 needs the tensor's amax before anything can be packed: phase 1 = every block's partial maximum over its share of
// the source rows, grid barrier (common.h), phase 2 = the kc pack, which folds the partials into the scale exactly as
// after a separate amax launch.  Up to four operands with the same batch count; the grid (gx, nops, nbatch) is sized
// by the host to stay co-resident.
struct FusedPackArgs { PackArgs a[4]; AmaxOp m[4]; unsigned* sync; };

template <int NP>
__global__ __launch_bounds__(256) void pack_kc_fused_kernel(FusedPackArgs args) {
  {
    const AmaxOp& o = args.m[blockIdx.y];
    __shared__ float red[4];
    const int nblk = gridDim.x * gridDim.z, lin = blockIdx.z * gridDim.x + blockIdx.x;
    float m = 0.f;
    const int total_rows = o.nbo * o.nbi * o.R;
    const int sub = threadIdx.x / o.tw, col = threadIdx.x & (o.tw - 1), rpi = 256 / o.tw;
    const bool vec = o.vec && (o.W & 3) == 0;
    const int wq = vec ? (o.W >> 2) : o.W;
    for (int row = lin * rpi + sub; row < total_rows; row += nblk * rpi) {
      const int z = row / o.R, r = row - z * o.R;
      const float* p = o.src + (long)(z / o.nbi) * o.so + (long)(z % o.nbi) * o.si + (long)r * o.ld;
      if (vec) {
        for (int c = col; c < wq; c += o.tw) {
          const float4 v = *reinterpret_cast<const float4*>(p + c * 4);
          m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
      } else {
        for (int c = col; c < wq; c += o.tw) m = fmaxf(m, fabsf(p[c]));
      }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) vilco_st_agent(o.parts + lin, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
  }
  vilco_grid_barrier(args.sync, gridDim.x * gridDim.y * gridDim.z, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
  pack_kc_body<NP>(args.a[blockIdx.y], blockIdx.z, blockIdx.x, gridDim.x);
}

// Launches the fused form when a barrier counter is available and the grid fits; returns false (nothing launched) when
// the caller has to use launch_amax + dispatch_pack*.  On success a[i].amax / namax describe the partials it wrote, so a
// transposing pack of the same tensor launched afterwards can reuse them.
inline bool dispatch_pack_fused(int NP, PackArgs* a, AmaxOp* m, int n, int nbatch, hipStream_t s, int site) {
  if (n < 1 || n > 4 || (long)n * nbatch > VILCO_SYNC_MAX_BLOCKS) return false;
  unsigned* sync = vilco_sync_counter(s, site);
  if (!sync) return false;
  int gx = 1;
  for (int i = 0; i < n; ++i) gx = kc_blocks(a[i]) > gx ? kc_blocks(a[i]) : gx;
  const int cap = VILCO_SYNC_MAX_BLOCKS / (n * nbatch);
  if (gx > cap) gx = cap;
  FusedPackArgs fa;
  for (int i = 0; i < n; ++i) {
    m[i].nblocks = gx * nbatch;                       // partial count = blocks per operand
    a[i].amax = m[i].parts;
    a[i].namax = gx * nbatch;
    fa.a[i] = a[i];
    fa.m[i] = m[i];
  }
  fa.sync = sync;
  const dim3 grid(gx, n, nbatch);
  if (NP == 1) hipLaunchKernelGGL((pack_kc_fused_kernel<1>), grid, dim3(256), 0, s, fa);
  else if (NP == 2) hipLaunchKernelGGL((pack_kc_fused_kernel<2>), grid, dim3(256), 0, s, fa);
  else hipLaunchKernelGGL((pack_kc_fused_kernel<3>), grid, dim3(256), 0, s, fa);
  return true;
}


void dispatch_pack_multi(int NP, const PackArgs4& args, int n, hipStream_t s, int nbatch = 1) {
  int gx = 1;
  for (int i = 0; i < n; ++i) gx = kc_blocks(args.a[i]) > gx ? kc_blocks(args.a[i]) : gx;
  const dim3 grid(gx, n, nbatch);
  if (NP == 1) hipLaunchKernelGGL((pack_kc_multi_kernel<1>), grid, dim3(256), 0, s, args);
  else if (NP == 2) hipLaunchKernelGGL((pack_kc_multi_kernel<2>), grid, dim3(256), 0, s, args);
  else hipLaunchKernelGGL((pack_kc_multi_kernel<3>), grid, dim3(256), 0, s, args);
}

void dispatch_pack_tr_multi(int NP, const PackArgs4& args, int n, hipStream_t s, int nbatch) {
  int gx = 1, gy = 1;
  for (int i = 0; i < n; ++i) {
    gx = (args.a[i].rows + 63) / 64 > gx ? (args.a[i].rows + 63) / 64 : gx;
    gy = (args.a[i].Kp + 63) / 64 > gy ? (args.a[i].Kp + 63) / 64 : gy;
  }
  const dim3 grid(gx, gy, n * nbatch);
  if (NP == 1) hipLaunchKernelGGL((pack_tr_multi_kernel<1>), grid, dim3(256), 0, s, args, nbatch);
  else if (NP == 2) hipLaunchKernelGGL((pack_tr_multi_kernel<2>), grid, dim3(256), 0, s, args, nbatch);
  else hipLaunchKernelGGL((pack_tr_multi_kernel<3>), grid, dim3(256), 0, s, args, nbatch);
}

void dispatch_pack(int NP, const PackArgs& a, bool tr, int nbatch, hipStream_t s) {
  if (NP == 1) launch_pack<1>(a, tr, nbatch, s);
  else if (NP == 2) launch_pack<2>(a, tr, nbatch, s);
  else launch_pack<3>(a, tr, nbatch, s);
}


}  // namespace
