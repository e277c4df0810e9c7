// Label assignment + losses of the MQ heads as two launches forward and two backward
// (MQ/libs/modeling/meta_archs.py: label_points_single_video :1253-1344, losses :1374-1524; losses.py:5-52 focal,
// :109-168 DIoU), instead of ~900 small tensor-expression kernels per step:
//
//   per (clip b, pyramid row r), one thread:
//     all-pairs point/GT assignment (centre sampling radius, regression range, shortest GT wins, ties within 1e-3
//     make the class target multi-hot), gaussian point weights from the learnable mu/sigma[ncls], sigmoid focal loss
//     over the classes (label smoothing), DIoU on positives (on relu(Scale_l * raw offsets), the regression head's last
//     two element-wise layers, meta_archs.py:344-346), row softmax for the "al" term (max_t softmax_c vs clip labels,
//     :1436-1446) whose per-(b, class) maximum is a packed 64-bit atomicMax;
//   then one block: #positives -> loss_normalizer EMA (:1407-1410, stays on the device), the al BCE, the four scalars.
// Backward recomputes the assignment (cheap: N GT per clip) and writes d logits, d raw offsets, and accumulates the few
// learnable-scalar gradients (mu/sigma x3, level scales) with atomics.
// Rows can be the plain concatenation of the levels or the LevelCat layout with separator rows (stride <= 0).
#include "common.h"

namespace {

constexpr int LT = 256;
constexpr int MAXC = 128;            // classes (22 at the first task, 110 after the last Ego4D task)

struct LossArgs {
  const float* logits; const float* offsets; const float* level_scale; const float* points;
  const int* row_level; const int* row_pos; const int* level_len; const float* gt; const float* gauss;
  int B, R, C, L, Nmax;
  float radius, smoothing;
};

struct Assign {
  bool gap, valid, pos;
  int best;                // argmin GT (first minimum), 0 when nothing matches (as torch.min over all-inf)
  unsigned bits[MAXC / 32];
  float t, stride, left, right, len;   // of the `best` GT
};

// gt block of clip b: [Nmax][2] segments, [Nmax] labels (as floats), [1] count
__device__ __forceinline__ const float* gt_of(const LossArgs& a, int b) { return a.gt + (long)b * (3 * a.Nmax + 1); }

__device__ __forceinline__ void assign_row(const LossArgs& a, int b, int r, Assign& s) {
  const float4 pt = *reinterpret_cast<const float4*>(a.points + 4L * r);
  s.gap = pt.w <= 0.f;
  s.valid = !s.gap && a.row_pos[r] < a.level_len[b * a.L + a.row_level[r]];
  s.pos = false;
  s.best = 0;
  s.t = pt.x; s.stride = pt.w;
#pragma unroll
  for (int i = 0; i < MAXC / 32; ++i) s.bits[i] = 0u;
  const float* g = gt_of(a, b);
  const int n = (int)g[3 * a.Nmax];
  float minlen = INFINITY;
  for (int j = 0; j < n; ++j) {
    const float s0 = g[2 * j], s1 = g[2 * j + 1];
    const float left = pt.x - s0, right = s1 - pt.x;
    bool inside;
    if (a.radius > 0.f) {
      const float ctr = 0.5f * (s0 + s1);
      const float lo = pt.x - fmaxf(ctr - pt.w * a.radius, s0);
      const float hi = fminf(ctr + pt.w * a.radius, s1) - pt.x;
      inside = fminf(lo, hi) > 0.f;
    } else {
      inside = fminf(left, right) > 0.f;
    }
    const float far = fmaxf(left, right);
    const bool ok = inside && far >= pt.y && far <= pt.z;
    const float len = ok ? (s1 - s0) : INFINITY;
    if (len < minlen) { minlen = len; s.best = j; }
  }
  if (minlen < INFINITY) {
    for (int j = 0; j < n; ++j) {          // every GT within 1e-3 of the shortest contributes its class (:1322-1331)
      const float s0 = g[2 * j], s1 = g[2 * j + 1];
      const float left = pt.x - s0, right = s1 - pt.x;
      bool inside;
      if (a.radius > 0.f) {
        const float ctr = 0.5f * (s0 + s1);
        inside = fminf(pt.x - fmaxf(ctr - pt.w * a.radius, s0), fminf(ctr + pt.w * a.radius, s1) - pt.x) > 0.f;
      } else {
        inside = fminf(left, right) > 0.f;
      }
      const float far = fmaxf(left, right);
      if (inside && far >= pt.y && far <= pt.z && (s1 - s0) <= minlen + 1e-3f) {
        const int c = (int)g[2 * a.Nmax + j];
        if (c >= 0 && c < a.C) s.bits[c >> 5] |= 1u << (c & 31);
      }
    }
    s.pos = s.valid;
  }
  const float s0 = n > 0 ? g[2 * s.best] : 0.f, s1 = n > 0 ? g[2 * s.best + 1] : 1.f;
  s.left = pt.x - s0; s.right = s1 - pt.x; s.len = s1 - s0;
}

// gaussian point weight exp(-(rel - mu)^2 / (2 sigma^2)) and rel - mu
__device__ __forceinline__ float gauss_w(float rel, float mu, float sg) {
  const float d = rel - mu;
  return expf(-(d * d) / (2.f * sg * sg));
}

__device__ __forceinline__ float focal_elem(float x, float t, float* dx) {
  const float p = 1.f / (1.f + expf(-x));
  const float ce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
  const float q = p + t - 2.f * p * t;                      // 1 - p_t
  const float al = 0.25f * t + 0.75f * (1.f - t);
  if (dx) *dx = al * ((p - t) * q * q + ce * 2.f * q * (1.f - 2.f * t) * p * (1.f - p));
  return al * ce * q * q;
}

// DIoU of non-negative (left, right) distances (losses.py:109-168); gradient wrt the prediction
__device__ __forceinline__ float diou_elem(float lp, float rp, float lg, float rg, float* dlp, float* drp) {
  const float eps = 1e-8f;
  const float inter = fminf(rp, rg) + fminf(lp, lg);
  const float uni = (lp + rp) + (lg + rg) - inter;
  const float uc = fmaxf(uni, eps);
  const float iou = inter / uc;
  const float enc = fmaxf(lp, lg) + fmaxf(rp, rg);
  const float ec = fmaxf(enc, eps);
  const float rho = 0.5f * (rp - lp - rg + lg);
  const float z = rho / ec;
  if (dlp) {
    const float di_l = lp <= lg ? 1.f : 0.f, di_r = rp <= rg ? 1.f : 0.f;          // d inter
    const float du_l = uni > eps ? 1.f - di_l : 0.f, du_r = uni > eps ? 1.f - di_r : 0.f;   // d clamp(union)
    const float de_l = enc > eps ? (lp >= lg ? 1.f : 0.f) : 0.f, de_r = enc > eps ? (rp >= rg ? 1.f : 0.f) : 0.f;
    *dlp = -(di_l * uc - inter * du_l) / (uc * uc) + 2.f * z * ((-0.5f) * ec - rho * de_l) / (ec * ec);
    *drp = -(di_r * uc - inter * du_r) / (uc * uc) + 2.f * z * (0.5f * ec - rho * de_r) / (ec * ec);
  }
  return 1.f - iou + z * z;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// forward, one thread per (b, r).  partial[(b * gridDim.x + bx) * 3 + {0,1,2}] = cls sum, reg sum, #pos
__global__ __launch_bounds__(LT) void loss_fwd_kernel(LossArgs a, float* __restrict__ partial,
                                                      unsigned long long* __restrict__ almax, int use_al) {
  __shared__ float red[4];
  __shared__ unsigned long long smax[MAXC];
  const int b = blockIdx.y, r = blockIdx.x * LT + threadIdx.x;
  if (use_al) for (int c = threadIdx.x; c < a.C; c += LT) smax[c] = 0ull;
  __syncthreads();
  float cls = 0.f, reg = 0.f, npos = 0.f;
  if (r < a.R) {
    Assign s;
    assign_row(a, b, r, s);
    const float* x = a.logits + ((long)b * a.R + r) * a.C;
    if (s.valid) {
      const float* g = gt_of(a, b);
      float wc = 1.f;
      if (s.pos) {
        const int c = (int)g[2 * a.Nmax + s.best];
        const float rel = ((s.right - s.left) * 0.5f) / (s.stride * s.len);
        wc = gauss_w(rel, a.gauss[c], a.gauss[a.C + c]);
        const float wl = gauss_w(rel, a.gauss[2 * a.C + c], a.gauss[3 * a.C + c]);
        const float wr = gauss_w(rel, a.gauss[4 * a.C + c], a.gauss[5 * a.C + c]);
        const float* o = a.offsets + ((long)b * a.R + r) * 2;
        const float sc = a.level_scale ? a.level_scale[a.row_level[r]] : 1.f;
        const float lp = a.level_scale ? fmaxf(sc * o[0], 0.f) : o[0], rp = a.level_scale ? fmaxf(sc * o[1], 0.f) : o[1];
        reg = diou_elem(lp, rp, s.left / s.stride, s.right / s.stride, nullptr, nullptr) * (0.5f * (wl + wr)) * wc;
        npos = 1.f;
      }
      float f = 0.f;
      for (int c = 0; c < a.C; ++c) {
        const float t = ((s.bits[c >> 5] >> (c & 31)) & 1u) ? 1.f : 0.f;
        f += focal_elem(x[c], t * (1.f - a.smoothing) + a.smoothing / (a.C + 1), nullptr);
      }
      cls = f * wc;
    }
    if (use_al && !s.gap) {
      // score = softmax_c(logits masked_fill(~valid, -1e7)): a masked row is uniform 1/C (and carries no gradient)
      float m = -INFINITY, z = 0.f;
      if (s.valid) {
        for (int c = 0; c < a.C; ++c) m = fmaxf(m, x[c]);
        for (int c = 0; c < a.C; ++c) z += expf(x[c] - m);
      }
      for (int c = 0; c < a.C; ++c) {
        const float p = s.valid ? expf(x[c] - m) / z : 1.f / a.C;
        // first maximum wins on ties: larger packed value = larger score, then SMALLER row
        const unsigned long long pk = ((unsigned long long)__float_as_uint(p) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)r);
        atomicMax(&smax[c], pk);
      }
    }
  }
  const float c0 = block_sum(cls, red), c1 = block_sum(reg, red), c2 = block_sum(npos, red);
  if (threadIdx.x == 0) {
    float* p = partial + ((long)b * gridDim.x + blockIdx.x) * 3;
    p[0] = c0; p[1] = c1; p[2] = c2;
  }
  if (use_al) {
    __syncthreads();
    for (int c = threadIdx.x; c < a.C; c += LT) atomicMax(&almax[(long)b * a.C + c], smax[c]);
  }
}

// one block: out[0..3] = cls, reg, al, final; saved[0] = the normaliser this step divides by; al_row[b*C+c] = argmax row
// (or -1), al_score likewise; the packed-maximum workspace is returned to zero for the next call.
__global__ __launch_bounds__(LT) void loss_finish_kernel(LossArgs a, const float* __restrict__ partial, int nparts,
                                                         unsigned long long* __restrict__ almax, int use_al,
                                                         float* __restrict__ loss_norm, float momentum, float loss_weight,
                                                         float al_weight, float* __restrict__ out,
                                                         float* __restrict__ saved, int* __restrict__ al_row,
                                                         float* __restrict__ al_score) {
  __shared__ double red[4][LT];
  double s0 = 0, s1 = 0, s2 = 0, al = 0;
  for (int i = threadIdx.x; i < nparts; i += LT) { s0 += partial[3 * i]; s1 += partial[3 * i + 1]; s2 += partial[3 * i + 2]; }
  if (use_al) {
    for (int i = threadIdx.x; i < a.B * a.C; i += LT) {
      const int b = i / a.C, c = i % a.C;
      const unsigned long long pk = almax[i];
      almax[i] = 0ull;
      const float sc = __uint_as_float((unsigned)(pk >> 32));
      al_row[i] = pk ? (int)(0xFFFFFFFFu - (unsigned)(pk & 0xFFFFFFFFull)) : -1;
      al_score[i] = sc;
      const float* g = gt_of(a, b);
      const int n = (int)g[3 * a.Nmax];
      bool inv = false;
      for (int j = 0; j < n; ++j) inv = inv || ((int)g[2 * a.Nmax + j] == c);
      al += inv ? -logf(sc) : -logf(1.f - sc);
    }
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2; red[3][threadIdx.x] = al;
  __syncthreads();
  for (int o = LT / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float num_pos = (float)red[2][0];
    const float norm = momentum * loss_norm[0] + (1.f - momentum) * fmaxf(num_pos, 1.f);
    const float cls_l = (float)(red[0][0] / (double)norm), reg_l = (float)(red[1][0] / (double)norm);
    const float al_l = use_al ? (float)(red[3][0] / (double)norm) : 0.f;
    out[0] = cls_l; out[1] = reg_l; out[2] = al_l;
    out[3] = cls_l + reg_l * loss_weight + al_l * al_weight;
    saved[0] = norm;
    loss_norm[0] = norm;
  }
}

// backward, one thread per (b, r): wts = {g_cls, g_reg, g_al} effective weights of the three UNNORMALISED sums
struct GradOut { const float* g[4]; };     // upstream gradients of {cls, reg, al, final}; null = 0

__global__ __launch_bounds__(LT) void loss_bwd_kernel(LossArgs a, GradOut go, float loss_weight,
                                                      float al_weight, const float* __restrict__ saved,
                                                      const int* __restrict__ al_row, const float* __restrict__ al_score,
                                                      int use_al, float* __restrict__ dlogits, float* __restrict__ doffsets,
                                                      float* __restrict__ contrib) {
  // contrib[(b R + r) * 8 ..]: this row's share of the PARAMETER gradients -- {key, d level_scale, 6 x d gauss} with key =
  // 256 * level + class for a positive row, -1 otherwise -- summed in a fixed order by loss_bwd_finish_kernel (round 6: these
  // were float atomicAdds, the only sums of the step whose order changed from launch to launch)
  const int b = blockIdx.y, r = blockIdx.x * LT + threadIdx.x;
  if (r >= a.R) return;
  float* crow = contrib + ((long)b * a.R + r) * 8;
  crow[0] = -1.f;
  const float inv_norm = 1.f / saved[0];
  const float gf = go.g[3] ? go.g[3][0] : 0.f;
  const float g_cls = ((go.g[0] ? go.g[0][0] : 0.f) + gf) * inv_norm;
  const float g_reg = ((go.g[1] ? go.g[1][0] : 0.f) + gf * loss_weight) * inv_norm;
  const float g_al = ((go.g[2] ? go.g[2][0] : 0.f) + gf * al_weight) * inv_norm;
  Assign s;
  assign_row(a, b, r, s);
  const long base = ((long)b * a.R + r);
  const float* x = a.logits + base * a.C;
  float* dx = dlogits + base * a.C;
  float* dofs = doffsets + base * 2;
  dofs[0] = 0.f; dofs[1] = 0.f;
  if (!s.valid) {
    for (int c = 0; c < a.C; ++c) dx[c] = 0.f;
    return;
  }
  const float* g = gt_of(a, b);
  float wc = 1.f, fsum = 0.f;
  int cg = 0;
  float rel = 0.f;
  if (s.pos) {
    cg = (int)g[2 * a.Nmax + s.best];
    rel = ((s.right - s.left) * 0.5f) / (s.stride * s.len);
    wc = gauss_w(rel, a.gauss[cg], a.gauss[a.C + cg]);
  }
  // al term: rows that hold the per-class maximum get the softmax Jacobian of that class
  float m = -INFINITY, z = 1.f, alsum = 0.f;       // alsum = sum_c coef_c * p_c over the classes this row wins
  bool any_al = false;
  if (use_al) {
    for (int c = 0; c < a.C; ++c) any_al = any_al || (al_row[b * a.C + c] == r);
    if (any_al) {
      z = 0.f;
      for (int c = 0; c < a.C; ++c) m = fmaxf(m, x[c]);
      for (int c = 0; c < a.C; ++c) z += expf(x[c] - m);
      const int n = (int)g[3 * a.Nmax];
      for (int c = 0; c < a.C; ++c)
        if (al_row[b * a.C + c] == r) {
          bool inv = false;
          for (int j = 0; j < n; ++j) inv = inv || ((int)g[2 * a.Nmax + j] == c);
          const float sc = al_score[b * a.C + c];
          alsum += (inv ? -1.f / sc : 1.f / (1.f - sc)) * (expf(x[c] - m) / z);
        }
    }
  }
  const int n = (int)g[3 * a.Nmax];
  for (int c = 0; c < a.C; ++c) {
    const float t = ((s.bits[c >> 5] >> (c & 31)) & 1u) ? 1.f : 0.f;
    float d;
    fsum += focal_elem(x[c], t * (1.f - a.smoothing) + a.smoothing / (a.C + 1), &d);
    float v = g_cls * wc * d;
    if (any_al) {
      const float p = expf(x[c] - m) / z;
      float coef = 0.f;
      if (al_row[b * a.C + c] == r) {
        bool inv = false;
        for (int j = 0; j < n; ++j) inv = inv || ((int)g[2 * a.Nmax + j] == c);
        const float sc = al_score[b * a.C + c];
        coef = inv ? -1.f / sc : 1.f / (1.f - sc);
      }
      v += g_al * p * (coef - alsum);            // sum_k coef_k p_k (delta_kc - p_c)
    }
    dx[c] = v;
  }
  if (s.pos) {
    const float mu_l = a.gauss[2 * a.C + cg], sg_l = a.gauss[3 * a.C + cg];
    const float mu_r = a.gauss[4 * a.C + cg], sg_r = a.gauss[5 * a.C + cg];
    const float mu_c = a.gauss[cg], sg_c = a.gauss[a.C + cg];
    const float wl = gauss_w(rel, mu_l, sg_l), wr = gauss_w(rel, mu_r, sg_r);
    const float* o = a.offsets + base * 2;
    const int lvl = a.row_level[r];
    const float sc = a.level_scale ? a.level_scale[lvl] : 1.f;
    const float lp = a.level_scale ? fmaxf(sc * o[0], 0.f) : o[0], rp = a.level_scale ? fmaxf(sc * o[1], 0.f) : o[1];
    float dl, dr;
    const float di = diou_elem(lp, rp, s.left / s.stride, s.right / s.stride, &dl, &dr);
    const float wlr = 0.5f * (wl + wr);
    const float gl = g_reg * wlr * wc * dl, gr = g_reg * wlr * wc * dr;
    if (a.level_scale) {
      const float ml = sc * o[0] > 0.f ? 1.f : 0.f, mr = sc * o[1] > 0.f ? 1.f : 0.f;
      dofs[0] = gl * ml * sc; dofs[1] = gr * mr * sc;
      crow[1] = gl * ml * o[0] + gr * mr * o[1];
    } else {
      dofs[0] = gl; dofs[1] = gr;
      crow[1] = 0.f;
    }
    // d w_cls = cls: g_cls * focal sum ; reg: g_reg * diou * wlr.   dw/dmu = w (rel - mu) / s^2, dw/ds = w (rel - mu)^2 / s^3
    const float dwc = g_cls * fsum + g_reg * di * wlr;
    const float dwl = g_reg * di * wc * 0.5f;
    const float dc = rel - mu_c, dL = rel - mu_l, dR = rel - mu_r;
    crow[2] = dwc * wc * dc / (sg_c * sg_c);
    crow[3] = dwc * wc * dc * dc / (sg_c * sg_c * sg_c);
    crow[4] = dwl * wl * dL / (sg_l * sg_l);
    crow[5] = dwl * wl * dL * dL / (sg_l * sg_l * sg_l);
    crow[6] = dwl * wr * dR / (sg_r * sg_r);
    crow[7] = dwl * wr * dR * dR / (sg_r * sg_r * sg_r);
    crow[0] = (float)(256 * lvl + cg);
  }
}

// One workgroup per class c (blockIdx.x < C: the six gaussian-weight gradients of class c) or per pyramid level (blockIdx.x - C:
// the gradient of that level's regression scale): thread t adds the matching rows t, t + 256, ... in increasing order, the 256
// partial sums are folded by a fixed tree -- the same bits on every launch.
__global__ __launch_bounds__(LT) void loss_bwd_finish_kernel(const float* __restrict__ contrib, long rows, int C, int L,
                                                             float* __restrict__ dscale, float* __restrict__ dgauss) {
  __shared__ float red[6][LT];
  const int blk = blockIdx.x, t = threadIdx.x;
  const bool cls = blk < C;
  const int want = cls ? blk : blk - C;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long i = t; i < rows; i += LT) {
    const float key = contrib[i * 8];
    if (key < 0.f) continue;
    const int k = (int)key;
    if (cls) {
      if ((k & 255) == want) {
        const float4 lo = *reinterpret_cast<const float4*>(contrib + i * 8 + 4);
        acc[0] += contrib[i * 8 + 2]; acc[1] += contrib[i * 8 + 3];
        acc[2] += lo.x; acc[3] += lo.y; acc[4] += lo.z; acc[5] += lo.w;
      }
    } else if ((k >> 8) == want) {
      acc[0] += contrib[i * 8 + 1];
    }
  }
  const int nv = cls ? 6 : 1;
  for (int v = 0; v < nv; ++v) red[v][t] = acc[v];
  __syncthreads();
  for (int w = LT / 2; w > 0; w >>= 1) {
    if (t < w)
      for (int v = 0; v < nv; ++v) red[v][t] += red[v][t + w];
    __syncthreads();
  }
  if (t == 0) {
    if (cls) { for (int v = 0; v < 6; ++v) dgauss[v * C + want] = red[v][0]; }
    else if (dscale) dscale[want] = red[0][0];
  }
}

LossArgs make_args(const vilco_loss_desc* d) {
  LossArgs a;
  a.logits = d->logits; a.offsets = d->offsets; a.level_scale = d->level_scale; a.points = d->points;
  a.row_level = d->row_level; a.row_pos = d->row_pos; a.level_len = d->level_len; a.gt = d->gt; a.gauss = d->gauss;
  a.B = d->B; a.R = d->R; a.C = d->C; a.L = d->L; a.Nmax = d->Nmax;
  a.radius = d->center_radius; a.smoothing = d->label_smoothing;
  return a;
}

bool bad_desc(const vilco_loss_desc* d) {
  return !d || !d->logits || !d->offsets || !d->points || !d->row_level || !d->row_pos || !d->level_len || !d->gt ||
         !d->gauss || !d->loss_norm || d->B < 1 || d->R < 1 || d->C < 1 || d->L < 1 || d->Nmax < 0;
}

}  // namespace

extern "C" size_t vilco_mq_loss_workspace(int32_t B, int32_t R, int32_t C) {
  // [partials: B * ceil(R/256) * 3 floats][al rows: B*C int][al scores: B*C float][contrib: B*R*8 floats], 16-byte aligned pieces
  const size_t parts = (size_t)B * ((R + LT - 1) / LT) * 3 * sizeof(float);
  return ((parts + 15) / 16) * 16 + 2 * (((size_t)B * C * 4 + 15) / 16) * 16 + (size_t)B * R * 8 * sizeof(float);      // + backward's per-row parameter-gradient shares
}

extern "C" int vilco_mq_loss_fwd(const vilco_loss_desc* d, float* out, float* saved, void* almax_state, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (bad_desc(d) || !out || !saved || !almax_state || !workspace) return VILCO_ERR_BADARG;
  if (d->C > MAXC) return VILCO_ERR_UNSUPPORTED;
  if (workspace_bytes < vilco_mq_loss_workspace(d->B, d->R, d->C)) return VILCO_ERR_WORKSPACE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  LossArgs a = make_args(d);
  const int gx = (d->R + LT - 1) / LT;
  float* partial = reinterpret_cast<float*>(workspace);
  const size_t parts = (((size_t)d->B * gx * 3 * sizeof(float)) + 15) / 16 * 16;
  int* al_row = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + parts);
  float* al_score = reinterpret_cast<float*>(reinterpret_cast<char*>(al_row) + (((size_t)d->B * d->C * 4 + 15) / 16) * 16);
  unsigned long long* almax = reinterpret_cast<unsigned long long*>(almax_state);
  const int use_al = d->use_al ? 1 : 0;
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(gx, d->B), dim3(LT), 0, s, a, partial, almax, use_al);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(LT), 0, s, a, partial, d->B * gx, almax, use_al, d->loss_norm,
                     d->momentum, d->loss_weight, d->al_weight, out, saved, al_row, al_score);
  return vilco_launch_status();
}

extern "C" int vilco_mq_loss_bwd(const vilco_loss_desc* d, const float* g_cls, const float* g_reg, const float* g_al,
                                 const float* g_final, const float* saved, const void* workspace, float* d_logits,
                                 float* d_offsets, float* d_level_scale, float* d_gauss, void* stream) {
  if (bad_desc(d) || !saved || !workspace || !d_logits || !d_offsets || !d_gauss) return VILCO_ERR_BADARG;
  GradOut grad_out{{g_cls, g_reg, g_al, g_final}};
  if (d->level_scale && !d_level_scale) return VILCO_ERR_BADARG;
  if (d->C > MAXC) return VILCO_ERR_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  LossArgs a = make_args(d);
  const int gx = (d->R + LT - 1) / LT;
  const size_t parts = (((size_t)d->B * gx * 3 * sizeof(float)) + 15) / 16 * 16;
  const int* al_row = reinterpret_cast<const int*>(reinterpret_cast<const char*>(workspace) + parts);
  const float* al_score = reinterpret_cast<const float*>(reinterpret_cast<const char*>(al_row) + (((size_t)d->B * d->C * 4 + 15) / 16) * 16);
  // (the workspace is the forward's: its last region is written here -- the `const` of the signature predates it)
  float* contrib = const_cast<float*>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(al_score) + (((size_t)d->B * d->C * 4 + 15) / 16) * 16));
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(gx, d->B), dim3(LT), 0, s, a, grad_out, d->loss_weight, d->al_weight, saved,
                     al_row, al_score, d->use_al ? 1 : 0, d_logits, d_offsets, contrib);
  hipLaunchKernelGGL(loss_bwd_finish_kernel, dim3(d->C + (d->level_scale ? d->L : 0)), dim3(LT), 0, s, contrib, (long)d->B * d->R, d->C,
                     d->L, d->level_scale ? d_level_scale : nullptr, d_gauss);
  return vilco_launch_status();
}
