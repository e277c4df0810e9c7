/*
 * vilco_hip.h -- C ABI of libvilco_hip.so, the MI355X (gfx950) kernel library behind the
 * drop-in MQ modules in vilco_amd/modeling.
 *
 * The reference (cruiseresearchgroup/ViLCo, MQ tree) has no C operator API for this path: its
 * operators are torch.nn.functional calls inside MQ/libs/modeling (python) plus ONE compiled
 * extension, nms_1d_cpu (MQ/libs/utils/csrc/nms_cpu.cpp:172-182).  Each entry point below cites
 * the reference code whose arithmetic it replaces.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer owned by the caller
 *    (PyTorch caching allocator) unless stated; `stream` is a hipStream_t passed as void*.
 *  - activations are TOKEN-MAJOR fp32: x[b][t][c] (c contiguous) -- the reference is channel-first
 *    [B,C,T] (blocks.py:107-108); vilco_transpose2d converts at the module boundary.
 *  - sequence masks are prefix masks (meta_archs.py:1175 `arange < len`), passed as int32 len[B].
 *  - functions never allocate, never synchronise, never throw; they return 0 or a negative status.
 */
#ifndef VILCO_HIP_H
#define VILCO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum vilco_status {
  VILCO_OK = 0,
  VILCO_ERR_BADARG = -1,      /* null pointer / negative size / misaligned operand */
  VILCO_ERR_UNSUPPORTED = -2, /* shape outside what the kernels implement */
  VILCO_ERR_LAUNCH = -3,      /* hipLaunchKernel reported an error */
  VILCO_ERR_WORKSPACE = -4    /* workspace too small */
};

const char* vilco_status_str(int status);
/* library / target identification: "vilco_hip <ver> gfx950" */
const char* vilco_version(void);
/* With VILCO_GRID_SYNC=1 in the environment two-stage reductions (column sums, amax -> pack) finish inside one      */
/* launch through a grid-wide barrier whose spin is bounded; this returns how many barriers gave up since the library */
/* was loaded (0 in a healthy process).  Synchronises the device.  Default: the two-launch forms (measured faster).   */
int vilco_sync_timeouts_read(void);

/* ------------------------------------------------------------------------------------------ */
/* GEMM family: every 1x1 conv, k=3 conv, nn.Linear, einsum projection and (round 1) the        */
/* attention contractions.  C[m][n] = epilogue(alpha * sum_k A(m,k) B(k,n)).                    */
/* Replaces aten::convolution / addmm / bmm / einsum under blocks.py:79,217-226,340-349,420-435, */
/* 533-539, meta_archs.py:216-235,309-331, modeling_xlnet_x.py:284-325,437-443,474-489.         */
/* Operands are split once into 16-bit planes in their natural row-major layout (vilco_pack; the same planes serve    */
/* the k-contiguous and the k-major use, the latter through transposing LDS reads), then ONE MFMA kernel runs on the  */
/* planes: 256 / 192 / 128 x 128 x 32 tiles chosen per problem, 8 waves in a ping-pong schedule, optional split-K.    */
/* Default precision 3: two fp16 parts of operands scaled by a per-tensor power of two (v_mfma_f32_16x16x32_f16,     */
/* 3 MFMAs per product, ~2^-22, fp32 accumulate).  Others: 0 = split-bf16 (hi+lo, 3 MFMAs, ~2^-17 relative),           */
/* 1 = single bf16 pass, 2 = three-part bf16 split (6 MFMAs, ~2^-25: numerically an fp32 GEMM, 8 exponent bits per    */
/* element -- the format of call sites whose tensors span too much range for one scale);  4 = the fp16 x2 planes, but the */
/* product takes their leading parts only (1 MFMA, 11-bit operands, fp32 accumulate): for the weight-  */
/* gradient products dW = dY^T X, whose rounding errors average over the B*T contraction and feed   */
/* nothing downstream (ops.py: dw_precision).                                                        */
/* ------------------------------------------------------------------------------------------ */
enum { VILCO_ACT_NONE = 0, VILCO_ACT_RELU = 1, VILCO_ACT_GELU = 2 };
enum { VILCO_TAP_NONE = 0, VILCO_TAP_A = 1, VILCO_TAP_B = 2 };

typedef struct vilco_gemm_desc {
  const float* A;
  const float* B;
  float* C;
  int32_t M, N, K;
  int32_t a_kcontig; /* 1: A(m,k) = A[m*lda + k];  0: A(m,k) = A[k*lda + m] */
  int32_t b_kcontig; /* 1: B(k,n) = B[n*ldb + k];  0: B(k,n) = B[k*ldb + n] */
  int64_t lda, ldb, ldc;
  /* batch z = zo*batch_inner + zi ; pointer offset = zo*s?o + zi*s?i (elements) */
  int32_t batch_outer, batch_inner;
  int64_t sAo, sAi, sBo, sBi, sCo, sCi;
  /* k=3 "same" conv as a GEMM over overlapped token rows (MaskedConv1D, blocks.py:79,114):
   * the tapped operand's contiguous dim spans 3*tapC (taps t-1,t,t+1), its row index is a token
   * (b*tapT + t); element address = base + row*ld + col - tapC, zero outside [0,tapT). */
  int32_t tap_operand, tapC, tapT;
  int32_t precision;
  float alpha, beta; /* result += beta * C_old (after the whole epilogue) */
  /* epilogue, applied in this order; each optional (null / 0 = off) */
  const float* bias;      /* [N]                                       */
  float* preact;          /* same layout as C: value before activation */
  int32_t act;            /* VILCO_ACT_*                               */
  const int32_t* row_len; /* row m -> b = m / rowT, t = m % rowT; zero the row when t >= row_len[b] */
  int32_t rowT;
  const float* colscale;  /* [N]  (AffineDropPath.scale, blocks.py:663-670) */
  const float* residual;  /* same layout as C; added after scaling */
  int32_t res_masked;     /* 1: residual is also zeroed on masked rows */
  /* device scratch for the bf16 operand planes and split-K partials; size from vilco_gemm_workspace() */
  void* workspace;
  size_t workspace_bytes;
  /* optional operands already packed by vilco_pack (NULL: the call packs A / B itself).  The packed tensor is the  */
  /* row-major matrix the operand lives in: [M][K] (a_kcontig = 1) or [K][M] (a_kcontig = 0), likewise [N][K] /    */
  /* [K][N] for B -- ONE pack of an activation, a weight or an output gradient serves every product it appears in  */
  /* (forward, dX = dY W and dW = dY^T X).  For tap_operand = NONE and batch 1 (A / B may then be NULL), and for    */
  /* the weight operand B ([N][3*tapC], k-contiguous) of a conv whose taps are on A (tap_operand = TAP_A).          */
  const void* a_planes;
  const void* b_planes;
  /* XLNet relative-position band (modeling_xlnet_x.py:204-214, 284-325): with T = bandT only the entries          */
  /* 0 <= p - T + i < T of the [T, 2T] position-score matrix (row i, column p) are ever used / non-zero.            */
  /*   band 1: C is that matrix (M = T rows i, N = 2T columns p): output tiles wholly outside the band are skipped  */
  /*           (left unwritten);  band 2: A is that matrix, k = p (M rows i): K-steps outside a tile's band are     */
  /*           skipped;  band 3: A is its transpose, M rows p, k = i: likewise.  0: dense.                          */
  int32_t band, bandT;
  /* fused inverted dropout on the stored output (after activation, row mask, column scale; before residual / beta):   */
  /* the mask of vilco_dropout(p, seed) at element index m*N + n.  Needs ldc == N, batch 1.  0: none.                     */
  float drop_p;
  uint32_t drop_seed;
  /* optional: the call leaves vilco_gemm_amax_parts(desc) partial maxima of |C| (as stored, after the whole epilogue)  */
  /* here -- one per workgroup of the kernel that writes C -- for the operand pack of the next product                  */
  /* (vilco_pack_item.amax), which then needs no pass of its own over C.  NULL: not wanted.                             */
  float* amax_out;
  /* optional (precision 3, operands the call packs itself -- A / B given as fp32): max|x| partials of the WHOLE operand  */
  /* tensor already on the device, left by the kernel that produced it; the call then skips its amax pass over it.      */
  const float* a_amax; int32_t a_namax;
  const float* b_amax; int32_t b_namax;
  /* 1: a_planes / b_planes hold the k=3 convs' zero-padded per-sequence image (vilco_pack_item.seq_len = tapT) instead of */
  /* the natural [rows32][cols32] layout.  Required for -- and only legal with -- the tapped operand of a conv product:     */
  /* tap_operand = TAP_A: a_planes (tapC % 8 == 0);  TAP_B (the weight gradient, M % 8 == 0, tapC % 8 == 0): BOTH operands, */
  /* a_planes = the image of dZ [K rows of width M], b_planes = the image of the conv input [K rows of width tapC].          */
  int32_t a_planes_seq, b_planes_seq;
  /* optional: one float per output row m (M floats, one batch element only): rows with row_mask[m] == 0 are zeroed exactly as */
  /* rows beyond row_len are -- a validity pattern that is not "the first len rows of every sequence" (the heads over the        */
  /* concatenated pyramid levels: per level, per clip; meta_archs.py:216-235 applies mask[level] after every conv)               */
  const float* row_mask;
} vilco_gemm_desc;

size_t vilco_gemm_workspace(const vilco_gemm_desc* d);
int vilco_gemm(const vilco_gemm_desc* d, void* stream);
/* n (1..4) INDEPENDENT products of one shape / orientation / format as ONE launch (round 5: the q / k / v projections of an
 * attention block -- MQ/libs/modeling/blocks.py:332-344 -- forward and dX; each alone is a 75 %-full round of tiles on 256 CUs).
 * Requires packed operands (a_planes / b_planes), precision 3, no batch / tap / band, equal M, N, K and orientations, and a
 * plan without split-K; anything else runs as n vilco_gemm calls in order.  Same arithmetic either way.  VILCO_GEMM_GROUP=0:
 * always ungrouped. */
int vilco_gemm_group(const vilco_gemm_desc* descs, int32_t n, void* stream);

/* Timing of the MFMA kernel alone (not the packs, not the split-K reduce): between begin and end every vilco_gemm   */
/* brackets its main kernel with HIP events on the caller's stream; end waits for them and returns the sum.          */
/* Tuning override: force the tile height (128 | 192 | 256; 0 = cost model) and split-K count (0 = heuristic) of every
 * following vilco_gemm in this process.  Initial values: environment VILCO_GEMM_BM / VILCO_GEMM_KS, read once. */
int vilco_gemm_force(int32_t bm, int32_t ks);
/* Split-K finish of every following vilco_gemm: 1 = inside the launch (the last-arriving split workgroup of a tile sums
 * the partial accumulators in split order and runs the epilogue), 0 = fp32 slabs + a reduce launch (default: measured
 * faster, gemm.hip).  Same summation order either way.  Initial value: environment VILCO_GEMM_FIXUP=1 enables. */
int vilco_gemm_set_fixup(int32_t on);
/* Main kernel of the fp16 x2 products (precision 3 / 4) of every following vilco_gemm: 1 = gemm_gl_kernel (round 5: 64-element
 * K chunks staged by LDS-DMA in whole 128-byte lines; default), 0 = gemm_pp_kernel (rounds 1-4: 32-element K-steps staged through
 * registers).  Initial value: environment VILCO_GEMM_GL=0 selects the old kernel. */
int vilco_gemm_set_gl(int32_t on);
/* Round 5: a two-part product of 257..512 192-row tiles (between one and two rounds on the 256 CUs) is launched as ONE full round */
/* of 192-row tiles over its first rows + one round of 128-row tiles over the rest (same arithmetic per tile, disjoint rows).   */
/* 0 (default: measured, no gain in the step) / 1; env VILCO_GEMM_TAIL128.                                                     */
int vilco_gemm_set_tail128(int32_t on);
/* Round 6: few-row NT products of the default precision (M <= 640 token rows: the pyramid levels at T' <= 288, the 77 text tokens, every
 * level of BASELINE configs[0]; reference: the nn.Linear / 1x1 MaskedConv1D calls of blocks.py:191-269 at those levels) run as ONE
 * launch of gemm_skinny_kernel (the eight waves of a workgroup split K, fragments straight from the operand planes, partial tiles
 * summed through LDS) instead of a split-K plan of the tiled kernel + its reduce launch.  1 (default) / 0; env VILCO_GEMM_SKINNY,
 * VILCO_GEMM_SKINNY_M = the largest M that takes it (640).  Same arithmetic per product (two fp16 parts, three MFMAs, fp32
 * accumulation); the summation order over K differs from the tiled kernels'. */
int vilco_gemm_set_skinny(int32_t on);
/* Generation of the process-wide GEMM configuration: bumped by vilco_gemm_force / _set_fixup / _set_gl / _set_tail128 / _set_skinny.  The host
 * side keys captured hipGraphs on it (a replay runs the plan that was recorded, not the current configuration). */
int64_t vilco_gemm_config_gen(void);
/* floats written to desc->amax_out by vilco_gemm(desc) (depends on the tile / split-K plan); 0: not available */
int32_t vilco_gemm_amax_parts(const vilco_gemm_desc* desc);
int vilco_gemm_profile_begin(void);
int vilco_gemm_profile_end(double* kernel_ms, int64_t* launches);
/* per-launch records of the last begin/end bracket (measurement tooling, tools/gemm_shapes.py): desc[i*10 + 0..9] =
 * M, N, K, batch, tile rows, split-K count, precision, a k-major, b k-major, tap flags; ms[i] = that launch's kernel
 * time.  Returns the number of records available (may exceed cap). */
int64_t vilco_gemm_profile_records(int64_t* desc, double* ms, int64_t cap);

/* Splits the fp32 row-major matrix src[rows][cols] (row stride ld) ONCE into the 16-bit operand planes of         */
/* `precision` ([part][rows32][cols32], zero padded; precision 3 also leaves the per-tensor power-of-two scale in  */
/* the buffer's header).  `planes` is device memory, 256-byte aligned, vilco_pack_bytes() long.                     */
size_t vilco_pack_bytes(int64_t rows, int64_t cols, int32_t precision);
int vilco_pack(const float* src, int64_t rows, int64_t cols, int64_t ld, int32_t precision, void* planes,
               size_t planes_bytes, void* stream);
/* up to four tensors in the same two launches (an activation and its layer's weight) */
typedef struct vilco_pack_item {
  const float* src;
  int64_t rows, cols, ld;
  void* planes;
  size_t planes_bytes;
  /* optional: `nbatch` matrices `batch_stride` floats apart (0 / 1 = one matrix); the planes are then            */
  /* [part][batch][rows32][cols32] and serve batched vilco_gemm calls whose A / B batches are numbered the same   */
  int32_t nbatch;
  int64_t batch_stride;
  /* relshift = 1: pack XLNet's unshifted view [rows][rows + cols] of src[rows][cols] (element (i,p) =            */
  /* src[i][p - rows + i] or 0): the adjoint of rel_shift_bnij, so dS feeds the position-term gradients directly  */
  int32_t relshift;
  /* optional (precision 3): `namax` partial maxima of |src| already on the device -- left by the kernel that produced  */
  /* src (vilco_layernorm_fwd_amax, vilco_act_bwd_amax, vilco_qkv_pre_fwd) -- so the pack needs no amax launch          */
  const float* amax;
  int32_t namax;
  /* seq_len = T > 0: src is rows / T token sequences of T rows (cols % 8 == 0) and the planes are the image the k=3 convs   */
  /* read -- every sequence with one zero row before and after it, [nseq * (T + 2) + zero rows][cols], no column padding:    */
  /* ONE pack of a conv's input x serves the forward product (vilco_gemm, tap_operand = TAP_A, a_planes) and the weight     */
  /* gradient (TAP_B, b_planes); one pack of dZ serves dX (TAP_A) and the weight gradient (TAP_B, a_planes).                 */
  /* Reference: the im2col-free form of F.conv1d under MaskedConv1D, blocks.py:106-130.                                      */
  int32_t seq_len;
} vilco_pack_item;
size_t vilco_pack_item_bytes(const vilco_pack_item* item, int32_t precision);   /* honours nbatch / relshift */
int vilco_pack_many(const vilco_pack_item* items, int32_t n, int32_t precision, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* LayerNorm over the channel dim of token-major rows: blocks.py:160-175 (biased variance, eps   */
/* inside sqrt) and the stock nn.LayerNorm calls (blocks.py:446-451, modeling_xlnet_x.py:236,473). */
/* relu=1 fuses the ReLU that follows every embed/head LN (backbones.py:219, meta_archs.py:270).  */
/* ------------------------------------------------------------------------------------------ */
int vilco_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                        float* mean, float* rstd, int64_t rows, int32_t C, float eps,
                        int32_t relu, void* stream);
/* the same, and the kernel also leaves *n_parts per-block partial maxima of |y| in amax_parts (device, >= 2048 floats): */
/* the operand pack of y (vilco_pack_item.amax) then needs no separate pass over it                                     */
int vilco_layernorm_fwd_amax(const float* x, const float* gamma, const float* beta, float* y,
                             float* mean, float* rstd, int64_t rows, int32_t C, float eps,
                             int32_t relu, float* amax_parts, int32_t* n_parts, void* stream);
/* the same, and y also written by the kernel as the fp16 x2 operand planes (precision 3) of the product that consumes it: */
/* seq_len = 0: vilco_pack's layout for [rows][C] (C % 32 == 0); seq_len = T > 0: the k=3 convs' zero-padded per-sequence    */
/* image (vilco_pack_item.seq_len; C % 8 == 0, rows % T == 0).  `planes`: device, 256-byte aligned,                           */
/* vilco_layernorm_planes_bytes() long (= vilco_pack_bytes / vilco_pack_item_bytes of the same tensor).  The planes' scale    */
/* comes from the bound max|gamma| sqrt(C) + max|beta| >= max|y| instead of the exact maximum: no pass over y, no pack launch. */
/* row_mask (optional, with or without planes): y[row][:] *= row_mask[row % mask_rows] -- the zero separator rows of the heads' */
/* concatenated pyramid levels (meta_archs.py:216-235 runs the shared head per level); with relu = 1 and a 0 / 1 mask          */
/* vilco_layernorm_bwd needs no mask of its own: it already drops the gradient wherever the saved y is 0.                        */
size_t vilco_layernorm_planes_bytes(int64_t rows, int32_t C, int32_t seq_len);
int vilco_layernorm_fwd_planes(const float* x, const float* gamma, const float* beta, float* y,
                               float* mean, float* rstd, int64_t rows, int32_t C, float eps,
                               int32_t relu, float* amax_parts, int32_t* n_parts, void* planes, size_t planes_bytes,
                               int32_t seq_len, const float* row_mask, int64_t mask_rows, void* stream);
size_t vilco_layernorm_bwd_workspace(int64_t rows, int32_t C);
/* y (forward output) is only read when relu=1.  dgamma/dbeta are overwritten. */
int vilco_layernorm_bwd(const float* dy, const float* x, const float* y, const float* gamma,
                        const float* mean, const float* rstd, float* dx, float* dgamma,
                        float* dbeta, int64_t rows, int32_t C, int32_t relu, void* workspace,
                        size_t workspace_bytes, void* stream);
/* dres (optional, [rows][C]): dx = LayerNorm backward + dres -- the gradient that reaches x over the residual connection  */
/* around the branch this LayerNorm opens (blocks.py:571-590: x + drop_path(attn(ln1(x))), out + drop_path(mlp(ln2(out)))): */
/* the sum autograd would form with a kernel of its own.  NULL: exactly vilco_layernorm_bwd.                               */
int vilco_layernorm_bwd_res(const float* dy, const float* x, const float* y, const float* gamma,
                            const float* mean, const float* rstd, const float* dres, float* dx, float* dgamma,
                            float* dbeta, int64_t rows, int32_t C, int32_t relu, void* workspace,
                            size_t workspace_bytes, void* stream);
/* Round 6: the same + partial maxima of |dx| (dx_amax_parts: device, >= 2048 floats; *n_parts = how many were written): dx is the
 * output gradient of the layer in front of the LayerNorm, whose mask / activation-backward kernel turns it into operand planes
 * (vilco_act_bwd_planes / _seq) from that bound. */
int vilco_layernorm_bwd_res_amax(const float* dy, const float* x, const float* y, const float* gamma,
                                 const float* mean, const float* rstd, const float* dres, float* dx, float* dgamma,
                                 float* dbeta, int64_t rows, int32_t C, int32_t relu, void* workspace,
                                 size_t workspace_bytes, float* dx_amax_parts, int32_t* n_parts, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Depthwise k=3 conv, stride 1|2, zero pad 1, no bias, output masked: MaskedMHCA's query/key/   */
/* value convs (blocks.py:312-334,363-368 via MaskedConv1D.forward :106-130).                    */
/* x[B][Tin][C], w[C][3], y[B][Tout][C], Tout = Tin/stride, valid(t') = stride*t' < in_len[b].   */
/* ------------------------------------------------------------------------------------------ */
int vilco_dwconv3_fwd(const float* x, const float* w, const int32_t* in_len, float* y, int32_t B,
                      int32_t Tin, int32_t C, int32_t stride, void* stream);
size_t vilco_dwconv3_bwd_workspace(int32_t B, int32_t Tin, int32_t C, int32_t stride);
int vilco_dwconv3_bwd(const float* dy, const float* x, const float* w, const int32_t* in_len,
                      float* dx, float* dw, int32_t B, int32_t Tin, int32_t C, int32_t stride,
                      void* workspace, size_t workspace_bytes, void* stream);

/* nn.MaxPool1d(3, 2, 1) skip path times the output mask (blocks.py:519-523,567). */
int vilco_maxpool3s2_fwd(const float* x, const int32_t* in_len, float* y, int32_t B, int32_t Tin,
                         int32_t C, void* stream);
int vilco_maxpool3s2_bwd(const float* dy, const float* x, const int32_t* in_len, float* dx,
                         int32_t B, int32_t Tin, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Row softmax over materialised attention scores S[B][H][Tq][Tk], in place.                     */
/* mode 0: key j masked (-inf) when j >= kv_len[b]   (blocks.py:258-260, 390-392)                */
/* mode 1: XLNet: score - 1e30 when (j >= kv_len[b] && j != i)  (modeling_xlnet_x.py:1184-1188,  */
/*         298-307); mode 2: no mask (ChannelAttention, blocks.py:432).                          */
/* ------------------------------------------------------------------------------------------ */
int vilco_softmax_fwd(float* s, const int32_t* kv_len, int32_t B, int32_t H, int32_t Tq,
                      int32_t Tk, int32_t mode, void* stream);
/* dS = P * (dP - sum_j dP*P), written over dp */
int vilco_softmax_bwd(float* dp, const float* p, int32_t B, int32_t H, int32_t Tq, int32_t Tk,
                      void* stream);
/* XLNet rel_shift_bnij fused with the add (modeling_xlnet_x.py:256-268,288,298):
 * s[b][h][i][j] += scale * bd[b][h][i][T - i + j],  bd is [B][H][T][2T]. */
int vilco_relshift_add(float* s, const float* bd, float scale, int32_t B, int32_t H, int32_t T,
                       void* stream);
/* adjoint: dbd[b][h][i][T-i+j] = scale * ds[b][h][i][j], all other entries 0 */
int vilco_relshift_bwd(const float* ds, float* dbd, float scale, int32_t B, int32_t H, int32_t T,
                       void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Fused (flash-style) masked multi-head attention, heads = channel slices of token-major q/k/v:  */
/* MaskedMHCA / MaskedMHA cores (blocks.py:383-400, 251-265) and, with `bias`, XLNet's             */
/* rel_attn_core (modeling_xlnet_x.py:270-320; bias = scale * rel_shift(bd)).  Scores stay on chip. */
/* q [B,Tq,H*hd], k/v [B,Tk,H*hd], bias [B,H,Tq,Tk] or null, lse [B,H,Tq] (saved for backward).     */
/* mask modes as vilco_softmax_fwd (+ 3: XLNet mask with the bias given as unshifted position scores [B,H,Tq,Tq+Tk]; */
/* + 4: sliding window |i - j| <= window below kv_len, Tq == Tk -- NLQ's LocalMaskedMHCA, NLQ/libs/modeling/blocks.py   */
/* :417-755; only the key tiles a query tile's windows reach are visited);                                           */
/* precision as vilco_gemm.  hd <= 160 (<= 64 in bf16 x3), hd % 4 == 0.  drop_p > 0: inverted dropout on the attention probabilities      */
/* (after the softmax, before P V) with a counter-based mask of its own (round 5; vilco_attn_dropout_mask writes it     */
/* out): one strong hash per row bh*Tq + i of stream drop_seed, a two-multiply finalizer per element (row key, j);      */
/* forward and backward must be given the same (drop_p, drop_seed).                                                    */
/* ------------------------------------------------------------------------------------------ */
int vilco_attn_supported(int32_t hd);
/* Output amax partials.  The hd = 64 / fp16 x2 / prefix-mask / no-bias / no-dropout kernels can leave max|x| partials of
 * their outputs (one float per workgroup) for the operand pack of the next product (vilco_pack_item.amax), which then
 * skips its amax launch.  Returns how many floats the o / dq buffer (key_side = 0, T = Tq) or the dk / dv buffers
 * (key_side = 1, T = Tk) must hold, or 0 when this configuration does not emit them (pass null then).  has_bias = 2
 * asks for the dS (dbias) partials of XLNet's relative attention (mask mode 3, Tq = Tk = T, key_side 0: dbias_amax). */
int32_t vilco_attn_amax_parts(int32_t B, int32_t H, int32_t T, int32_t hd, int32_t mode, int32_t precision,
                              int32_t has_bias, float drop_p, int32_t key_side);
/* Input amax partials (precision 3): max|x| partials of q / k / v / dout already on the device, e.g. left by the GEMM   */
/* that produced them (vilco_gemm_desc.amax_out).  A tensor with a NULL pointer or a count of 0 gets its own amax pass. */
typedef struct vilco_attn_amax_in {
  const float* q; int32_t nq;
  const float* k; int32_t nk;
  const float* v; int32_t nv;
  const float* dout; int32_t ndo;       /* backward only */
} vilco_attn_amax_in;
/* workspace = 16-bit operand planes (q, k natural; v transposed, or natural on the hd = 64 fast path), built inside
 * the call by the pack kernels; amax_in (may be null): see above; o_amax / d*_amax: see vilco_attn_amax_parts (null = not wanted) */
size_t vilco_attn_fwd_workspace(int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t hd, int32_t precision);
int vilco_attn_fwd(const float* q, const float* k, const float* v, const float* bias,
                   const int32_t* kv_len, float* o, float* lse, int32_t B, int32_t H, int32_t Tq,
                   int32_t Tk, int32_t hd, float scale, int32_t mode, int32_t window, int32_t precision, float drop_p,
                   uint32_t drop_seed, const vilco_attn_amax_in* amax_in, float* o_amax, void* workspace,
                   size_t workspace_bytes, void* stream);
/* the same, and o also written as the fp16 x2 operand planes (vilco_pack's layout for [B * Tq][H * hd], precision 3,      */
/* vilco_pack_bytes long, 256-byte aligned) of the output projection -- the hd = 64 kernels only (when                       */
/* vilco_attn_amax_parts(...) > 0, or XLNet's relative attention at hd = 64).  Scale from the bound max|o| <= max|v| / keep. */
int32_t vilco_attn_planes_supported(int32_t Tq, int32_t Tk, int32_t hd, int32_t mode, int32_t precision, int32_t has_bias,
                                    float drop_p);
int vilco_attn_fwd_planes(const float* q, const float* k, const float* v, const float* bias,
                   const int32_t* kv_len, float* o, float* lse, int32_t B, int32_t H, int32_t Tq,
                   int32_t Tk, int32_t hd, float scale, int32_t mode, int32_t window, int32_t precision, float drop_p,
                   uint32_t drop_seed, const vilco_attn_amax_in* amax_in, float* o_amax, void* workspace,
                   size_t workspace_bytes, void* o_planes, size_t o_planes_bytes, void* stream);
size_t vilco_attn_bwd_workspace(int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t hd, int32_t precision);
/* dq/dk/dv are overwritten; dbias (optional, [B,H,Tq,Tk]) receives dS.  Deterministic (no atomics). */
int vilco_attn_bwd(const float* q, const float* k, const float* v, const float* bias,
                   const int32_t* kv_len, const float* o, const float* lse, const float* dout,
                   float* dq, float* dk, float* dv, float* dbias, int32_t B, int32_t H, int32_t Tq,
                   int32_t Tk, int32_t hd, float scale, int32_t mode, int32_t window, int32_t precision, float drop_p,
                   uint32_t drop_seed, const vilco_attn_amax_in* amax_in, float* dq_amax, float* dk_amax, float* dv_amax,
                   float* dbias_amax, void* workspace, size_t workspace_bytes, void* stream);
/* Round 5, XLNet's relative attention (mask mode 3, hd = 64, precision 3, Tq = Tk): the backward writes dS -- the gradient  */
/* of the position scores, modeling_xlnet_x.py:256-288 -- directly as the fp16 x2 operand planes of the UNSHIFTED            */
/* [Tq][Tq + Tk] view (the layout vilco_pack_many gives an item with relshift = 1, nbatch = B*H), ready for the two          */
/* band-limited products d(qr) = d(bd) kr and d(kr) = d(bd)^T qr (vilco_gemm_desc.a_planes, band 2 / 3), instead of fp32     */
/* dS + a pack pass (0.68 GB written, 0.68 GB read and 1.36 GB written again at config P).  `ds_planes`: at least            */
/* vilco_attn_dsplanes_bytes(B, H, Tq) bytes, 256-byte aligned, whose out-of-band columns (p < Tq - i, p >= Tq + Tk - i of   */
/* row i) and padding are ZERO: the kernel writes the band only, so a buffer zeroed once can be reused call after call.      */
/* `dbias` and `dbias_amax` must be NULL with it and Tk a multiple of 64.  ds_planes NULL: exactly vilco_attn_bwd.          */
/* XLNet's position scores bd[b][h][i][p] = qr[b][i][h] . kr[(b)][p][h] for the band p in [T - i, 2T - i) of the unshifted    */
/* [T][2T] matrix -- the part rel_shift_bnij keeps (modeling_xlnet_x.py:204-214, 256-288); the rest of bd is left unwritten.   */
/* qr [B][T][H*hd], kr [2T][H*hd] (per_clip 0) or [B][2T][H*hd] (per_clip 1: the reference drops out the expanded position     */
/* embedding per batch element), bd [B][H][T][2T] fp32, read in place by vilco_attn_fwd / _bwd (mask mode 3).  hd = 64,       */
/* precision 3 only.  Replaces the band-1 vilco_gemm with K = 64 (write-bound at 1.9 TB/s; this kernel: the attention kernels'  */
/* first product with the qr fragments resident).                                                                             */
size_t vilco_xl_scores_workspace(int32_t B, int32_t H, int32_t T, int32_t per_clip);
int vilco_xl_scores(const float* qr, const float* kr, float* bd, int32_t B, int32_t H, int32_t T, int32_t hd,
                    int32_t per_clip, int32_t precision, void* workspace, size_t workspace_bytes, void* stream);
size_t vilco_attn_dsplanes_bytes(int32_t B, int32_t H, int32_t T);
int vilco_attn_bwd_dsplanes(const float* q, const float* k, const float* v, const float* bias,
                   const int32_t* kv_len, const float* o, const float* lse, const float* dout,
                   float* dq, float* dk, float* dv, float* dbias, int32_t B, int32_t H, int32_t Tq,
                   int32_t Tk, int32_t hd, float scale, int32_t mode, int32_t window, int32_t precision, float drop_p,
                   uint32_t drop_seed, const vilco_attn_amax_in* amax_in, float* dq_amax, float* dk_amax, float* dv_amax,
                   float* dbias_amax, void* workspace, size_t workspace_bytes, void* ds_planes, size_t ds_planes_bytes,
                   void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Candidate decode of PtTransformer.inference_single_video (MQ meta_archs.py:1594-1692, NLQ meta_archs.py:1253-1338)   */
/* for ONE clip, all L pyramid levels in one call: per level sigmoid(logit) > pre_nms_thresh on the valid positions,  */
/* the pre_nms_topk highest of those (exact; ties at the cut admitted in index order), segment = (t - off_l * stride,   */
/* t + off_r * stride), kept when longer than duration_thresh.  logits [R][C], offsets [R][2] (the regression head's    */
/* relu(Scale_l(x))), points [R][4] = (t, lo, hi, stride) share one row layout; level l owns rows level_row0[l] ..      */
/* level_row0[l] + level_len[l] - 1 (level_len = valid length of the clip at that level).  Outputs: the candidates of    */
/* level 0, then level 1, ... (inside a level in index order -- every consumer orders by score itself), out_total[0] =   */
/* how many; capacity L * topk rows.  L <= 64.                                                                           */
/* ------------------------------------------------------------------------------------------ */
size_t vilco_decode_workspace(int32_t L, int32_t topk);
int vilco_decode(const float* logits, const float* offsets, const float* points, const int32_t* level_row0,
                 const int32_t* level_len, int32_t C, int32_t L, int32_t topk, float pre_nms_thresh,
                 float duration_thresh, float* out_segs, float* out_scores, int64_t* out_labels,
                 int32_t* out_total, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Elementwise / reduction glue of TransformerBlock.forward (blocks.py:561-593).                 */
/* ------------------------------------------------------------------------------------------ */
/* out = a * (mask_a ? m : 1) + colscale[c] * rowscale[b] * bval ; colscale/rowscale/len optional */
int vilco_scale_add_fwd(float* out, const float* a, const float* bval, const float* colscale,
                        const float* rowscale, const int32_t* len, int32_t mask_a, int32_t B,
                        int32_t T, int32_t C, void* stream);
size_t vilco_colsum_workspace(int64_t rows, int32_t C);
/* da = dout*(mask_a?m:1) ; db = dout*colscale*rowscale ; dcolscale = sum_rows dout*bval*rowscale.
 * da / db / dcolscale may be null to skip. */
int vilco_scale_add_bwd(const float* dout, const float* bval, const float* colscale,
                        const float* rowscale, const int32_t* len, int32_t mask_a, float* da,
                        float* db, float* dcolscale, int32_t B, int32_t T, int32_t C,
                        void* workspace, size_t workspace_bytes, void* stream);
/* the same + partial maxima of |db| (db_amax_parts: device, >= 2048 floats; *n_parts = how many were written): db is the upstream
 * gradient of the residual branch's last layer, whose activation-backward kernel turns it into operand planes
 * (vilco_act_bwd_planes) */
int vilco_scale_add_bwd_amax(const float* dout, const float* bval, const float* colscale, const float* rowscale,
                             const int32_t* len, int32_t mask_a, float* da, float* db, float* dcolscale, int32_t B,
                             int32_t T, int32_t C, void* workspace, size_t workspace_bytes, float* db_amax_parts,
                             int32_t* n_parts, void* stream);
/* Inverted dropout y = keep ? x/(1-p) : 0 with a counter-based mask (element i of the stream `seed` at `offset + i`):   */
/* the backward pass is the same call on dy.  x = NULL writes the mask factors (0 or 1/(1-p)) -- what the parity tests  */
/* hand to the oracle.  nn.Dropout in modeling_xlnet_x.py:308,327,486,488,1201,1228,1280 and blocks.py:226,268,349.      */
int vilco_dropout(const float* x, float* y, int64_t n, float p, uint32_t seed, uint64_t offset, void* stream);
/* The mask factors (0 or 1/(1-p)) vilco_attn_fwd / vilco_attn_bwd apply to the attention probabilities under (p, seed):  */
/* y[rows = B*H*Tq][cols = Tk].  For tests and for replaying a step's masks in the CPU oracle; the reference draws them    */
/* from torch's Philox stream (nn.Dropout on the probabilities, MQ/libs/modeling/modeling_xlnet_x.py:308).                 */
int vilco_attn_dropout_mask(float* y, int64_t rows, int32_t cols, float p, uint32_t seed, void* stream);
/* Step word of the counter-based masks.  Every kernel that draws a dropout mask (vilco_dropout, vilco_act_bwd, the fused  */
/* epilogue dropout of vilco_gemm, the attention-probability dropout of vilco_attn_*) uses seed + word * 0x9E3779B1, where  */
/* `word` is ONE 32-bit value in device memory (0 after load: the effective seed is then the argument).  A training step    */
/* captured as a hipGraph replays its kernel arguments; with vilco_seed_word_bump as the graph's first node every replay    */
/* draws fresh masks, forward and backward of one replay the same ones.  (The reference draws from torch's Philox stream,  */
/* MQ/libs/modeling/blocks.py:226,268; modeling_xlnet_x.py:308: a stateful generator has no place in a replayed graph.)    */
/* _set / _bump are stream-ordered launches; _get synchronises the device (tests, logging).                                */
int vilco_seed_word_set(uint32_t value, void* stream);
int vilco_seed_word_bump(void* stream);
int vilco_seed_word_get(uint32_t* out);
/* out = alpha*a + beta*b (b may be null) */
int vilco_axpby(float* out, const float* a, const float* b, float alpha, float beta, int64_t n,
                void* stream);
/* dz = dropmask(dy) * act'(aux) * rowmask ; aux = pre-activation (gelu) or output (relu); act NONE = mask only;
 * drop_p > 0: the dropout mask (p, seed) of vilco_gemm's fused epilogue dropout, element index r*C + c.
 * Optional dbias[C] = column sums of dz (needs workspace). */
int vilco_act_bwd(const float* dy, const float* aux, float* dz, float* dbias, int32_t act,
                  const int32_t* len, int32_t T, int64_t rows, int32_t C, float drop_p, uint32_t drop_seed,
                  void* workspace, size_t workspace_bytes, void* stream);
/* the same + partial maxima of |dz| (amax_parts: device, >= 2048 floats; *n_parts = how many were written) */
int vilco_act_bwd_amax(const float* dy, const float* aux, float* dz, float* dbias, int32_t act,
                       const int32_t* len, int32_t T, int64_t rows, int32_t C, float drop_p, uint32_t drop_seed,
                       void* workspace, size_t workspace_bytes, float* amax_parts, int32_t* n_parts, void* stream);
/* the same, and dz written by this kernel as the fp16 x2 operand planes of its consumers (`planes`: vilco_pack's layout for
 * [rows][C] at precision 3, vilco_pack_bytes() long, 256-byte aligned; C % 32 == 0) -- no vilco_pack of dz follows.  The
 * planes' scale comes from a BOUND instead of the exact maximum: max|dy| (dy_amax: n_dy_amax partial maxima of |dy| left by
 * the producer of dy, e.g. vilco_gemm_desc.amax_out) x 1 / (1 - drop_p) x max|act'|; it is the same power-of-two rule, at
 * most two binary orders below the exact-maximum scale.  dz may be NULL (planes only). */
int vilco_act_bwd_planes(const float* dy, const float* aux, float* dz, float* dbias, int32_t act,
                         const int32_t* len, int32_t T, int64_t rows, int32_t C, float drop_p, uint32_t drop_seed,
                         void* workspace, size_t workspace_bytes, float* amax_parts, int32_t* n_parts,
                         const float* dy_amax, int32_t n_dy_amax, void* planes, size_t planes_bytes,
                         const float* row_mask, void* stream);      /* row_mask: as vilco_gemm_desc.row_mask (rows floats) or NULL */
/* Round 6: the same with the planes in the k=3 convs' zero-padded per-sequence image when seq_len > 0 (vilco_pack_item.seq_len: row
 * (b, t) at b * (seq_len + 2) + 1 + t, the pad rows and the slack zeroed here; C % 8 == 0, rows % seq_len == 0) -- the output gradient
 * of a masked k=3 conv (MQ/libs/modeling/blocks.py:79-84: conv output * mask) goes from the mask multiply straight into the operand
 * image of its dX and weight-gradient products, no fp32 dz, no vilco_pack_many.  seq_len = 0: vilco_act_bwd_planes.
 * vilco_act_bwd_planes_bytes: size of `planes` for either layout. */
size_t vilco_act_bwd_planes_bytes(int64_t rows, int32_t C, int32_t seq_len);
int vilco_act_bwd_planes_seq(const float* dy, const float* aux, float* dz, float* dbias, int32_t act,
                             const int32_t* len, int32_t T, int64_t rows, int32_t C, float drop_p, uint32_t drop_seed,
                             void* workspace, size_t workspace_bytes, float* amax_parts, int32_t* n_parts,
                             const float* dy_amax, int32_t n_dy_amax, void* planes, size_t planes_bytes, int32_t seq_len,
                             const float* row_mask, void* stream);
/* out[c] = sum_r x[r][c] */
int vilco_colsum(const float* x, float* out, int64_t rows, int32_t C, void* workspace,
                 size_t workspace_bytes, void* stream);
/* x[b][t][:] *= (t < len[b]) ; optional add: out = x + pe[t][c] * m  (backbones.py:222-226) */
int vilco_mask_rows(float* x, const int32_t* len, int32_t B, int32_t T, int32_t C, void* stream);
int vilco_add_pe(float* out, const float* x, const float* pe, const int32_t* len, int32_t B,
                 int32_t T, int32_t C, void* stream);
/* batched 2-D transpose: in[z][R][S] -> out[z][S][R]  (channel-first <-> token-major boundary) */
int vilco_transpose2d(const float* in, float* out, int32_t batch, int32_t R, int32_t S,
                      void* stream);
/* out[i][j][k] (contiguous, dims d0,d1,d2) = in[off + i*s0 + j*s1 + k*s2]  (conv weight permutes) */
int vilco_permute3(const float* in, float* out, int32_t d0, int32_t d1, int32_t d2, int64_t off,
                   int64_t s0, int64_t s1, int64_t s2, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Step glue of train_one_epoch (MQ/libs/utils/train_utils.py:343-351): clip_grad_norm_ + optimizer.step()  */
/* as multi-tensor kernels.  ptrs = device int64 [4][n] (param, grad, state1, state2 pointers of the n      */
/* gradient-bearing fp32 tensors), numel [n]; the work is cut into nchunks chunks of `chunk` elements        */
/* (chunk_tensor / chunk_off).  norm_coef (device float[2]) = {total L2 norm, min(1, max_norm/(norm+1e-6))}; */
/* pass it to vilco_optim_step to scale gradients without a host sync, or null.                              */
/* kind 0 = torch.optim.AdamW (decoupled decay; tensor_step = device float[n], each tensor's step count      */
/* after this update, for the bias corrections), 1 = SGD with momentum.                                      */
/* lr / wd are HOST arrays indexed by parameter group (group[n] on the device).                              */
/* ------------------------------------------------------------------------------------------ */
int vilco_grad_norm(const int64_t* ptrs, const int64_t* numel, const int32_t* chunk_tensor,
                    const int64_t* chunk_off, int32_t n, int32_t nchunks, int32_t chunk, float max_norm,
                    float* partial, float* norm_coef, void* stream);
int vilco_optim_step(int32_t kind, const int64_t* ptrs, const int64_t* numel, const int32_t* chunk_tensor,
                     const int64_t* chunk_off, const int32_t* group, int32_t n, int32_t nchunks, int32_t chunk,
                     const float* lr, const float* wd, int32_t ngroups, float beta1, float beta2, float eps,
                     float momentum, const float* tensor_step, const float* norm_coef, void* stream);
/* the same, and chunk_amax[c] (device float[nchunks], or null) = max|p| of the UPDATED parameter over chunk c: the   */
/* chunks of one tensor are consecutive, so chunk_amax + first_chunk(t) with count chunks(t) is a vilco_pack_item.amax */
/* for next step's pack of weight t -- the optimizer produces the scale of the fp16 x2 weight planes, the pack skips   */
/* its own amax pass over the weight.                                                                                  */
int vilco_optim_step_amax(int32_t kind, const int64_t* ptrs, const int64_t* numel, const int32_t* chunk_tensor,
                          const int64_t* chunk_off, const int32_t* group, int32_t n, int32_t nchunks, int32_t chunk,
                          const float* lr, const float* wd, int32_t ngroups, float beta1, float beta2, float eps,
                          float momentum, const float* tensor_step, const float* norm_coef, float* chunk_amax,
                          void* stream);
/* the same with the learning rates read from DEVICE memory when lr_dev != null (float[ngroups]; `lr` is then ignored):  */
/* a training iteration captured as a hipGraph replays its arguments, so the per-iteration schedule value               */
/* (MQ/libs/utils/lr_schedulers.py:71-104, stepped at train_utils.py:351) and the per-tensor step counts `tensor_step`   */
/* are memory the host (or a captured increment) rewrites between replays.                                              */
int vilco_optim_step_dev(int32_t kind, const int64_t* ptrs, const int64_t* numel, const int32_t* chunk_tensor,
                         const int64_t* chunk_off, const int32_t* group, int32_t n, int32_t nchunks, int32_t chunk,
                         const float* lr, const float* wd, int32_t ngroups, float beta1, float beta2, float eps,
                         float momentum, const float* tensor_step, const float* norm_coef, float* chunk_amax,
                         const float* lr_dev, void* stream);
/* dst[0..n) = vals[0..n): n <= 16 HOST floats carried as kernel arguments (stream-ordered, no staging buffer) -- how the */
/* host hands this iteration's learning rates to a captured optimizer step.                                              */
int vilco_store_f32(float* dst, const float* vals, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* q/k/v pre-projection of MaskedMHCA fused with the block's first LayerNorm (MQ/libs/modeling/blocks.py:561-563 */
/* `self.ln1(x)`, :363-369 query/key/value_conv (depthwise k=3, stride 1|2, masked) + query/key/value_norm):       */
/*   h = LN1(x);  y_j = LN_j(dwconv3(h; w_j) * mask),  j = q, k, v.   One read of x, three writes (+ h on request).  */
/* w / gam / bet / y / mean / rstd are HOST arrays of three device pointers (q, k, v); w_j is the [C][1][3] conv    */
/* weight, gam_j / bet_j the [C] LayerNorm affine (the reference's [1,C,1] tensors).  mean1 / rstd1 [B*T] and       */
/* mean_j / rstd_j [B*T/stride] are kept for backward; h may be NULL.  C must be a multiple of 256, at most 2304.   */
/* Backward recomputes the conv outputs from h: dc_j (scratch, [B][T/stride][C] each) receives the gradient wrt the  */
/* masked conv outputs, dh = dh_ext (may be NULL) + the conv-transpose of the three; dparams (15 C floats) = d gam_q, */
/* d bet_q, d gam_k, d bet_k, d gam_v, d bet_v as [6][C], then d w_q, d w_k, d w_v as [3][C][3] (each the weight's    */
/* own [C][1][3] layout).  LN1's own backward is vilco_layernorm_bwd on (dh, x, mean1, rstd1).                       */
/* ------------------------------------------------------------------------------------------ */
int vilco_qkv_pre_supported(int32_t C);
/* amax_parts (may be NULL): three device arrays of vilco_qkv_pre_amax_parts(B, T, stride) floats that receive the        */
/* partial maxima of |q|, |k|, |v| (0 = too many partials: not emitted).                                                 */
int vilco_qkv_pre_amax_parts(int32_t B, int32_t T, int32_t stride);
int vilco_qkv_pre_fwd(const float* x, const float* ln1_g, const float* ln1_b, const float* const* w,
                      const float* const* gam, const float* const* bet, const int32_t* len, float* h,
                      float* const* y, float* mean1, float* rstd1, float* const* mean, float* const* rstd,
                      float* const* amax_parts, int32_t B, int32_t T, int32_t C, int32_t stride, float eps1, float eps,
                      void* stream);
size_t vilco_qkv_pre_bwd_workspace(int32_t B, int32_t T, int32_t C, int32_t stride);
int vilco_qkv_pre_bwd(const float* h, const float* const* w, const float* const* gam, const float* const* dy,
                      const float* const* mean, const float* const* rstd, const int32_t* len,
                      const float* dh_ext, float* const* dc, float* dh, float* dparams, int32_t B, int32_t T,
                      int32_t C, int32_t stride, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Point labelling + losses of the heads in two launches each way: label_points_single_video                    */
/* (MQ/libs/modeling/meta_archs.py:1253-1344), losses (:1374-1447: focal on valid points x gaussian weights,      */
/* DIoU on positives, "al" loss, loss_normalizer EMA :1407-1410), sigmoid_focal_loss / ctr_diou_loss_1d           */
/* (losses.py:5-52, 109-168) and the regression head's last two layers relu(Scale_l(x)) (meta_archs.py:344-346).   */
/* Rows r of a clip are the pyramid points of all levels laid end to end, optionally with separator rows           */
/* (points[r].stride <= 0).  gt = float [B][3*Nmax + 1]: Nmax (start, end) pairs, Nmax class ids, the count.        */
/* gauss = [6][C]: mu, sigma, mu_reg_left, sigma_reg_left, mu_reg_right, sigma_reg_right.                          */
/* loss_norm (device float[1]) is read and updated in place.  almax_state: device uint64 [B*C], zero before the    */
/* first call and left zero by every call.  out = {cls_loss, reg_loss, al_loss, cls + loss_weight*reg +            */
/* al_weight*al}; saved[0] = the normaliser used.  center_radius <= 0: center_sample 'none'.                       */
/* bwd: g_cls / g_reg / g_al / g_final = device scalars, the upstream gradients of the four outputs (NULL = 0).  */
/* ------------------------------------------------------------------------------------------ */
typedef struct vilco_loss_desc {
  const float* logits;       /* [B][R][C] */
  const float* offsets;      /* [B][R][2] raw head output when level_scale != NULL, else already relu(scale * x) */
  const float* level_scale;  /* [L] or NULL */
  const float* points;       /* [R][4] = (t, reg_lo, reg_hi, stride) */
  const int32_t* row_level;  /* [R] */
  const int32_t* row_pos;    /* [R] position of the row inside its level */
  const int32_t* level_len;  /* [B][L] valid length of every level */
  const float* gt;           /* [B][3*Nmax + 1] */
  const float* gauss;        /* [6][C] */
  float* loss_norm;          /* [1] */
  int32_t B, R, C, L, Nmax;
  float center_radius, label_smoothing, momentum, loss_weight, al_weight;
  int32_t use_al;
} vilco_loss_desc;
size_t vilco_mq_loss_workspace(int32_t B, int32_t R, int32_t C);
int vilco_mq_loss_fwd(const vilco_loss_desc* d, float* out, float* saved, void* almax_state, void* workspace,
                      size_t workspace_bytes, void* stream);
/* backward: `workspace` is the forward's (its last region, B*R*8 floats, is scratch of THIS call: every row's share of the gradients of
 * the per-level regression scales and the gaussian-weight parameters, summed in a fixed order by a second launch -- no float atomics:
 * all gradients of a step are the same bits on every launch, round 6) */
int vilco_mq_loss_bwd(const vilco_loss_desc* d, const float* g_cls, const float* g_reg, const float* g_al,
                      const float* g_final, const float* saved, const void* workspace, float* d_logits,
                      float* d_offsets, float* d_level_scale, float* d_gauss, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Continual-learning regularisers of MQ/libs/cl_methods/EWC.py:6-22 (get_regularized_loss) and MAS.py:5-21   */
/* (get_mas_regularized_loss), called per iteration from train_utils.py:337-344, as ONE multi-tensor launch:    */
/*   out[0] = lambda * sum_t sum_{i < numel[t]} F_t[i] (opt_t[i] - p_t[i])^2,   grad_t[i] -= 2 lambda F (opt - p) */
/* ptrs = device int64 [4][n]: parameter, its gradient (accumulated into), importance F (Fisher / |grad|),      */
/* consolidated value opt; numel[t] = elements of opt_t (a PREFIX of the parameter when the class head has grown */
/* since the task, EWC.py:19-21).  Same chunk table as the optimizer.  shared_params != 0: some parameter occurs */
/* in more than one entry (several consolidated tasks) -> gradient accumulated with atomics.                    */
/* ------------------------------------------------------------------------------------------ */
int vilco_cl_penalty(const int64_t* ptrs, const int64_t* numel, const int32_t* chunk_tensor,
                     const int64_t* chunk_off, int32_t n, int32_t nchunks, int32_t chunk, float lambda,
                     int32_t shared_params, float* partial, float* out, void* stream);

/* Deferred finishing (csrc/defer.hip).  The second stage of every two-stage column reduction (LayerNorm d-gamma / d-beta,  */
/* bias and scale gradients, depthwise-tap gradients: reference autograd sums under blocks.py:152-166, 106-130, 628-641)  */
/* and the slab sum of a split-K product with a plain epilogue are small dependent launches whose results -- parameter     */
/* gradients -- nothing reads before backward ends.  While vilco_defer_set(1) is in force (process-wide) they are            */
/* recorded instead of launched; vilco_defer_flush(stream) issues all recorded items as a few batched launches (same       */
/* arithmetic, same order: bitwise the results of the individual launches) and switches recording off.  The partial buffers */
/* (workspaces) of recorded calls must stay alive until the flush.                                                        */
int vilco_defer_set(int32_t on);
int64_t vilco_defer_pending(void);
int vilco_defer_flush(void* stream);

/* ------------------------------------------------------------------------------------------ */
/* 1-D NMS on the device, replacing nms_1d_cpu (MQ/libs/utils/csrc/nms_cpu.cpp).                 */
/* Segments of all classes are passed concatenated; seg_off[nseg+1] gives each class's range      */
/* (batched_nms's per-class loop, nms.py:124-152, becomes one launch: one workgroup per class).   */
/* Outputs are per class at the same offsets: out_idx (indices LOCAL to the class, int64, in the  */
/* reference's return order) and out_cnt[nseg].  Index outputs are bit-exact vs the reference.    */
/* ------------------------------------------------------------------------------------------ */
size_t vilco_nms_workspace(int64_t n_total, int32_t nseg);
/* nms_cpu.cpp:19-58.  segs[n][2], scores[n]. */
int vilco_nms_1d(const float* segs, const float* scores, const int64_t* seg_off, int32_t nseg,
                 int64_t n_total, float iou_threshold, int64_t* out_idx, int64_t* out_cnt,
                 void* workspace, size_t workspace_bytes, void* stream);
/* nms_cpu.cpp:67-160.  dets[n][3] rows 0..cnt-1 of each class = (x1, x2, decayed score).
 * max_num > 0 stops each class after max_num picks (exact for the caller, nms.py:56-63). */
int vilco_softnms_1d(const float* segs, const float* scores, const int64_t* seg_off, int32_t nseg,
                     int64_t n_total, float iou_threshold, float sigma, float min_score,
                     int32_t method, int64_t max_num, float* dets, int64_t* out_idx,
                     int64_t* out_cnt, void* workspace, size_t workspace_bytes, void* stream);
/* Which device kernel a class of a vilco_softnms_1d call runs on is chosen per class from its size:
 *   kind 0  register-resident kernel (method 2 = Gaussian, nms_cpu.cpp:131-136; n <= 30 720)
 *   kind 1  row-strided kernel       (methods 0 / 1 / 2, nms_cpu.cpp:122-137;   n <= 65 536)
 *   kind 2  one-pass-per-pick kernel (any method, any n)
 * vilco_nms_set_kernel(-1) (the default) = the lowest kind that can take the class; kind k >= 0 = no kernel below kind k
 * (tests run every kernel on the same inputs this way).  Returns the previous setting.  All kernels produce the
 * reference's indices and scores bit for bit.  vilco_nms_last_kernels(): bit k set = the last vilco_softnms_1d call
 * launched kind k (host-side bookkeeping of the calling thread's last call; not synchronised). */
int vilco_nms_set_kernel(int32_t kind);
int vilco_nms_last_kernels(void);

#ifdef __cplusplus
}
#endif
#endif /* VILCO_HIP_H */
