/*
 * druggen_hip.h -- C ABI of libdruggen_hip.so (gfx950 / MI355X).
 *
 * The reference (HUBioDataLab/DrugGEN) has no FFI, plugin table or native code:
 * its hot path is PyTorch eager ops inside src/model/{layers,models,loss}.py.
 * The entry points below are therefore cut at the op groups that path launches
 * (SURVEY.md section 2.1 / 8b); each cites the reference lines it replaces.  A
 * maintainer binds them with ctypes; druggen_amd/_lib.py
 * is that binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to a contiguous, channel-last tensor; the
 *    caller (PyTorch) owns all memory, kernels never allocate or retain;
 *  - `dtype` (DG_DTYPE_F32 / DG_DTYPE_BF16) is the storage type of the ACTIVATION
 *    tensors of a call (the `void*` arguments): BASELINE configs[1] runs fp32,
 *    configs[2] keeps every [B,N,N,C] / [B,N,C] activation and activation gradient
 *    in bf16 in HBM.  Parameters, biases, LayerNorm statistics, weight gradients
 *    and all in-register arithmetic / MFMA accumulation are float32 in both modes;
 *  - `stream` is the caller's hipStream_t (torch.cuda.current_stream().cuda_stream);
 *    calls only enqueue work and return, there is no implicit synchronisation;
 *  - return value 0 = success, negative = library error (DG_E_*), positive =
 *    hipError_t; dg_last_error_string() gives a thread-local description;
 *  - no environment variable is read and there is no global mutable state except
 *    the opt-in profiler (dg_prof_*) and the edge-row threshold (dg_set_edge_rows);
 *    reduce batches, riding launches and the traversal direction are per host
 *    thread.  The library is re-entrant for nn.DataParallel's one-thread-per-GPU
 *    replicas (reference train.py:220-223).
 *
 * Shapes: B molecules, N = vertexes, C = dim (C % 4 == 0, C >= 8, N <= 96 for
 * the attention kernels), R = number of rows of a [R, C] row matrix.
 */
#ifndef DRUGGEN_HIP_H
#define DRUGGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DG_VERSION 232            /* 0.3.2: 0.3.1 + dg_ffn_ln_fwd_f32 (fused float32 feed-forward forward) */
#define DG_DTYPE_F32  0
#define DG_DTYPE_BF16 1
/* float32 activations whose 384-wide feed-forward HIDDEN tensors (h = relu(fc1 x), dh, and their second-order twins:
 * every [R,384] operand of dg_row_gemm / dg_linear_wgrad / dg_edge_ffn_ln_*) are stored as ONE fp16 plane: row r is
 * multiplied by the power of two that puts its largest magnitude into [2^14, 2^15), rounded to nearest-even, and the
 * inverse scale is kept per row.  Layout of such a buffer (dg_hidden_bytes(R, 384, dtype) bytes): [R][384] fp16, then --
 * at byte offset dg_hidden_scale_offset(R, 384) -- [R] float32 inverse scales.  Element error <= 2^-11 relative (rms
 * ~2e-4): inside the 1e-3 parity bar of the float32 configuration, and the two dominant kernels move 25-37 % fewer bytes.
 * Every other activation ([R,128] rows, statistics) stays float32; accepted wherever a call has a 384-wide operand, treated
 * as DG_DTYPE_F32 elsewhere.                                                                                          */
#define DG_DTYPE_F32_H16 2
/* float32 activations whose [R,384] hidden tensors keep the TOP 24 BITS of every float32 (sign, exponent, 15 mantissa bits:
 * 16 significant bits, round half up), three bytes per element, no scales: dg_hidden_bytes(R, 384, dtype) = 1152 R bytes, rows
 * contiguous, four elements = 12 bytes little endian.  Element error <= 2^-17 relative -- two orders inside the 1e-3 parity
 * bar -- for 25 % fewer bytes per hidden tensor.  Same calls and rules as DG_DTYPE_F32_H16.                              */
#define DG_DTYPE_F32_H24 3
/* dg_edge_ffn_ln_bwd / _bwd_pair only: the forward's hidden tensor `h` is plain float32, the BACKWARD's hidden tensor `dh` is
 * stored as DG_DTYPE_F32_H16 (4) / _H24 (5).  Rounding h perturbs the forward pass, and a perturbed forward flips ReLU masks in
 * the layers behind it -- gradient errors of the order of the SQUARE ROOT of the perturbation on small batches (measured: 1.4e-3
 * to 4e-3 per tensor on the two-molecule goldens even with the 2^-17 of _H24).  dh only travels through linear maps (dx = dz +
 * dh W1, dW1 = dh^T x): its rounding stays a rounding.                                                                   */
#define DG_DTYPE_F32_DH16 4
#define DG_DTYPE_F32_DH24 5
/* The float32-class storage of the FORWARD's hidden tensor: the hi / lo fp16 split of csrc/row_gemm.hip under one power-of-two
 * scale per row, done ONCE by the launch that writes h instead of by every launch that reads it.  Layout
 * (dg_hidden_bytes(R, 384, dtype)): [R][384] fp16 hi plane, then at byte offset dg_hidden_scale_offset(R, 384) the [R][384] fp16
 * lo plane, then at twice that offset the [R] float32 inverse scales -- 4 bytes per element like float32, hi + lo carries 22
 * significand bits of every element within 2^-18 of its row maximum.  dg_row_gemm reads both planes (three products, float32
 * class: nothing of the forward pass changes class); dg_linear_wgrad reads the hi plane only (the weight gradient is a backward
 * tensor: dW2 = dz^T h_hi, half the bytes, two products).  DG_DTYPE_F32_H32_DH16 (7): dg_edge_ffn_ln_bwd(_pair) with such an h
 * and dh as DG_DTYPE_F32_H16.                                                                                             */
#define DG_DTYPE_F32_H32 6
#define DG_DTYPE_F32_H32_DH16 7
#define DG_E_SHAPE   (-1)         /* unsupported shape (wrapper must not continue) */
#define DG_E_ARG     (-2)         /* null pointer / bad argument */
#define DG_E_WORKSPACE (-3)       /* workspace too small */
#define DG_EDGE_ROWS 65536        /* default of dg_set_edge_rows(): launches over at least this many rows count as edge-level */

typedef void* dg_stream_t;        /* hipStream_t */

int         dg_version(void);
const char* dg_last_error_string(void);
/* bytes of a [R,H] hidden tensor of the feed-forward for activations of `dtype` (H = 384), and the byte offset of the
 * inverse row scales inside a DG_DTYPE_F32_H16 buffer (0 for the other dtypes)                                         */
size_t dg_hidden_bytes(int64_t R, int H, int dtype);
size_t dg_hidden_scale_offset(int64_t R, int H);

/* ---- graph attention core: src/model/layers.py:119-134 (MHA.forward) --------
 *   s[b,i,j,c] = alpha * q[b,i,c] * k[b,j,c] * (e^2 + e)[b,i,j,c]      (lines 119-125)
 *   p = softmax_j(s);  o[b,i,c] = sum_j p[b,i,j,c] * v[b,j,c]          (lines 130-134)
 * q,k,v,o: [B,N,C]   e,s: [B,N,N,C]   alpha = 1/sqrt(C/heads).
 * `s` is the tensor the reference feeds to out_e (line 127); pass NULL to skip
 * writing it (Discriminator's last block never reads it, models.py:202-207).      */
int dg_attn_core_fwd(const void* q, const void* k, const void* v, const void* e,
                     void* s, void* o, int B, int N, int C, float alpha, int dtype, dg_stream_t stream);

/* First-order backward of the core (what autograd runs for lines 119-134).
 * ws = dL/ds [B,N,N,C], wo = dL/do [B,N,C]  ->  dq,dk,dv [B,N,C], de [B,N,N,C].
 * Recomputes s and p from (q,k,e); nothing but the inputs is saved.
 * ws may be NULL (treated as zeros).                                              */
int dg_attn_core_bwd(const void* q, const void* k, const void* v, const void* e,
                     const void* ws, const void* wo,
                     void* dq, void* dk, void* dv, void* de,
                     int B, int N, int C, float alpha, int dtype, dg_stream_t stream);
/* The same with `add_e` (NULL or [B,N,N,C]) added to de on its way out: the double backward of the gradient penalty
 * (loss.py:32-47) hands the second-order adjoint of e to the first-order pass this way.                          */
int dg_attn_core_bwd_add(const void* q, const void* k, const void* v, const void* e, const void* ws, const void* wo,
                         const void* add_e, void* dq, void* dk, void* dv, void* de, int B, int N, int C, float alpha,
                         int dtype, dg_stream_t stream);

/* Second-order: backward of dg_attn_core_bwd, needed by the WGAN-GP gradient
 * penalty (src/model/loss.py:32-39 create_graph=True, train.py:367).
 * (tq,tk,tv,te) are the adjoints of (dq,dk,dv,de).  Outputs are the adjoints of
 * the six inputs of dg_attn_core_bwd: gq,gk,gv [B,N,C], ge [B,N,N,C],
 * gws [B,N,N,C], gwo [B,N,C].  ws and gws may be NULL.                           */
int dg_attn_core_bwd2(const void* q, const void* k, const void* v, const void* e,
                      const void* ws, const void* wo,
                      const void* tq, const void* tk, const void* tv, const void* te,
                      void* gq, void* gk, void* gv, void* ge, void* gws, void* gwo,
                      int B, int N, int C, float alpha, int dtype, dg_stream_t stream);

/* ---- fused attention half of the edge branch: src/model/layers.py:116-135 + 186-190 ----
 *   e  = y We^T + be                       (layers.py:116; never written to HBM)
 *   s  = alpha q_i k_j (e^2 + e), p = softmax_j s, o_i = sum_j p v_j        (:119-134)
 *   y2 = LayerNorm(y + s Woe^T + boe) * gamma4 + beta4                       (:127, :188-190)
 * y, y2, pre4: [B,N,N,C]; q, k, v, o: [B,N,C] (node-level projections :111-113 stay row GEMMs);
 * mean4, rstd4: [B N N].  One kernel: the tile of one (b, i) -- N rows -- is read once, `e` and `s`
 * stay on chip; HBM traffic = read y, write y2 + pre4 (3 edge passes instead of 8).
 * `packed` = dg_attn_half_pack(e.weight, out_e.weight) for the same dtype.  Pass y2 = NULL for the
 * Discriminator's last block (models.py:202-207: only o is needed; boe / gamma4 / beta4 / pre4 /
 * mean4 / rstd4 are then ignored).  C == 128, N <= 96; dtype DG_DTYPE_BF16 or DG_DTYPE_F32.   */
size_t dg_attn_half_packed_bytes(int dtype);
int dg_attn_half_pack(const float* we, const float* woe, void* packed, int dtype, dg_stream_t stream);
int dg_attn_half_fwd(const void* y, const void* q, const void* k, const void* v, const void* packed,
                     const float* be, const float* boe, const float* gamma4, const float* beta4,
                     void* o, void* y2, void* pre4, float* mean4, float* rstd4,
                     int B, int N, int C, float alpha, float eps, int dtype, dg_stream_t stream);

/* The same attention half, FORWARD, for float32 rows (csrc/attn_half_f32.hip): e projection, scores, softmax and node
 * output, out_e, residual, ln4 in ONE producer / consumer kernel per call,
 *   e = y We^T + be;  sc = alpha q_i k_j (e + 1) e;  o_i = sum_j softmax_j(sc) v_j;  y2 = LN(y + sc Woe^T + boe),
 * with everything the float32 backward reads written on the way: e, sc (`s`), the pre-LayerNorm sum (`pre_ln`), mean / rstd
 * -- e, s and pre_ln may be NULL when no backward will follow (their stores are then skipped).  we_packed / woe_packed:
 * dg_row_gemm_pack(e.weight / out_e.weight, 128, 128, mode 0) (fp16 hi + lo arithmetic of dg_row_gemm); C == 128, N <= 96
 * -- above 48 neighbours a row group takes two passes of the 48-row stage, an online softmax joins them --
 * (others DG_E_SHAPE: the caller takes dg_row_gemm + dg_attn_core_fwd + dg_row_gemm).  HBM traffic: read y once, write e,
 * s, y2, pre_ln -- 5 x 4 R C bytes against 8 for the three launches.  Results are those of the unfused launches up to the
 * rounding of a different accumulation order.                                                                            */
int dg_attn_half_f32_fwd(const float* y, const float* q, const float* k, const float* v, const void* we_packed,
                         const float* be, const void* woe_packed, const float* boe, const float* gamma4,
                         const float* beta4, float* e, float* s, float* o, float* y2, float* pre_ln, float* mean4,
                         float* rstd4, int B, int N, int C, float alpha, float eps, dg_stream_t stream);

/* First part of the float32 backward of that half (csrc/attn_half_f32_bwd.hip): ln4 backward, ds = dz4 Woe and the attention
 * core backward in ONE kernel -- given dy2 = d loss / d y2 and d_o = d loss / d o it writes dz4 = LayerNormBackward(dy2; pre4,
 * mean4, rstd4, gamma4), de (gradient of the e projection's output), dq, dk, dv and dgamma4 / dbeta4 (nullable; adjacent in
 * memory inside dg_linear_wgrad_batch_begin / _end their reduction joins the batch).  ds never reaches HBM; e is read once.
 * The second part is dy = de We + dz4: dg_row_gemm (residual epilogue) or dg_row_gemm_ln_bwd.  woe_dgrad_packed:
 * dg_row_gemm_pack(out_e.weight, 128, 128, mode 1).  C == 128, N <= 48; a workgroup walks whole molecules (dk / dv are summed
 * over i in registers, fixed order), so fewer than ~128 molecules leave compute units idle: small batches take
 * dg_row_gemm_ln_bwd_in + dg_attn_core_bwd.  workspace >= dg_attn_half_f32_bwd1_workspace_bytes(B).  ds: NULL, or [B,N,N,128]
 * to ALSO write ds (a pass that a second order differentiates -- the gradient penalty's first backward -- saves it).      */
size_t dg_attn_half_f32_bwd1_workspace_bytes(int B);
int dg_attn_half_f32_bwd1(const float* dy2, const float* pre4, const float* mean4, const float* rstd4, const float* gamma4,
                          const void* woe_dgrad_packed, const float* e, const float* q, const float* k, const float* v,
                          const float* d_o, float* dz4, float* ds, float* de, float* dq, float* dk, float* dv,
                          float* dgamma4, float* dbeta4, void* workspace, size_t workspace_bytes, int B, int N, int C,
                          float alpha, dg_stream_t stream);

/* Backward of dg_attn_half_fwd given dz4 = d loss / d (y + s Woe^T + boe) -- the ln4 backward (dg_ln_residual_bwd on
 * pre4, mean4, rstd4) runs first and also yields dgamma4 / dbeta4 -- and d_o = d loss / d o.  e, s, p are recomputed
 * from y;  dy = dz4 + de We;  dq, dk, dv [B,N,C];  dwe/dwoe [C,C], dbe/dboe [C] (float32) are accumulated inside the
 * kernel (pass dwe = NULL to skip all four: input-gradient-only passes, loss.py:32-39 / the D pass of the G step).
 * dz4 = NULL: Discriminator's last block (no out_e / ln4: dwoe, dboe untouched).  HBM traffic: read y, dz4, write dy.
 * workspace >= dg_attn_half_bwd_workspace_bytes(B, N).  Results are bit-reproducible (fixed-order partial sums).       */
size_t dg_attn_half_bwd_workspace_bytes(int B, int N);
int dg_attn_half_bwd(const void* y, const void* dz4, const void* q, const void* k, const void* v, const void* d_o,
                     const void* packed, const float* be, void* dy, void* dq, void* dk, void* dv,
                     float* dwe, float* dbe, float* dwoe, float* dboe, void* workspace, size_t workspace_bytes,
                     int B, int N, int C, float alpha, int dtype, dg_stream_t stream);

/* ---- residual + LayerNorm: src/model/layers.py:185-192 ----------------------
 *   y = LayerNorm(a + r) * gamma + beta, eps = 1e-5, over the last dim C.
 * r may be NULL (ln1, layers.py:185).  a, r, y: [R,C]; mean, rstd: [R] (saved
 * for the backward).                                                              */
int dg_ln_residual_fwd(const void* a, const void* r, const float* gamma, const float* beta,
                       void* y, float* mean, float* rstd, int64_t R, int C, float eps,
                       int dtype, dg_stream_t stream);

/* Workspace (bytes) for the column reductions of the two calls below. */
size_t dg_ln_workspace_bytes(int64_t R, int C);

/* dy [R,C] -> dz [R,C] (gradient of both a and r), dgamma [C], dbeta [C].       */
int dg_ln_residual_bwd(const void* a, const void* r, const float* gamma,
                       const float* mean, const float* rstd, const void* dy,
                       void* dz, float* dgamma, float* dbeta,
                       void* workspace, size_t workspace_bytes,
                       int64_t R, int C, int dtype, dg_stream_t stream);

/* Same, with dz += dz_add (nullable): a second gradient source of the pre-LayerNorm sum
 * (the second-order pass of the gradient penalty) joins inside the kernel.             */
int dg_ln_residual_bwd_add(const void* a, const void* r, const float* gamma,
                           const float* mean, const float* rstd, const void* dy, const void* dz_add,
                           void* dz, float* dgamma, float* dbeta,
                           void* workspace, size_t workspace_bytes,
                           int64_t R, int C, int dtype, dg_stream_t stream);

/* Backward of dg_ln_residual_bwd w.r.t. the adjoint tz of dz (the gradient
 * penalty differentiates the first backward w.r.t. inputs only):
 * -> gz [R,C] (adjoint of a and r), gdy [R,C] (adjoint of dy), ggamma [C].       */
int dg_ln_residual_bwd2(const void* a, const void* r, const float* gamma,
                        const float* mean, const float* rstd, const void* dy, const void* tz,
                        void* gz, void* gdy, float* ggamma,
                        void* workspace, size_t workspace_bytes,
                        int64_t R, int C, int dtype, dg_stream_t stream);

/* ---- Linear weight/bias gradient over the edge rows ---------------------------
 * What autograd runs as mm(dy.t(), x) and sum(dy, 0) for every nn.Linear of
 * src/model/layers.py (MHA q/k/v/e/out_e/out_n at :86-95, MLP fc1/fc2 at :36-38)
 * and again in the gradient-penalty double backward (src/model/loss.py:32-39):
 *   dw[n][k] = sum_r dy[r][n] x[r][k]   (dw: [N,K], nn.Linear layout)
 *   db[n]    = sum_r dy[r][n]           (db may be NULL)
 * dy: [R,N], x: [R,K] (dtype), dw/db float32.  fp32: operands split 3-way into bf16, six MFMA
 * cross products (fp32-class accuracy); bf16: one v_mfma_f32_32x32x16_bf16 per product.  Split
 * over the rows, fixed-order reduction (bit-reproducible).
 * dy_mask (nullable, [R,N]): use dy * (dy_mask > 0), i.e. the ReLU backward of
 * MLP.fc1 (layers.py:51) folded into the operand load.
 * Supported (N,K): multiples of 32 from the table in csrc/linear_wgrad.hip
 * (128x128, 384x128, 128x384, 128x64, ...), plus N <= 16 with K % 4 == 0 (readout
 * layers, models.py:67-68: a streaming VALU kernel); others return DG_E_SHAPE.   */
size_t dg_linear_wgrad_workspace_bytes(int64_t R, int N, int K);
int dg_linear_wgrad(const void* dy, const void* dy_mask, const void* x, float* dw, float* db,
                    void* workspace, size_t workspace_bytes,
                    int64_t R, int N, int K, int dtype, dg_stream_t stream);

/* dW [384,128] = [dy0 | dy1 | dy2]^T x and db [384] (nullable): the weight gradients of three Linear(128,128) that share
 * their input x [R,128] -- q / k / v of an attention block (layers.py:111-113 backward) -- in ONE launch; rows 0..127 /
 * 128..255 / 256..383 of dW (and of db) are the three gradients.  float32 only (others DG_E_SHAPE); workspace as for
 * dg_linear_wgrad(R, 384, 128).                                                                                     */
int dg_linear_wgrad3(const void* dy0, const void* dy1, const void* dy2, const void* x, float* dw, float* db,
                     void* workspace, size_t workspace_bytes, int64_t R, int dtype, dg_stream_t stream);

/* Deferred reduces: between _batch_begin() and _batch_end() every dg_linear_wgrad call with N > 16 only runs its
 * split-K kernel and records its fixed-order reduce; _batch_end() runs up to 8 recorded reduces in ONE launch (a 9th
 * call reduces at once).  Each call of a batch needs its OWN workspace; dw / db are complete after _batch_end().  The
 * state is per host thread.  (The six projections of an attention block, fc1 + fc2 of a feed-forward block.)
 * dg_ln_residual_bwd(_add) calls inside a batch whose dgamma and dbeta are adjacent (dbeta == dgamma + C) join it too:
 * their partial sums (the workspace) must then stay untouched until _batch_end().                                  */
int dg_linear_wgrad_batch_begin(void);
int dg_linear_wgrad_batch_end(dg_stream_t stream);

/* Riding launches: the node branch of an Encoder_Block (mlp / ln5, reference layers.py:191) runs the same kernels as its edge
 * branch (mlp2 / ln6, :192) over N times fewer rows -- launches of 10-25 us that are all prologue and tail.  Between
 * _pair_begin() and _pair_end() a float32 dg_row_gemm with (K,N) = (128,384) or (384,128), or a float32 dg_linear_wgrad
 * / dg_linear_wgrad3 on the producer / consumer kernel, over at most 65 536 rows is NOT launched when it is called: it
 * waits (one per kernel) and runs as a second problem inside the NEXT launch of the same kernel (same epilogue class for
 * the 384 -> 128 GEMM, same shape for the weight gradient), whose workgroups are split between the two in proportion to
 * their rows.  Results are those of separate launches (weight gradients: a different, still fixed, number of partial
 * sums).  The CALLER guarantees that no other launch reads a waiting problem's output before its carrier was called:
 * issue node, edge, node, edge ...  _pair_end() launches whatever still waits on `stream`.  Regions nest (the outermost
 * _pair_end() launches); per host thread.                                                                             */
int dg_launch_pair_begin(void);
int dg_launch_pair_end(dg_stream_t stream);

/* ---- fp32-MFMA row GEMM with fused epilogues ------------------------------------
 * The dense layers applied to every edge / node row: MHA projections
 * (src/model/layers.py:111-116,127,135) and MLP.fc1/fc2 (:50-53), forward and
 * input-gradient, with the elementwise ops around them folded in:
 *   y[R,N] = epi( a[R,K] . B ),
 *   epi(v) = LN?( relu?(v + bias) * relu_mask? + residual? )
 * `packed` is the weight in MFMA fragment order made by dg_row_gemm_pack from the
 * nn.Linear weight w[rows,cols]: mode 0 -> B[k][n] = w[n][k] (forward, N=rows,
 * K=cols); mode 1 -> B[k][n] = w[k][n] (input gradient dx = dy.w, N=cols, K=rows).
 * (K,N) in {(128,128), (128,384), (384,128)}; others DG_E_SHAPE.
 * fp32 arithmetic: rows of a (per 128-wide chunk) and columns of B are scaled by an exact power of two
 * and split hi + lo into fp16; three MFMA products with fp32 accumulation, inverse scales in the
 * epilogue -- element-wise error at the level of an fp32 GEMM (tests/test_hip_kernels.py).
 * ReLU backward without a pass over the activations: a forward launch with
 * relu != 0 can write one bit per output element into relu_bits_out
 * (dg_row_gemm_mask_words(R,K,N) uint32 words, laid out per tile/wave/lane); the
 * input-gradient launch of the NEXT layer (same R, K, N geometry) takes it as
 * mask_bits and zeroes the masked outputs (reference: threshold_backward of
 * layers.py:51).  LayerNorm epilogue (gamma != NULL; layers.py:187-192) needs
 * N == 128, writes mean/rstd [R] and, if pre_ln != NULL, the pre-LayerNorm sum.   */
size_t dg_row_gemm_packed_bytes(int n_out, int k_contract, int dtype);
int dg_row_gemm_pack(const float* w, void* packed, int rows, int cols, int mode, int dtype, dg_stream_t stream);
/* Many packs in one launch (after an optimizer step every weight of the network is stale at once): `table` is a DEVICE
 * array of n entries { const float* w; void* packed; int64 rows; int64 cols; int64 mode; const float* w1; const float*
 * w2 } (7 x int64 each; w1 = w2 = NULL except for a dg_row_gemm_pack3 stack, where rows = 384), max_dim >= every rows /
 * cols; for DG_DTYPE_BF16 every rows / cols must be a multiple of 32 (and w1 / w2 are ignored).                      */
int dg_row_gemm_pack_batch(const void* table, int n, int max_dim, int dtype, dg_stream_t stream);
/* Three Linear(128,128) that share their input (q / k / v of an attention block, layers.py:111-113) as ONE operand:
 *   dg_row_gemm_pack3   the vertical stack [w0; w1; w2] ([384,128]) packed like dg_row_gemm_pack (mode 0: the 128 -> 384
 *                       forward operand; mode 1: the 384 -> 128 input-gradient operand); dg_row_gemm_packed_bytes(384, 128)
 *                       resp. (128, 384) bytes;
 *   dg_row_gemm_lin3    y_i [R,128] = a [R,128] . w_i^T + b_i (b_i nullable), i = 0..2, one launch (mode-0 pack);
 *   dg_row_gemm_sum3    y [R,128] = a0 . w0 + a1 . w1 + a2 . w2 (+ residual [R,128], nullable), one launch (mode-1 pack).
 * float32 activations only (DG_E_SHAPE otherwise: callers keep three dg_row_gemm launches for bfloat16).              */
int dg_row_gemm_pack3(const float* w0, const float* w1, const float* w2, void* packed, int cols, int mode, int dtype,
                      dg_stream_t stream);
int dg_row_gemm_lin3(const void* a, const void* packed, void* y0, void* y1, void* y2, int64_t R, const float* b0,
                     const float* b1, const float* b2, int dtype, dg_stream_t stream);
int dg_row_gemm_sum3(const void* a0, const void* a1, const void* a2, const void* packed, void* y, int64_t R,
                     const void* residual, int dtype, dg_stream_t stream);
size_t dg_row_gemm_mask_words(int64_t R, int K, int N, int dtype);
int dg_row_gemm(const void* a, const void* packed, void* y, int64_t R, int K, int N,
                const float* bias, int relu, unsigned* relu_bits_out, const unsigned* mask_bits,
                const void* residual,
                const float* gamma, const float* beta, float* mean, float* rstd, void* pre_ln,
                float eps, int dtype, dg_stream_t stream);
/* Input-gradient GEMM whose output is the gradient of a LayerNorm OUTPUT, with that LayerNorm's backward as the
 * epilogue (src/model/layers.py:187-192 backward: the dgrad of the next Linear feeds ln4 / ln6):
 *   v  = a[R,K] . B + residual                      (K = N = 128; residual nullable)
 *   dz = rstd (v gamma - mean(v gamma) - xhat mean(v gamma xhat)),  xhat = (ln_pre - ln_mean) ln_rstd
 *   dgamma = sum_r v xhat,  dbeta = sum_r v          (either may be NULL)
 * v itself is never written: one launch replaces the GEMM and the dg_ln_residual_bwd that would read it back
 * (2 of its 3 [R,128] passes).  float32 only.  workspace >= dg_row_gemm_ln_bwd_workspace_bytes(dtype).            */
size_t dg_row_gemm_ln_bwd_workspace_bytes(int dtype);
int dg_row_gemm_ln_bwd(const void* a, const void* packed, void* dz, int64_t R, int K, const void* residual,
                       const void* ln_pre, const float* ln_mean, const float* ln_rstd, const float* ln_gamma,
                       float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, int dtype,
                       dg_stream_t stream);
/* The mirror image: the GEMM whose A operand is the INPUT gradient of a LayerNorm (src/model/layers.py:187-190
 * backward: ln4 / ln3 feed the input-gradient GEMM of out_e / out_n), with that LayerNorm's backward on the way in:
 *   dz = rstd (dy gamma - mean(dy gamma) - xhat mean(dy gamma xhat)),  xhat = (ln_pre - ln_mean) ln_rstd
 *   y  = dz[R,K] . B                                   (K = N = 128, plain epilogue)
 *   dgamma = sum_r dy xhat,  dbeta = sum_r dy          (either may be NULL)
 * dz [R,128] is written too (the residual path and the weight gradient need it); one launch replaces
 * dg_ln_residual_bwd + dg_row_gemm and one of their five [R,128] passes.  float32 only.
 * workspace >= dg_row_gemm_ln_bwd_workspace_bytes(dtype); inside dg_linear_wgrad_batch_begin / _end the dgamma / dbeta
 * reduction joins the batch when dbeta == dgamma + 128 (the workspace must then stay untouched until _end).      */
int dg_row_gemm_ln_bwd_in(const void* dy, const void* ln_pre, const float* ln_mean, const float* ln_rstd,
                          const float* ln_gamma, const void* packed, void* dz, void* y, float* dgamma, float* dbeta,
                          void* workspace, size_t workspace_bytes, int64_t R, int K, int N, int dtype,
                          dg_stream_t stream);
/* dtype = DG_DTYPE_BF16 (csrc/gemm_bf16.hip): a, y, residual, pre_ln are bf16; the packed weight is the
 * bf16 fragment-order copy made by dg_row_gemm_pack(..., DG_DTYPE_BF16, ...); one MFMA per product,
 * fp32 accumulate and epilogue arithmetic.  Bit masks are available for every shape in this mode (the
 * layout is private to the (K, N, dtype) triple: only pass a mask to a launch of the same triple).   */

/* ---- feed-forward half of an Encoder_Block: src/model/layers.py:191-192 + MLP (:40-54) ----
 *   y = LayerNorm(x + fc2(relu(fc1(x)))) * gamma + beta      (dim C = 128, hidden H = 384)
 * One call per direction; internally the row-GEMM / LayerNorm / weight-gradient kernels above,
 * sequenced on `stream` (bias, ReLU, ReLU mask, residual and LayerNorm all live in GEMM
 * epilogues).  Weights are passed in fragment order: forward packs (dg_row_gemm_pack mode 0) for
 * _fwd, input-gradient packs (mode 1) for _bwd.  The forward saves h [R,H], the packed ReLU
 * bits (dg_row_gemm_mask_words(R,C,H) words), the pre-LayerNorm sum [R,C] and mean/rstd [R].
 * _bwd: dz_add (nullable, [R,C]) is added to the LayerNorm input-gradient (dy == NULL with dz_add == dz: dz already
 * holds that gradient -- made by dg_row_gemm_ln_bwd -- and dgamma / dbeta are left alone); outputs: dz [R,C] and
 * dh [R,H] (scratch the caller owns), dx (nullable), dgamma, dbeta,
 * dw1 [H,C], db1, dw2 [C,H], db2 (dw1/dw2 nullable = skip the weight gradients).             */
size_t dg_edge_ffn_ln_workspace_bytes(int64_t R, int C, int H);
int dg_edge_ffn_ln_fwd(const void* x, const void* w1_packed, const float* b1, const void* w2_packed,
                       const float* b2, const float* gamma, const float* beta,
                       void* y, void* h, unsigned* relu_bits, void* pre_ln, float* mean, float* rstd,
                       int64_t R, int C, int H, float eps, int dtype, dg_stream_t stream);
int dg_edge_ffn_ln_bwd(const void* x, const void* h, const unsigned* relu_bits, const void* pre_ln,
                       const float* mean, const float* rstd, const float* gamma,
                       const void* w1_dgrad_packed, const void* w2_dgrad_packed, const void* dy,
                       const void* dz_add,
                       void* dz, void* dh, void* dx, float* dgamma, float* dbeta,
                       float* dw1, float* db1, float* dw2, float* db2,
                       void* workspace, size_t workspace_bytes, int64_t R, int C, int H, int dtype,
                       dg_stream_t stream);

/* The two feed-forward halves of an Encoder_Block -- mlp / ln5 over the B N node rows, mlp2 / ln6 over the B N^2 edge rows
 * (layers.py:191-192) -- in ONE call per direction: exactly dg_edge_ffn_ln_fwd / _bwd for `node` and for `edge` (same
 * arguments, as structs; each with its own workspace of dg_edge_ffn_ln_workspace_bytes(R, C, H)), issued node, edge, node,
 * edge ... inside dg_launch_pair_begin / _end, so that every node-level launch rides in the edge-level launch of the same
 * kernel; the backward's reductions (LayerNorm dgamma / dbeta, four weight gradients) share one launch.  float32 rows on the
 * producer / consumer kernels ride; other dtypes / kernels simply run one after the other.                                  */
typedef struct dg_ffn_fwd_args {
    const void* x; const void* w1_packed; const float* b1; const void* w2_packed; const float* b2;
    const float* gamma; const float* beta;
    void* y; void* h; unsigned* relu_bits; void* pre_ln; float* mean; float* rstd;
    int64_t R; float eps;
} dg_ffn_fwd_args;
typedef struct dg_ffn_bwd_args {
    const void* x; const void* h; const unsigned* relu_bits; const void* pre_ln; const float* mean; const float* rstd;
    const float* gamma; const void* w1_dgrad_packed; const void* w2_dgrad_packed; const void* dy; const void* dz_add;
    void* dz; void* dh; void* dx; float* dgamma; float* dbeta; float* dw1; float* db1; float* dw2; float* db2;
    void* workspace; size_t workspace_bytes; int64_t R;
} dg_ffn_bwd_args;
int dg_edge_ffn_ln_fwd_pair(const dg_ffn_fwd_args* node, const dg_ffn_fwd_args* edge, int C, int H, int dtype,
                            dg_stream_t stream);
int dg_edge_ffn_ln_bwd_pair(const dg_ffn_bwd_args* node, const dg_ffn_bwd_args* edge, int C, int H, int dtype,
                            dg_stream_t stream);

/* ---- the same feed-forward half as FUSED kernels, bf16 configuration only (csrc/ffn_bf16.hip) ----
 *   y = LayerNorm(x + fc2(relu(fc1(x)))) * gamma + beta,  C = 128, H = 384, all activations bf16.
 * The [R,384] hidden tensor never reaches HBM: per 64-row tile it lives in LDS; the backward
 * recomputes it.  `packed` holds the four bf16 fragment-order copies of (W1 [384,128], W2 [128,384])
 * made by dg_ffn_bf16_pack (dg_ffn_bf16_packed_bytes() bytes).  The forward saves pre_ln [R,128] bf16,
 * mean / rstd [R] and one ReLU-mask bit per hidden element (dg_ffn_bf16_mask_words(R) uint32 words);
 * pre_ln and relu_bits may be NULL when no backward will follow.  The forward stores whole 64-row tiles: its four
 * outputs must be allocated with dg_ffn_bf16_padded_rows(R) rows (the first R are the result).
 * _bwd: dy [R,128] -> dz [R,128] (scratch the caller owns: LayerNorm input gradient, bf16), dx (nullable),
 * dgamma, dbeta, and -- when dw1 != NULL -- dw1 [384,128], db1, dw2 [128,384], db2 (float32);
 * bits_scratch: dg_ffn_bf16_mask_words(R) words of scratch, needed with the weight gradients.
 * First order only: the gradient penalty's twice-differentiated pass uses dg_edge_ffn_ln_fwd/_bwd.   */
int64_t dg_ffn_bf16_padded_rows(int64_t R);     /* rows y / pre_ln / mean / rstd of _fwd must hold (R rounded up to whole tiles) */
size_t dg_ffn_bf16_packed_bytes(void);
int dg_ffn_bf16_pack(const float* w1, const float* w2, void* packed, dg_stream_t stream);
size_t dg_ffn_bf16_mask_words(int64_t R);
size_t dg_ffn_bf16_workspace_bytes(int64_t R);
int dg_ffn_ln_fwd_bf16(const void* x, const void* packed, const float* b1, const float* b2, const float* gamma,
                       const float* beta, void* y, void* pre_ln, float* mean, float* rstd, unsigned* relu_bits,
                       int64_t R, float eps, dg_stream_t stream);
int dg_ffn_ln_bwd_bf16(const void* x, const void* pre_ln, const float* mean, const float* rstd,
                       const unsigned* relu_bits, const float* gamma, const void* packed, const float* b1,
                       const void* dy, void* dz, void* dx, float* dgamma, float* dbeta,
                       float* dw1, float* db1, float* dw2, float* db2, unsigned* bits_scratch,
                       void* workspace, size_t workspace_bytes, int64_t R, dg_stream_t stream);


/* ---- the same feed-forward half as ONE FUSED forward kernel, float32 activations (csrc/ffn_fused_f32.hip) ----
 *   y = LayerNorm(x + fc2(relu(fc1(x)))) * gamma + beta   (src/model/layers.py:50-53, 191-192),  C = 128, H = 384,
 * float32-class arithmetic (fp16 hi + lo planes, three MFMA products, fp32 accumulation: the arithmetic of dg_row_gemm).
 * The [R,384] hidden tensor never reaches HBM as float32: the rows stay in registers between fc1 and fc2, both weights
 * stream L2 -> LDS from the fragment-order copy made by dg_ffn_f32_pack (dg_ffn_f32_packed_bytes() bytes; re-pack after every
 * update of w1 [384,128] / w2 [128,384]).  Arguments: dg_ffn_fwd_args with `w1_packed` = that copy (`w2_packed` is ignored);
 * `h` = a DG_DTYPE_F32_H16 buffer (dg_hidden_bytes(R, 384, DG_DTYPE_F32_H16): the hi fp16 plane of h under one row scale -- all
 * the unfused backward reads of h, dW2 = dz^T h), `relu_bits` = dg_row_gemm_mask_words(R, 128, 384, DG_DTYPE_F32) words in the
 * layout dg_row_gemm writes (dg_edge_ffn_ln_bwd(_pair) with dtype DG_DTYPE_F32_H16 runs on the result unchanged), `pre_ln`
 * [R,128]; h, relu_bits and pre_ln are all NULL when no backward follows.  `node` may be NULL; otherwise its (few) rows ride
 * in the launch over `edge`'s rows (own weights, own outputs).                                                            */
size_t dg_ffn_f32_packed_bytes(void);
int dg_ffn_f32_pack(const float* w1, const float* w2, void* packed, dg_stream_t stream);
int dg_ffn_ln_fwd_f32(const dg_ffn_fwd_args* node, const dg_ffn_fwd_args* edge, dg_stream_t stream);

/* ---- edge embedding + symmetrisation: src/model/models.py:57-61,92-94 (Generator) and
 * :159-163,197-199 (Discriminator) ---------------------------------------------------
 *   f(z) = act(W2.act(W1.z + b1) + b2),  out[b,i,j,:] = (f(a[b,i,j,:]) + f(a[b,j,i,:])) / 2
 * a: [B,N,N,E] (E <= 16), w1: [64,E], w2: [128,64] given in the fragment order made by
 * dg_embed_sym_pack (forward) and dg_row_gemm_pack(w2,128,64,mode 1) (backward), out / g:
 * [B,N,N,128]; act: 0 relu, 1 leaky(0.01), 2 sigmoid, 3 tanh (models.py:39-46).
 * The backward recomputes both layers (nothing but the inputs is saved) and returns
 * da (may be NULL when the input needs no gradient), dw1, db1, dw2, db2.            */
size_t dg_embed_sym_packed_floats(void);
size_t dg_embed_sym_workspace_bytes(int B, int N);
int dg_embed_sym_pack(const float* w2, float* packed, dg_stream_t stream);
/* Layer-2 weight as the input-gradient operand (dh = dz2 . W2) in fp32 MFMA fragment order, for _bwd. */
size_t dg_embed_sym_dgrad_packed_floats(void);
int dg_embed_sym_pack_dgrad(const float* w2, float* packed, dg_stream_t stream);
int dg_embed_sym_fwd(const float* a, const float* w1, const float* b1, const float* w2_packed, const float* b2,
                     void* out, int B, int N, int E, int H, int C, int act, int dtype, dg_stream_t stream);
int dg_embed_sym_bwd(const float* a, const float* w1, const float* b1, const float* w2_packed,
                     const float* w2_dgrad_packed, const float* b2, const void* g,
                     float* da, float* dw1, float* db1, float* dw2, float* db2,
                     void* workspace, size_t workspace_bytes,
                     int B, int N, int E, int H, int C, int act, int dtype, dg_stream_t stream);
/* dtype: storage of `out` / `g` (the [B,N,N,128] edge tensor); the one-hot input `a`, its gradient and
 * the parameters stay float32.                                                                       */
/* The same backward for bf16 gradients `g` and the piecewise-linear activations (act 0 = relu, 1 = leaky), E <= 8: whole
 * row blocks g[b,i,:,:] and g[b,:,i,:] stream through LDS, one bf16 MFMA per product (the bf16 configuration's
 * arithmetic), fp32 accumulation -- ~5x faster than dg_embed_sym_bwd at configs[2].  w2 is the RAW [C,H] float32
 * parameter (fragments are built in the kernel); da may be NULL.                                                   */
size_t dg_embed_sym_bwd_bf16_workspace_bytes(int B, int N);
int dg_embed_sym_bwd_bf16(const float* a, const float* w1, const float* b1, const float* w2, const float* b2,
                          const void* g, float* da, float* dw1, float* db1, float* dw2, float* db2,
                          void* workspace, size_t workspace_bytes, int B, int N, int E, int H, int C, int act,
                          dg_stream_t stream);

/* Backward of dg_embed_sym_bwd for the gradient penalty (src/model/loss.py:32-39 differentiates d out / d a):
 * t [B,N,N,E] is the adjoint of da.  Outputs: gg [B,N,N,C] (dtype) = adjoint of g, gw1 [H,E], gw2 [C,H] (fp32).
 * Only the piecewise-linear activations (act = relu, leaky): act'' = 0, so nothing reaches a, b1 and b2 and the
 * adjoints of dw1/db1/dw2/db2 are not taken (they are not differentiated on this path); others: DG_E_ARG.   */
int dg_embed_sym_bwd2(const float* a, const float* w1, const float* b1, const float* w2_packed,
                      const float* w2_dgrad_packed, const float* b2, const void* g, const float* t,
                      void* gg, float* gw1, float* gw2, void* workspace, size_t workspace_bytes,
                      int B, int N, int E, int H, int C, int act, int dtype, dg_stream_t stream);

/* One-hot fast path of the same op (reference src/data/utils.py:15-23 makes the generator's input and the
 * discriminator's real batch one-hot): with labels l [B,N,N] (int32, 0 <= l < E) and the E x C table
 * T[c] = f(e_c) (computed and differentiated by the caller: E rows of an MLP),
 *   fwd: out[b,i,j,:] = (T[l_ij] + T[l_ji]) / 2          bwd: dT[c,:] = sum_rows g_ij ([l_ij = c] + [l_ji = c]) / 2
 * HBM-bound gather / segmented sum instead of R = B N^2 MLP evaluations.  C = 128, E <= 16.          */
size_t dg_onehot_embed_workspace_bytes(int E, int C);
int dg_onehot_embed_fwd(const int* labels, const float* table, void* out, int B, int N, int E, int C, int dtype,
                        dg_stream_t stream);
int dg_onehot_embed_bwd(const int* labels, const void* g, float* dtable, void* workspace, size_t workspace_bytes,
                        int B, int N, int E, int C, int dtype, dg_stream_t stream);

/* ---- the steps either side of the path -------------------
 * dg_densify: reference src/data/utils.py:128-137 -- PyG to_dense_adj (scatter-ADD
 * of edge_attr at [b = u/N, u%N, v%N]; every graph is padded to N nodes) followed by
 * label2onehot (utils.py:15-23).  COO arrays are int64 device pointers; `labels`
 * [B,N,N] int32 scratch, `a` [B,N,N,E] float32 out, `bad_count` (int32, device)
 * receives the number of entries whose summed label fell outside [0,E).            */
int dg_densify(const int64_t* edge_src, const int64_t* edge_dst, const int64_t* edge_attr, int64_t n_edges,
               int B, int N, int E, int* labels, float* a, int* bad_count, dg_stream_t stream);

/* dg_adamw_flat: one torch.optim.AdamW update (reference train.py:213-214,368,384;
 * decoupled weight decay, no amsgrad) over flat float32 buffers; `step` >= 1.       */
int dg_adamw_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                  dg_stream_t stream);

/* Same update with the step count in device memory (`step_counter`, int32, incremented by the
 * call): safe to capture into a hipGraph and replay.                                */
int dg_adamw_flat_devstep(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                          float lr, float beta1, float beta2, float eps, float weight_decay,
                          int* step_counter, dg_stream_t stream);

/* Skinny readout: nn.Linear(128 -> N), N <= 16, over edge / node rows (reference models.py:67-68,100-101: readout_e,
 * readout_n).  x [R,128] in the activation dtype (`dtype`), w [N,128], b [N] or NULL, y [R,N] float32 (the logits are
 * float32 in every activation mode); _dgrad: dx [R,128] (activation dtype) = dy [R,N] (float32) . w.  The weight
 * gradient is dg_linear_wgrad's N <= 16 variant.                                                                     */
int dg_skinny_linear_fwd(const void* x, const float* w, const float* b, float* y, int64_t R, int N, int K, int dtype,
                         dg_stream_t stream);
int dg_skinny_linear_dgrad(const float* dy, const float* w, void* dx, int64_t R, int N, int K, int dtype,
                           dg_stream_t stream);
/* dw [N,K], db [N] (nullable) from float32 dy [R,N] and x [R,K] of `dtype`; workspace as for dg_linear_wgrad. */
int dg_skinny_linear_wgrad(const float* dy, const void* x, float* dw, float* db, void* workspace, size_t workspace_bytes,
                           int64_t R, int N, int K, int dtype, dg_stream_t stream);

/* Tail of the Discriminator head (reference models.py:173-178, 207: node_mlp after its first Linear): rows [R,64] of
 * pre-activations z1 -> a1 = act(z1) [R,64], a2 = act(a1 W2^T + b2) [R,32], a3 = act(a2 W3^T + b3) [R,16], out = a3 W4^T + b4
 * [R,1] in ONE launch (6 on the BLAS + ATen).  w2 [32,64], w3 [16,32], w4 [1,16], float32; act: 0 ReLU, 1 LeakyReLU(0.01)
 * (other activations: not served, the caller keeps torch.nn.Sequential).
 *   dg_head_chain, m1 = m2 = m3 = NULL: the forward (o1..o4 = a1, a2, a3, out).
 *   dg_head_chain, m1..m3 = a1..a3 (the forward's activations), `in` = t [R,64]: the second order of the gradient penalty --
 *     o1 = t . act'(a1), o2 = (o1 W2^T) . act'(a2), o3 = (o2 W3^T) . act'(a3), o4 = o3 W4^T (biases unused); with t the
 *     adjoint of dg_head_bwd's g1, o4 is the adjoint of g_out and (g2, o1), (g3, o2), (g_out, o3) are dg_head_wgrad's operands
 *     for the adjoints of W2, W3, W4.
 *   dg_head_bwd: g3 = (g_out W4) . act'(a3) [R,16], g2 = (g3 W3) . act'(a2) [R,32], g1 = (g2 W2) . act'(a1) [R,64] = d z1.
 *   dg_head_wgrad: dw4 [1,16] = l4^T r4, dw3 [16,32] = l3^T r3, dw2 [32,64] = l2^T r2, db = column sums of l4, l3, l2
 *     (all three or NULL); first order: l = (g_out, g3, g2), r = (a3, a2, a1).  Rows are summed in ascending order by one
 *     thread per result element: bit-reproducible.                                                                    */
int dg_head_chain(const float* in, const float* m1, const float* m2, const float* m3, const float* w2, const float* b2,
                  const float* w3, const float* b3, const float* w4, const float* b4, float* o1, float* o2, float* o3,
                  float* o4, int64_t R, int act, dg_stream_t stream);
int dg_head_bwd(const float* g_out, const float* a1, const float* a2, const float* a3, const float* w2, const float* w3,
                const float* w4, float* g3, float* g2, float* g1, int64_t R, int act, dg_stream_t stream);
int dg_head_wgrad(const float* l4, const float* r4, const float* l3, const float* r3, const float* l2, const float* r2,
                  float* dw4, float* db4, float* dw3, float* db3, float* dw2, float* db2, int64_t R, dg_stream_t stream);

/* Node embedding (reference models.py:52-56, 154-158: node_layers = Linear(E, 64) - act - Linear(64, 128) - act - Dropout,
 * applied at :91 / :196) over the R = B N node rows, float32, E <= 16; act: 0 ReLU, 1 LeakyReLU(0.01).
 *   dg_embed_node_chain, m1 = m2 = NULL: o1 = act(in W1^T + b1) [R,64], o2 = act(o1 W2^T + b2) [R,128] in one launch.
 *   dg_embed_node_chain, m1 / m2 = the forward's o1 / o2, `in` = t [R,E]: the gradient penalty's second order --
 *     o1 = (t W1^T) . act'(m1), o2 = (o1 W2^T) . act'(m2) (biases unused): with t the adjoint of dg_embed_node_bwd's dz, o2 is
 *     the adjoint of its upstream gradient g, and g1^T t, g2^T o1 (dg_linear_wgrad) are the adjoints of W1, W2.
 *   dg_embed_node_bwd: g2 = g . act'(a2) [R,128], g1 = (g2 W2) . act'(a1) [R,64], dz = g1 W1 [R,E] (dz may be NULL).
 *     The parameter gradients are dg_linear_wgrad(g2, a1) and dg_linear_wgrad(g1, z).                               */
int dg_embed_node_chain(const float* in, const float* m1, const float* m2, const float* w1, const float* b1,
                        const float* w2, const float* b2, float* o1, float* o2, int64_t R, int E, int act,
                        dg_stream_t stream);
int dg_embed_node_bwd(const float* g, const float* a1, const float* a2, const float* w1, const float* w2, float* g2,
                      float* g1, float* dz, int64_t R, int E, int act, dg_stream_t stream);

/* dg_argmax_decode: reference inference.py:197-198 `torch.max(x, -1)[1]` on logits
 * [rows, E] -> uint8 labels [rows] (first maximum), so only bytes cross PCIe.      */
int dg_argmax_decode(const float* logits, int64_t rows, int E, unsigned char* out, dg_stream_t stream);

/* ---- opt-in kernel timing with HIP events (bench.py roofline) ----------------
 * dg_prof_enable(mask): bit k of `mask` set = every launch of kernel id k (enum below) is
 * bracketed by two events on the caller's stream (0 = off, -1 = all).  dg_prof_read()
 * synchronises the recorded events and returns the launches and their total device time. */
enum {
    DG_K_ATTN_FWD = 0,
    DG_K_ATTN_BWD = 1,
    DG_K_ATTN_BWD2 = 2,
    DG_K_LN_FWD = 3,
    DG_K_LN_BWD = 4,
    DG_K_LN_BWD2 = 5,
    DG_K_LINEAR_WGRAD = 6,
    DG_K_ROW_GEMM = 7,
    DG_K_EMBED_SYM = 8,
    DG_K_FFN = 9,          /* fused bf16 feed-forward kernels: forward, dx */
    DG_K_FFN_WGRAD = 10,   /* fused bf16 feed-forward weight-gradient kernels */
    DG_K_ATTN_HALF_FWD = 11,   /* fused attention half of the edge branch: forward */
    DG_K_ATTN_HALF_BWD = 12,   /* ... backward */
    /* row GEMMs over EDGE-level row counts (R >= DG_EDGE_ROWS), by shape; smaller launches stay in DG_K_ROW_GEMM */
    DG_K_ROW_GEMM_E_128 = 13,      /* 128 -> 128 (q/k/v/e/out projections and their input gradients) */
    DG_K_ROW_GEMM_E_N384 = 14,     /* 128 -> 384 (fc1; dh = dz W2) */
    DG_K_ROW_GEMM_E_K384 = 15,     /* 384 -> 128 (fc2 [+ residual + LayerNorm]; dx = dh W1) */
    /* weight gradients over EDGE-level row counts, by shape [N,K] of dW; smaller launches stay in DG_K_LINEAR_WGRAD */
    DG_K_LINEAR_WGRAD_E_128 = 16,  /* dW [128,128] (the six projections of an attention block) */
    DG_K_LINEAR_WGRAD_E_N384 = 17, /* dW [384,128] (fc1) */
    DG_K_LINEAR_WGRAD_E_K384 = 18, /* dW [128,384] (fc2) */
    /* fused bf16 feed-forward over NODE-level row counts (R < DG_EDGE_ROWS): DG_K_FFN / DG_K_FFN_WGRAD hold the edge-level launches */
    DG_K_FFN_NODE = 19,
    DG_K_FFN_WGRAD_NODE = 20,
    DG_K_FFN_F32 = 21,         /* fused float32 feed-forward forward (dg_ffn_ln_fwd_f32), edge-level launches */
    DG_K_FFN_F32_NODE = 22,    /* ... launches without an edge-level problem */
    DG_K_COUNT = 24
};
int dg_prof_enable(int mask);
int dg_prof_reset(void);
/* The row count from which a launch counts as EDGE-level (profiler keys above, traversal direction of the streaming kernels):
 * process-wide, default DG_EDGE_ROWS.  A caller that knows its batch sets it between the node-level (B N, 2 B N) and the
 * edge-level (B N^2) row counts of its step -- e.g. B N^2 / 2 -- so that a large batch's node-level launches (B = 2048:
 * 92 160 rows) are not filed under the edge-level keys.  rows <= 0 restores the default.                                    */
int     dg_set_edge_rows(int64_t rows);
int64_t dg_edge_rows(void);
int dg_prof_read(int kernel_id, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* DRUGGEN_HIP_H */
