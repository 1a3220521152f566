/*
 * tokenreduction_hip.h -- C ABI of the MI355X (gfx950) ViT-with-token-reduction hot path.
 *
 * The reference (JoakimHaurum/TokenReduction) is pure PyTorch and has NO native boundary
 * (SURVEY.md section 2a); its hot path is model.forward() of models/{deit_viz,topk,evit}.py.
 * This header is the native boundary this build introduces under the reference's Python
 * plugin API.  Each entry point names the reference lines it replaces.
 *
 * Conventions (every entry point):
 *   - all data pointers are DEVICE pointers owned by the caller (e.g. the PyTorch allocator);
 *     outputs and workspace are pre-allocated by the caller; nothing is allocated or freed here;
 *   - work is enqueued on the caller's stream (`tr_stream_t` = hipStream_t); no call synchronises,
 *     so every call is hipGraph-capturable; no global mutable state (re-entrant per stream);
 *   - bf16 tensors are passed as uint16_t* (raw bits); row-major, innermost dimension contiguous;
 *   - return value: 0 = ok, <0 = error (TR_ERR_*); tr_last_error() gives a thread-local message;
 *   - head_dim is 64 for every DeiT size (models_act.py:1087,1120,1153) and is required.
 */
#ifndef TOKENREDUCTION_HIP_H
#define TOKENREDUCTION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tr_stream_t; /* hipStream_t */

#define TR_OK 0
#define TR_ERR_SHAPE (-1)     /* unsupported / inconsistent shape */
#define TR_ERR_ALIGN (-2)     /* pointer or leading dimension not 16-byte aligned */
#define TR_ERR_NULL (-3)      /* required pointer is NULL */
#define TR_ERR_LAUNCH (-4)    /* hipLaunch failed (message has hipGetErrorString) */
#define TR_ERR_CONFIG (-5)    /* invalid tr_vit_config */

/* GEMM epilogues: out = epilogue(A[M,K] * W[N,K]^T + bias[N]) */
#define TR_EPI_BF16 0       /* out bf16 [M,N]                            (qkv Linear, topk.py:44)          */
#define TR_EPI_GELU_BF16 1  /* out bf16 [M,N] = gelu_erf(.)              (timm Mlp fc1+act)                */
#define TR_EPI_RESID_F32 2  /* out fp32 [M,N] += (.)  (in-place residual: topk.py:87 / :95)                */
#define TR_EPI_F32 3        /* out fp32 [M,N]                            (head, topk.py:203)               */
#define TR_EPI_PATCH_F32 4  /* out fp32 token rows: row m=(b,p) -> out[(b*(P+1)+1+p), :] = (.) + pos[1+p,:]
                               (PatchEmbed + pos_embed, topk.py:181-186); aux = pos_embed, aux_i = P        */

int tr_version(void);
const char* tr_last_error(void);

/* Launch profiler: between tr_profile_begin(stream) and tr_profile_end, every launch this thread enqueues through the entry points
 * below drops a HIP event on `stream`; tr_profile_end waits for the stream and returns, per launch group ("mark"), a label
 * (48 chars), the milliseconds since the previous mark (kernel + its dependent-launch boundary), and the group's algorithmic
 * FLOPs / bytes where the entry point states them (0 otherwise).  Returns the number of marks (only `max` are written).
 * Not capturable in a hipGraph.  This is how bench.py times the executor kernel by kernel. */
int tr_profile_begin(tr_stream_t s);
int tr_profile_end(int max, char* labels, float* ms, double* flops, double* bytes);

/* a1 (PatchEmbed, call site topk.py:181): unfold 16x16 patches.  img fp32 [B,C,H,W] ->
 * cols bf16 [B*(H/p)*(W/p), C*p*p], column order (c, iy, ix) = Conv2d weight.view(D,-1) order. */
int tr_im2col_bf16(const float* img, uint16_t* cols, int B, int C, int H, int W, int patch, tr_stream_t s);
int tr_im2col_f32(const float* img, float* cols, int B, int C, int H, int W, int patch, tr_stream_t s);   /* fp32 validation path */

/* a2 (topk.py:183-186): x[b*N + 0, :] = cls_token + pos_embed[0] for every image (fp32). */
int tr_cls_pos_rows(const float* cls_token, const float* pos_embed, float* x, int B, int N, int D, tr_stream_t s);

/* a1 + a2 in one launch (the eval forward's patch embedding; timm PatchEmbed + topk.py:181-186): x fp32 [B, P+1, D] with
 * x[b,0,:] = cls_token + pos_embed[0], x[b,1+p,:] = W . patch(b,p) + bias + pos_embed[1+p]; W bf16 [D, C*16*16] (Conv2d weight.view(D,-1)),
 * img fp32 [B,C,HW,HW].  The unfold happens on the way into the LDS (the bf16 column matrix of tr_im2col_bf16 is never written), one
 * workgroup per image and 384 output columns.  Same arithmetic as tr_im2col_bf16 + tr_gemm_bf16(TR_EPI_PATCH_F32) + tr_cls_pos_rows.
 * tr_patch_embed_supported: patch 16, HW %% 16 == 0, D %% 384 == 0 (other shapes: the three-launch path). */
int tr_patch_embed_supported(int C, int HW, int patch, int D);
int tr_patch_embed_bf16(const float* img, const uint16_t* W, const float* bias, const float* cls_token, const float* pos_embed, float* x,
                        int B, int C, int HW, int patch, int D, tr_stream_t s);

/* a3/a4 Linear layers (nn.Linear: y = x W^T + b, W is [N,K] row-major like the state dict).
 * A bf16 [M,K], W bf16 [N,K], bias fp32 [N]; K % 64 == 0, N % 4 == 0.  `out` dtype/meaning per epilogue;
 * aux/aux_i only for TR_EPI_PATCH_F32. */
int tr_gemm_bf16(const uint16_t* A, const uint16_t* W, const float* bias, void* out, const float* aux, int aux_i,
                 int M, int N, int K, int epilogue, tr_stream_t s);

/* a4, eval forward: the whole timm Mlp of a block (models/topk.py:78 construction, :95 `x = x + self.mlp(self.norm2(x))`; the residual
 * add stays in the next LayerNorm) in ONE launch: out bf16 [M,D] = fc2(gelu_erf(fc1(xn))) with the [M,Hd] hidden activation never
 * leaving the CU (csrc/tr_mlp_fused.hip).  Bit-identical to tr_gemm_bf16(TR_EPI_GELU_BF16) followed by tr_gemm_bf16(TR_EPI_BF16).
 * `packed`: the two weight matrices in fragment-major order and fc2's bias as an accumulator image, tr_mlp_pack_bytes(D,Hd) bytes,
 * written by tr_mlp_pack_bf16 from the bf16 [Hd,D] / [D,Hd] matrices and the fp32 [D] bias (repack whenever they change).  tr_mlp_fused_supported: D == 384, Hd %% 32 == 0 (other widths: the pair). */
int tr_mlp_fused_supported(int D, int Hd);
/* Which Mlp the eval executor runs where tr_block_weights.mlp_pk is given: 1 = the fused launch wherever the shape is supported, 0 = never,
 * -1 (default) = for more blocks of 128 rows than the device has compute units (stream-K schedule) and where a single round of blocks fills
 * at least three quarters of them (one workgroup per CU: hipDeviceAttributeMultiprocessorCount of the current device, 256 on MI355X; the
 * two Mlp forms are bit-identical, so this is a speed choice only).  Process-wide; returns the previous mode. */
int tr_set_mlp_fused(int mode);
size_t tr_mlp_pack_bytes(int D, int Hd);
int tr_mlp_pack_bf16(const uint16_t* fc1_w, const uint16_t* fc2_w, const float* fc2_b, void* packed, int D, int Hd, tr_stream_t s);
/* scratch (nullable; tr_mlp_fused_scratch_bytes(D,Hd) bytes for the CURRENT device, 16-byte aligned): with it a launch of more blocks of
 * 128 rows than the device has compute units deals its steps evenly over the workgroups (stream-K: a block that straddles two workgroups
 * hands its fp32 accumulator over, exactly); without it whole blocks round-robin.  Same bits either way.
 * A scratch belongs to ONE launch at a time: it holds that launch's accumulator slots and hand-over counters (zeroed by a memset node in
 * front of the kernel), so two launches that may overlap -- other streams, other threads -- need a scratch each.
 * A consumer workgroup polls for its predecessor's accumulator (bounded: seconds; tr_set_mlp_poll_max(iterations), negative = the default
 * again, 0 = every hand-over counts as abandoned (tests), returns the previous bound); a poll
 * that runs out does not stop the process: the launch finishes on whatever the slot held and leaves an error record in the scratch, which
 * tr_mlp_fused_status (waits for the stream, reads and clears the record) turns into TR_ERR_LAUNCH. */
size_t tr_mlp_fused_scratch_bytes(int D, int Hd);
int tr_set_mlp_poll_max(int iterations);
int tr_mlp_fused_status(void* scratch, size_t scratch_bytes, int D, int Hd, tr_stream_t s);
int tr_mlp_fused_bf16(const uint16_t* xn, const void* packed, const float* fc1_b, uint16_t* out, void* scratch, size_t scratch_bytes, int M, int D,
                      int Hd, tr_stream_t s);
/* topk.py:95's `self.mlp(self.norm2(x))` in ONE launch: out bf16 [M,D] = fc2(gelu(fc1(LayerNorm(x + delta; g, b, eps)))) with x the fp32
 * residual stream [M,D] and delta the pending bf16 residual of the attention branch [M,D] (neither is written): the kernel's fc1 waves
 * normalise their rows in registers on the way in.  Bit-identical to tr_layernorm2_bf16(x, .., NULL, .., delta, .., NULL, .., g, b, xn, ..)
 * followed by tr_mlp_fused_bf16(xn, ..): the LayerNorm launch (8 B per element through HBM) and the bf16 rows between the two are gone.
 * tr_set_mlp_ln: where the eval executor takes this form in place of a lazy norm2 followed by the fused Mlp -- 1 (default): where that launch
 * is one round of whole blocks (measured faster there, slower under the stream-K schedule, which normalises a block once per workgroup that
 * touches it) and everywhere in a forward marked `concurrent` (those launches run whole blocks), 2: wherever the fused Mlp runs, 0: never;
 * returns the previous setting. */
int tr_set_mlp_ln(int mode);
int tr_mlp_fused_ln_bf16(const float* x, const uint16_t* delta, const float* g, const float* b, float eps, const void* packed,
                         const float* fc1_b, uint16_t* out, void* scratch, size_t scratch_bytes, int M, int D, int Hd, tr_stream_t s);
/* a3/a4, eval forward: the START of a block in one launch -- topk.py:86-87 `self.attn(self.norm1(x))` with :44 `self.qkv(x)`, and the previous
 * block's pending residual adds folded in:  v = (x [+ d1]) [+ d2]  (fp32 stream row [M,D] + pending bf16 residuals [M,D], in that order:
 * d1 = attention branch, d2 = Mlp branch of the previous block; d2 needs d1),  x_out[M,D] = v  (out of place, x_out != x; NULL exactly when
 * d1 is NULL: nothing pending, nothing rewritten),  out bf16 [M,N] = LayerNorm(v; g, b, eps) . W^T + bias.
 * Bit-identical to tr_layernorm_bf16 / tr_layernorm2_bf16 followed by tr_gemm_bf16(TR_EPI_BF16) -- and without the LayerNorm launch's
 * 12-14 bytes per element at the memory roof: four of the kernel's twelve waves normalise the next 128-row block while the other eight
 * multiply the current one (csrc/tr_lnlin.hip).  tr_lnlin_supported: D == 384, N %% 64 == 0, 128 <= N <= 4096.
 * `packed`: W [N,D] (nn.Linear layout, bf16) in fragment-major order, tr_lnlin_pack_bytes(D,N) bytes, written by tr_lnlin_pack_bf16 (repack
 * whenever W changes).  `scratch`: tr_lnlin_scratch_bytes(D,N) bytes for the current device (two 96-KiB row slots per compute unit), one
 * launch at a time like tr_mlp_fused_bf16's. */
int tr_lnlin_supported(int D, int N);
size_t tr_lnlin_pack_bytes(int D, int N);
size_t tr_lnlin_scratch_bytes(int D, int N);
int tr_lnlin_pack_bf16(const uint16_t* W, void* packed, int D, int N, tr_stream_t s);
int tr_lnlin_bf16(const float* x, const uint16_t* d1, const uint16_t* d2, float* x_out, const float* g, const float* b, float eps,
                  const void* packed, const float* bias, uint16_t* out, void* scratch, size_t scratch_bytes, int M, int D, int N, tr_stream_t s);
/* The tail of a block and the head of the next in ONE launch (topk.py:95 `x = x + self.mlp(self.norm2(x))`, then the next block's :87
 * `self.norm1(x)`):  x[M,D] (fp32 stream, already holding the attention branch's residual) += fc2(gelu(fc1(xn))) + fc2_b, IN PLACE, and
 * xn_next[M,D] (bf16, != xn) = LayerNorm(x; next_g, next_b, eps).  The kernel's fc2 wave owns whole rows in registers: its accumulator
 * starts at the stream row, so the fc2 output is never rounded to bf16 on its way into the stream, and the row's mean / variance are taken in
 * registers (two passes, tr_layernorm's formulas).  Replaces fc1, fc2 and the residual-add + LayerNorm launch behind them; not bit-identical
 * to that sequence (one rounding fewer), same tolerance class.  packed / scratch as for tr_mlp_fused_bf16 (fc2_b is passed again: the bias
 * image inside `packed` is not used here). */
/* Whether the eval executor runs that form where it runs the fused Mlp and the next block starts with a plain norm1 (1) or keeps
 * tr_mlp_fused_bf16 + the LayerNorm launch (0 = default: measured, the one-launch form is 4 % slower on the headline forward -- its
 * epilogue stalls the workgroup; profiles/r05_mlp_lab.md).  Process-wide; returns the previous setting. */
int tr_set_mlp_resid_ln(int on);
int tr_mlp_fused_resid_ln_bf16(const uint16_t* xn, const void* packed, const float* fc1_b, const float* fc2_b, float* x, const float* next_g,
                               const float* next_b, float eps, uint16_t* xn_next, void* scratch, size_t scratch_bytes, int M, int D, int Hd,
                               tr_stream_t s);

/* a4 nn.LayerNorm(D, eps) rows of the fp32 residual stream -> bf16 (topk.py:86 norm1, :95 norm2, :201 norm), with the
 * PENDING residual add folded in: if delta != NULL (bf16 rows at stride ldd: the output of attn.proj / mlp.fc2),
 * x[row] += delta[row] is written back first (`x = x + drop_path(...)`, topk.py:87 / :95), then y[row] = LN(x[row]).
 * x fp32 rows at stride ldx (floats); y bf16 [M,D].  D % 4 == 0, D <= 1024. */
int tr_layernorm_bf16(float* x, long ldx, const uint16_t* delta, long ldd, const float* gamma, const float* beta, uint16_t* y,
                      int M, int D, float eps, tr_stream_t s);
int tr_layernorm_f32(float* x, long ldx, const float* delta, long ldd, const float* gamma, const float* beta, float* y, int M,
                     int D, float eps, tr_stream_t s);                                                  /* fp32 validation path */
/* Same, out of place: x_out[row] = x[row] + delta[row] (rows at stride ldxo; x_out == x is the in-place form above).  The training
 * forward uses it so that the input of every norm survives for the backward pass. */
int tr_layernorm_bf16_to(const float* x, long ldx, float* x_out, long ldxo, const uint16_t* delta, long ldd, const float* gamma,
                         const float* beta, uint16_t* y, int M, int D, float eps, tr_stream_t s);
/* y = LayerNorm((x + delta) + delta2) with TWO pending residuals (delta2 nullable), the sum written to x_out -- or, x_out == NULL, not
 * written at all.  The eval executor's "lazy norm2": a norm2 that no reduction follows normalises x + d_attn without storing it, the
 * next norm1 adds d_attn and d_mlp in the reference's order (topk.py:87, :95) and writes the stream once: bit-identical to two
 * tr_layernorm_bf16 calls, 22 instead of 24 bytes per element and block. */
int tr_layernorm2_bf16(const float* x, long ldx, float* x_out, long ldxo, const uint16_t* delta, long ldd, const uint16_t* delta2, long ldd2,
                       const float* gamma, const float* beta, uint16_t* y, int M, int D, float eps, tr_stream_t s);

/* out[i] = x[i] + delta[i] for n elements (delta nullable; bf16, fp32 when delta_is_f32): the residual stream after a block as
 * the reference's viz_data["Features"] records it (topk.py:197) -- x itself absorbs the pending mlp output only in the next norm. */
int tr_residual_snapshot(const float* x, const void* delta, int delta_is_f32, float* out, size_t n, tr_stream_t s);
/* dst fp32 [B,N] = src fp32 [N] in every row (the Heuristic family's per-block key mask, heuristic.py:247-258). */
int tr_broadcast_rows(const float* src, float* dst, int B, int N, tr_stream_t s);

/* a3 (topk.py:44-51 == deit_viz.py:43-51): softmax(q k^T / sqrt(64)) v for every (image, head).
 * qkv bf16 [B*N, 3*H*64] (columns [q|k|v], head-major), out bf16 [B*N, H*64].
 * cls_rows (nullable) fp32 [B,H,N]: the CLS query's softmax row (attn[:, :, 0, :], topk.py:59) --
 * the only part of the N x N matrix the reduction reads, so the matrix is never materialised.
 * size (nullable) fp32 [B,N]: ToMe's proportional attention (Attention_ToMe.forward tome.py:48-49): log(size[key]) is added
 * to every query's logit for that key; a 1/0 key mask works the same way (ATS, ats.py:117-120: log 0 = -inf -> weight 0).
 * colsum_part (nullable) fp32 [B,H,4,N]: column sums of the softmax matrix, one partial per wave of the workgroup -- summed
 * over (H, 4) they are K-Medoids' token weights sum_h sum_q attn[b,h,q,:] (kmedoids.py:240); partials keep the summation
 * order fixed (no float atomics).
 * Any N: up to 224 tokens a query's whole score row sits in registers (attention16_kernel); beyond (384 x 384 inputs: 577) keys pass
 * through the LDS in chunks of 128 with an online softmax; column sums together with a key bias need N <= 608. */
int tr_attention_bf16(const uint16_t* qkv, uint16_t* out, float* cls_rows, const float* size, float* colsum_part, int B, int N,
                      int H, tr_stream_t s);
/* fp32 validation path (N <= 640; K/V of a head in LDS up to N = 256, read from L2 beyond): same contract in the reference's
 * arithmetic (expf softmax, fp32 everywhere). */
int tr_attention_f32(const float* qkv, float* out, float* cls_rows, const float* size, float* colsum_part, int B, int N, int H,
                     tr_stream_t s);
/* TR_PREC_BF16X3: same contract, q k^T and P v as split-bf16 products on the matrix cores (fp32 softmax, whole row in registers);
 * N <= 224 on MFMA, longer sequences are forwarded to tr_attention_f32. */
int tr_attention_split(const float* qkv, float* out, float* cls_rows, const float* size, float* colsum_part, int B, int N, int H,
                       tr_stream_t s);
/* a11 (forward only): Policy_Attention.forward dyvit.py:53-67 with softmax_with_policy :39-51 -- the attention of DyViT's
 * TRAINING forward, where pruned tokens stay in the sequence and are masked by policy fp32 [B,N] of 1/0:
 *   attn = (exp(s - max_k s) * pol + eps/N) / (sum_k exp(s - max_k s) * pol + eps),  pol[q][k] = policy[k], 1 for k == q,
 * eps = 1e-6.  bf16: any N (beyond 224 tokens -- 384 x 384 inputs -- an online-softmax kernel over 128-key chunks); fp32: N <= 256.
 * Used by the DyViT training executor (backward: tr_attention_policy_bwd_bf16 / tr_attention_policy_bwd_long_bf16). */
int tr_attention_policy_bf16(const uint16_t* qkv, uint16_t* out, const float* policy, int B, int N, int H, tr_stream_t s);
int tr_attention_policy_f32(const float* qkv, float* out, const float* policy, int B, int N, int H, tr_stream_t s);

/* a6 (topk.py:55-65 == evit.py:77-87) + a8 (evit.py:25-46 complement_idx):
 * scores[b,j] = mean_h cls_rows[b,h,1+j] (j < P = N-1); idx[b,:K] = indices of the K largest scores in
 * DESCENDING score order (ties: lowest index first -- torch.topk's CPU tie order is unspecified);
 * compl (nullable) [B,P-K] = indices not selected, ascending.  scores (nullable) fp32 [B,P]. int32 indices. */
int tr_cls_topk(const float* cls_rows, int32_t* idx, int32_t* compl_idx, float* scores, int B, int H, int N, int K,
                tr_stream_t s);

/* a7 (topk.py:89-93) [+ a9 evit.py:111-123] fused with norm2 (topk.py:95):
 * with x := x + delta when delta != NULL (bf16 [B,N,D], the pending attn.proj output -- topk.py:87 precedes the gather):
 * x_out[b,0]=x[b,0]; x_out[b,1+r]=x[b,1+idx[b,r]] (r<K); if compl_idx: x_out[b,K+1]=sum_j x[b,1+compl[b,j]]*scores[b,compl[b,j]].
 * Also y = LayerNorm(x_out) in bf16.  x fp32 [B,N,D] -> x_out fp32 [B,N_out,D], y bf16 [B,N_out,D],
 * N_out = K+1 (+1 with fuse).  idx == NULL means identity (N_out = N, x_out may be NULL -> only y written). */
int tr_gather_layernorm_bf16(const float* x, const uint16_t* delta, const int32_t* idx, const int32_t* compl_idx, const float* scores,
                             const float* gamma, const float* beta, float* x_out, uint16_t* y, int B, int N, int K,
                             int D, float eps, tr_stream_t s);
int tr_gather_layernorm_f32(const float* x, const float* delta, const int32_t* idx, const int32_t* compl_idx, const float* scores,
                            const float* gamma, const float* beta, float* x_out, float* y, int B, int N, int K, int D,
                            float eps, tr_stream_t s);                                                  /* fp32 validation path */

/* The training forward's Linear + GELU (timm Mlp fc1, the DyViT / SiT predictor layers): pre bf16 [M,N] = A W^T + bias stays for the
 * backward, h bf16 [M,N] = gelu(pre) on the ROUNDED pre-activation -- bitwise tr_gemm_bf16(TR_EPI_BF16) followed by tr_gelu_bf16, in
 * one launch.  K % 64 == 0, N % 8 == 0. */
int tr_gemm_gelu_keep_bf16(const uint16_t* A, const uint16_t* W, const float* bias, uint16_t* pre, uint16_t* h, int M, int N, int K,
                           tr_stream_t s);

/* Backward of Linear -> GELU's second half (timm Mlp fc2 -> act, the DyViT / SiT predictor layers): out bf16 [M,N] =
 * bf16(A W^T) * gelu'(pre) (a data gradient: no bias), pre bf16 [M,N] the pre-activation tr_gemm_gelu_keep_bf16 kept (engine.py:76 loss.backward() through
 * timm Mlp) -- bitwise tr_gemm_bf16(TR_EPI_BF16) followed by tr_gelu_bwd_bf16, in one launch.  K % 64 == 0, N % 8 == 0. */
int tr_gemm_dgelu_bf16(const uint16_t* A, const uint16_t* W, const uint16_t* pre, uint16_t* out, int M, int N, int K, tr_stream_t s);

/* fp32 validation path of the Linear layers: A fp32 [M,K], W fp32 [N,K], out fp32; K % 16 == 0.
 * epilogue: TR_EPI_F32 (bias), TR_EPI_GELU_BF16 (bias + exact-erf GELU, fp32 out), TR_EPI_PATCH_F32. */
int tr_gemm_f32(const float* A, const float* W, const float* bias, float* out, const float* aux, int aux_i, int M, int N, int K,
                int epilogue, tr_stream_t s);
/* TR_PREC_BF16X3: the same Linear (same operands, epilogues and fp32 output) on the matrix cores -- every fp32 operand is split
 * into hi = bf16(v), lo = bf16(v - hi) and a product is taken as hi*hi + hi*lo + lo*hi in the fp32 MFMA accumulator (~2^-17
 * relative per product).  K % 32 == 0 and N % 4 == 0 run on MFMA; other shapes are forwarded to tr_gemm_f32. */
int tr_gemm_split(const float* A, const float* W, const float* bias, float* out, const float* aux, int aux_i, int M, int N, int K,
                  int epilogue, tr_stream_t s);

/* ---- reducers that run before a block (csrc/tr_prune.hip) ------------------------------------------------------------
 * tr_pool_broadcast: PredictorLG.forward dyvit.py:115-118 with policy == 1 (eval): h [B,N,C] (bf16, or fp32 when is_f32), in
 *   place: channels C/2..C-1 of every row become mean over the image's PATCH rows (1..N-1) of that channel + eps.
 * tr_dyvit_score: out_conv.4 + LogSoftmax + [:,:,0] (dyvit.py:108-109,231): h [M,C] -> scores fp32 [M]; w fp32 [2,C], bias [2].
 * tr_sit_merge: TokenSlimmingModule.forward sit.py:37-39: logits fp32 [B,N,ldl] (row 0 of an image = CLS, ignored; first K
 *   columns used), softmax(logits*scale) over the patch-token axis, x_out[b,1+k,:] = sum_p w[b,p,k] x[b,1+p,:]; x_out[b,0] =
 *   x[b,0]; x, x_out fp32 [B,N,D] / [B,K+1,D].  soft (nullable): fp32 [B,K,N-1].
 * tr_rownorm: F.normalize(x, dim=-1) (sinkhorn.py:70): x fp32 [M,D] -> xh fp32 [M,D] and the same rows as a GEMM operand
 *   (bf16, fp32 when lp_is_f32).
 * tr_sinkhorn: log_optimal_transport (sinkhorn.py:41-56) per image on scores fp32 [B,N,ldl] (row 0 = CLS ignored; columns
 *   0..K-1 = token . centre): wt (may alias scores) gets the transport plan token-major [B,N,ldl], soft (nullable) the same
 *   cluster-major [B,K,N-1] (Soft_Assignment_Maps).  K*(N-1) floats stay in LDS when they fit, else in the scores buffer.
 * tr_weighted_merge: x_out[b,1+k,:] = sum_p wt[b,1+p,k] * src[b,1+p,:]; x_out[b,0] = x[b,0]   (sinkhorn.py:83 with
 *   src = unit-norm tokens). */
int tr_pool_broadcast(void* h, int is_f32, int B, int N, int C, float eps, tr_stream_t s);
int tr_rownorm(const float* x, float* xh, void* xh_lp, int lp_is_f32, int M, int D, tr_stream_t s);
int tr_sinkhorn(const float* scores, int ldl, float eps, int iters, float* wt, float* soft, int B, int N, int K, tr_stream_t s);
int tr_weighted_merge(const float* wt, int ldl, const float* x, const float* src, float* x_out, int B, int N, int K, int D,
                      tr_stream_t s);
int tr_dyvit_score(const void* h, int is_f32, const float* w, const float* bias, float* scores, int M, int C, tr_stream_t s);
int tr_sit_merge(const float* logits, int ldl, float scale, const float* x, float* x_out, float* soft, int B, int N, int K,
                 int D, tr_stream_t s);
/* tr_sit_merge with the summed rows taken from `src` fp32 [B,N,D] instead of x (PatchMerger sums the LayerNorm-ed tokens,
 * patchmerger.py:36-39); the CLS row still comes from x. */
int tr_softassign_merge(const float* logits, int ldl, float scale, const float* x, const float* src, float* x_out, float* soft,
                        int B, int N, int K, int D, tr_stream_t s);
/* Same result on MFMA (what the bf16 executor uses; K <= 192): apply_softmax != 0 first turns `logits` IN PLACE into the
 * token-axis softmax of logits*scale (and writes `soft` if given), then x_out[b,1+k,:] = sum_p logits[b,1+p,k] * src[b,1+p,:]
 * with both operands split into bf16 hi + lo (relative error ~2^-16).  apply_softmax == 0: `logits` already holds the
 * weights (Sinkhorn's transport plan). */
int tr_softassign_merge_fast(float* logits, int ldl, float scale, int apply_softmax, const float* x, const float* src, float* x_out,
                             float* soft, int B, int N, int K, int D, tr_stream_t s);
/* ---- DPC-KNN (csrc/tr_cluster.hip) --------------------------------------------------------------------------------------
 * tr_dpcknn_cluster: cluster_dpc_knn dpcknn.py:44-100 (token_mask=None) on the patch rows of x fp32 [B,N,D] (row 0 = CLS,
 *   ignored): centers int32 [B,K] = topk(score, K) in descending-score order (index_down), idx_cluster int32 [B,N-1],
 *   scores fp32 [B,N-1] (distance-to-denser-token * density).  noise (nullable) fp32 [B,N-1]: the uniform [0,1) draws the
 *   reference adds * 1e-6 to the densities (dpcknn.py:71-72); NULL = none.  k = nearest neighbours (args.k_neighbors).
 *   ws: tr_dpcknn_workspace_floats(B,N) floats of scratch.  fast_dist != 0 (also tr_kmedoids): the Gram product of the
 *   distance matrix runs on MFMA with hi/lo-split bf16 operands (relative error ~2^-16) instead of fp32 VALU -- what the bf16
 *   executor uses; 0 = the reference's fp32 arithmetic (validation executor).  For tr_dpcknn_cluster: 1 = the ONE-LAUNCH kernel
 *   below wherever it applies (else the staged launches), 2 = always the staged launches (sqnorm, distances to ws, density,
 *   parent distance, top-K, assignment) -- kept for P > 208 and as the comparison of the tests.
 * tr_dpcknn_cluster_fused (csrc/tr_cluster_fused.hip): the same clustering in one launch, one workgroup per image, the distance
 *   matrix held in LDS as its upper triangle (26 <= N-1 <= 208, D % 32 == 0, D <= 1024, k <= 5: tr_dpcknn_fused_supported);
 *   no workspace.  Identical to the staged launches except that element (i,j) below the diagonal is read from (j,i), which the
 *   staged matrix matches only up to the last bit.
 * tr_cluster_merge_layernorm: merge_tokens dpcknn.py:103-132 with token_weight = exp(x . score_w + score_b) (CTM, :155-157;
 *   score_w NULL = equal weights), then LayerNorm(gamma, beta, eps) of the merged tokens: x_out fp32 [B,K+1,D] (row 0 = CLS
 *   copied), y = LN(x_out) bf16 (fp32 when y_is_f32).  w_ws: [B,N-1] floats of scratch (token weights).
 * tr_kmedoids: k_medoids_fit kmedoids.py:40-85 with token weights (the equal_weight branch draws from the numpy global RNG and
 *   is not built): colsum_part fp32 [B,H,4,N] from the PREVIOUS block's attention -> w = its sum over (H,4) on the patch
 *   columns; initial medoids = topk(w, K); `iters` rounds of {assign to the nearest medoid; medoid k = the member with the
 *   smallest w_i * sum_j dist_ij, index 0 if the cluster is empty (kmedoids.py:74-79)}; final assignment.
 *   centers int32 [B,K] (cluster_idx), assign int32 [B,N-1].  ws: tr_dpcknn_workspace_floats(B,N) floats. */
size_t tr_dpcknn_workspace_floats(int B, int N);
int tr_kmedoids(const float* x, const float* colsum_part, float* ws, int32_t* centers, int32_t* assign, int B, int N, int D, int H,
                int K, int iters, int fast_dist, tr_stream_t s);
/* k_medoids_fit with token_weight = None (args.equal_weight, kmedoids.py:43-58): the first medoid is init_idx for every image -- the
 * reference draws it with np.random.choice on the host, so it is an input --, the others by the farthest-point rule of :47-56,
 * then the same iterations with unit weights. */
int tr_kmedoids_equal(const float* x, int init_idx, float* ws, int32_t* centers, int32_t* assign, int B, int N, int D, int K, int iters,
                      int fast_dist, tr_stream_t s);
int tr_dpcknn_cluster(const float* x, const float* noise, float* ws, int32_t* centers, int32_t* idx_cluster, float* scores,
                      int B, int N, int D, int K, int k, int fast_dist, tr_stream_t s);
int tr_dpcknn_fused_supported(int N, int D, int k);
int tr_dpcknn_cluster_fused(const float* x, const float* noise, int32_t* centers, int32_t* idx_cluster, float* scores, int B, int N,
                            int D, int K, int k, tr_stream_t s);
int tr_cluster_merge_layernorm(const float* x, const float* score_w, const float* score_b, float* w_ws,
                               const int32_t* idx_cluster, const float* gamma, const float* beta, float* x_out, void* y,
                               int y_is_f32, int B, int N, int K, int D, float eps, tr_stream_t s);

/* ---- ATS (csrc/tr_ats.hip) ----------------------------------------------------------------------------------------------
 * tr_ats_sample: AdaptiveTokenSampling.forward ats.py:52-84.  cls_rows fp32 [B,H,N] (softmax row of the CLS query),
 *   qkv [B*N, 3*H*64] (bf16, or fp32 when qkv_is_f32; only V is read), mask (nullable = all valid) fp32 [B,N] of 1/0,
 *   steps fp32 [n_steps] = the module's sample_steps (ats.py:48).  ids int32 [B,K]: id 0 (CLS) first, then the sorted unique
 *   sampled token ids (1-based positions), then 0 padding up to the static bound K; new_mask fp32 [B,K] of 1/0
 *   (ids != 0, CLS 1).  cdf_out (nullable) fp32 [B,N-1]: the cdf the samples were taken on.
 * tr_ats_gather: x_out[b,t] = x[b, ids[b,t]] (fp32 [B,K,D], ats.py:157) and ao_out[b,t] = ao[b, ids[b,t]] (the rows of
 *   attn @ v the sampled queries keep, ats.py:86,129); ao bf16 (fp32 when ao_is_f32). */
int tr_ats_sample(const float* cls_rows, const void* qkv, int qkv_is_f32, const float* mask, const float* steps, int n_steps,
                  int32_t* ids, float* new_mask, float* cdf_out, int B, int N, int H, int K, tr_stream_t s);
int tr_ats_gather(const float* x, const void* ao, int ao_is_f32, const int32_t* ids, float* x_out, void* ao_out, int B, int N,
                  int K, int D, tr_stream_t s);
/* The reference's dynamic width (ats.py:77-78: pad_sequence pads the unique ids to the BATCH maximum, so the token count after a sampling
 * block is 1 + max unique ids, not the static bound K).  tr_ats_width: *width (device int32) = max over the images of their valid ids
 * (row sums of new_mask [B,K]); tr_ats_narrow: the first Kw columns of ids / new_mask as contiguous [B,Kw] arrays, the form tr_ats_gather
 * and the next block's key mask take.  Used by tr_vit_forward when tr_vit_config.ats_dynamic is set. */
int tr_ats_width(const float* new_mask, int32_t* width, int B, int K, tr_stream_t s);
int tr_ats_narrow(const int32_t* ids, const float* new_mask, int32_t* ids_out, float* mask_out, int B, int K, int Kw, tr_stream_t s);

/* a13 bipartite_soft_matching (tome.py:230-277, class_token=True) on metric = k.mean(1) (tome.py:58), read straight from the
 * K third of qkv ([B*N, 3*H*64]; bf16, or fp32 when qkv_is_f32).  Tokens at even positions form set A (CLS = A[0], never
 * merged), odd positions set B.  Outputs (int32): src_idx [B,r] = the r A-tokens with the largest best-match score, in
 * descending order; dst_idx [B,r] = the B-token each is merged into; unm_idx [B, ceil(N/2)-r] = the other A-tokens, ascending.
 * 3 <= N <= 600, 1 <= r <= (N-1)/2 (tome.py:253). */
int tr_tome_match(const void* qkv, int qkv_is_f32, int32_t* unm_idx, int32_t* src_idx, int32_t* dst_idx, int B, int N, int H,
                  int r, tr_stream_t s);

/* a14 merge_wavg (tome.py:309-323) over the merge closure (tome.py:279-289), with the pending residual (delta: bf16, or fp32
 * when f32_path) added first and norm2 (tome.py:101) applied after: x_out [B,N-r,D] = merge((x+delta)*size)/merge(size) in the
 * order [unmerged A-tokens | all B-tokens]; size_out [B,N-r]; y = LayerNorm(x_out) (bf16, or fp32 when f32_path).
 * size_in == NULL means all ones (first merge, tome.py:318-319). */
int tr_tome_merge_layernorm(const float* x, const void* delta, int f32_path, const float* size_in, const int32_t* unm_idx,
                            const int32_t* src_idx, const int32_t* dst_idx, const float* gamma, const float* beta, float* x_out,
                            float* size_out, void* y, int B, int N, int r, int D, float eps, tr_stream_t s);

/* ---- backward kernels (csrc/tr_backward.hip, csrc/tr_attention_bwd.hip) -----------------------------------------------------
 * The reference's training step is `loss.backward()` through the eager forward (engine.py:50-76, mp_scaler.py:10-11); these are
 * the hand-written gradients of the ops above.  Reductions over tokens are two-stage with a fixed order (bitwise reproducible).
 * `accumulate` != 0: the gradient is ADDED to the destination (gradient accumulation, engine.py:41), else it is overwritten.
 * tr_wgrad_bf16: nn.Linear weight gradient dW[N,K] (+)= dY[M,N]^T X[M,K]; dY, X bf16 at row strides ldy, ldx (elements); yskip > 0:
 *   dY row m is row m + m/yskip + 1 of its tensor (the patch rows of a [B, P+1, D] tensor, yskip = P: PatchEmbed's weight gradient).
 *   ws: tr_wgrad_workspace_floats(M,N,K) floats (partials of the token splits).  N, K, ldy, ldx multiples of 8.
 * tr_colsum_bf16: nn.Linear bias gradient db[N] (+)= sum_m dY[m,n]; ws: tr_colsum_workspace_floats(M,N) floats.
 * tr_gelu_bf16 / tr_gelu_bwd_bf16: h = gelu(pre) with the SAME fit as TR_EPI_GELU_BF16 (the training forward stores fc1's
 *   pre-activation, then applies this); dh := dh * gelu'(pre) in place (exact erf form).
 * tr_layernorm_bwd: dy bf16 [M,D] (gradient wrt the LayerNorm output), x fp32 rows (the saved LayerNorm INPUT, row stride ldx),
 *   g_in (nullable) fp32 rows: gradient already flowing in the residual stream at these rows; g_out = g_in + dLN -> fp32 rows
 *   (stride ldgo) and gb_out (nullable) bf16 [rows, D].  d_gamma, d_beta fp32 [D].  idx != NULL: the rows are the gathered rows
 *   [B, n_in] of a Top-K block (n_in = K+1, or K+2 with EViT's fused token); row r of image b is written to row
 *   (r == 0 ? 0 : 1 + idx[b,r-1]) of [B, n_out, D] and the fused row to g_fused fp32 [B, D] (the caller zero-fills g_out/gb_out).
 *   ws: tr_layernorm_bwd_workspace_floats(M, D) floats.
 * tr_attention_bwd_bf16: d qkv from d out (see csrc/tr_attention_bwd.hip), N <= 224.
 * tr_head_bwd: dxn bf16 [B,D] = dlogits W; dW (+)= dlogits^T xn; db (+)= colsum(dlogits)   (topk.py:203; W, xn bf16); the weight and
 *   bias gradients run through tr_wgrad_bf16 / tr_colsum_bf16 on dl16, a bf16 copy of dlogits ([B,C] scratch); ws as tr_wgrad_bf16.
 * tr_embed_bwd: d pos_embed [N,D] (+)= sum_b g[b,n,:], d cls_token [D] (+)= sum_b g[b,0,:]   (topk.py:183-186).
 * tr_evit_fuse_bwd: evit.py:117-120: g_out[b,1+c,:] = scores[b,c] g_fused[b] (fp32 + bf16 copy) for the complement tokens c,
 *   dscore[b,1+c] = <x[b,1+c] + delta[b,1+c], g_fused[b]>  (dscore fp32 [B,N], zero-filled by the caller).
 * tr_tome_merge_bwd: merge_wavg (tome.py:309-323): g_out[b,i,:] = size_in[b,i] / size_out[b,o(i)] * g_merged[b,o(i),:];
 *   inv_map int32 [B,N] scratch. */
size_t tr_wgrad_workspace_floats(int M, int N, int K);
int tr_wgrad_bf16(const uint16_t* dY, long ldy, int yskip, const uint16_t* X, long ldx, float* dW, int accumulate, float* ws,
                  size_t ws_floats, int M, int N, int K, tr_stream_t s);
/* Both parameter gradients of an nn.Linear in one pass over dY (the bias sums come out of the weight-gradient kernel's staging
 * registers) and one reduce launch; ws as tr_wgrad_bf16. */
int tr_linear_bwd_params(const uint16_t* dY, long ldy, int yskip, const uint16_t* X, long ldx, float* dW, float* db, int accumulate,
                         float* ws, size_t ws_floats, int M, int N, int K, tr_stream_t s);

/* Parameter gradients of TWO Linear layers at once -- one weight-gradient launch and one reduce launch for both (the backward executor
 * pairs fc2 + fc1 and proj + qkv of a block; engine.py:76).  Each layer as in tr_linear_bwd_params with yskip = 0; results identical to
 * two separate calls up to the (fixed, reproducible) summation order over token ranges. */
/* ... and of up to FOUR (what the backward executor does with fc2, fc1, proj and qkv of a block when nothing rewrites their dY in between):
 * layer i: dW_i [N,K] (+)= dY_i[M, ldy]^T X_i[M, ldx], db_i [N] (+)= column sums of dY_i. */
typedef struct tr_linear_grad {
  const uint16_t* dY; long ldy;
  const uint16_t* X;  long ldx;
  float* dW; float* db;
  int M, N, K;
} tr_linear_grad;
size_t tr_linear_bwd_group_workspace_floats(const tr_linear_grad* layers, int n);      /* only M, N, K of each layer are read */
int tr_linear_bwd_group(const tr_linear_grad* layers, int n, int accumulate, float* ws, size_t ws_floats, tr_stream_t s);
size_t tr_linear_bwd_params2_workspace_floats(int M0, int N0, int K0, int M1, int N1, int K1);
int tr_linear_bwd_params2(const uint16_t* dY0, long ldy0, const uint16_t* X0, long ldx0, float* dW0, float* db0, int M0, int N0, int K0,
                          const uint16_t* dY1, long ldy1, const uint16_t* X1, long ldx1, float* dW1, float* db1, int M1, int N1, int K1,
                          int accumulate, float* ws, size_t ws_floats, tr_stream_t s);
size_t tr_colsum_workspace_floats(int M, int N);
int tr_colsum_bf16(const uint16_t* dY, long ldy, int yskip, float* db, int accumulate, float* ws, size_t ws_floats, int M, int N,
                   tr_stream_t s);
int tr_gelu_bf16(const uint16_t* pre, uint16_t* h, size_t n, tr_stream_t s);
int tr_gelu_bwd_bf16(const uint16_t* pre, uint16_t* dh, size_t n, tr_stream_t s);
size_t tr_layernorm_bwd_workspace_floats(int M, int D);
int tr_layernorm_bwd(const uint16_t* dy, const float* x, long ldx, const float* gamma, const float* g_in, long ldgi, float* g_out,
                     long ldgo, uint16_t* gb_out, const int32_t* idx, int K, int n_in, int n_out, float* g_fused, float* dgamma,
                     float* dbeta, int accumulate, float* ws, size_t ws_floats, int M, int D, float eps, tr_stream_t s);
int tr_layernorm_bwd_scatter_add(const uint16_t* dy, const float* x, const float* gamma, const float* g_in, float* g_out,
                                 const int32_t* idx, int K, int n_out, float* dgamma, float* dbeta, int accumulate, float* ws,
                                 size_t ws_floats, int M, int D, float eps, tr_stream_t s);   /* repeated ids: rows are added (atomics) */
int tr_attention_bwd_bf16(const uint16_t* qkv, const uint16_t* dout, const float* size, const float* dcls, uint16_t* dqkv, int B,
                          int N, int H, tr_stream_t s);
/* The same gradient for ANY sequence length (the training executor uses it beyond 224 tokens: 384 x 384 inputs): keys in blocks of 64,
 * three launches per call -- per-query statistics (log-sum-exp and delta = sum_k p dP by online softmax), dQ per query block, dK / dV
 * per key block -- 9 N x N x 64 products per head instead of 5, no float atomics.  ws: tr_attention_bwd_long_workspace_floats floats. */
size_t tr_attention_bwd_long_workspace_floats(int B, int N, int H);
int tr_attention_bwd_long_bf16(const uint16_t* qkv, const uint16_t* dout, const float* size, const float* dcls, uint16_t* dqkv, float* ws,
                               size_t ws_floats, int B, int N, int H, tr_stream_t s);
int tr_head_bwd(const float* dlogits, const uint16_t* W, const uint16_t* xn, uint16_t* dxn, float* dW, float* db, int accumulate,
                uint16_t* dl16, float* ws, size_t ws_floats, int B, int C, int D, tr_stream_t s);
int tr_embed_bwd(const float* g, float* dpos, float* dcls, int accumulate, int B, int N, int D, tr_stream_t s);
int tr_evit_fuse_bwd(const float* x, const uint16_t* delta, const int32_t* compl_idx, const float* scores, const float* g_fused,
                     float* g_out, uint16_t* gb_out, float* dscore, int B, int N, int K, int D, tr_stream_t s);
int tr_tome_merge_bwd(const float* g_merged, const float* size_in, const float* size_out, const int32_t* unm_idx,
                      const int32_t* src_idx, const int32_t* dst_idx, int32_t* inv_map, float* g_out, uint16_t* gb_out, int B, int N,
                      int r, int D, tr_stream_t s);
int tr_f32_to_bf16(const float* src, uint16_t* dst, size_t n, tr_stream_t s);
/* ---- backward of the soft-assignment reducers (csrc/tr_soft_bwd.hip): SiT sit.py:36-40, PatchMerger patchmerger.py:35-39, Sinkhorn
 * sinkhorn.py:41-86.  Forward: out[b,1+k,:] = sum_p wt[b,1+p,k] * src[b,1+p,:] (tr_softassign_merge*), wt token-major fp32 [B,N,ldl], row 0
 * of an image (CLS) unused.  With g = d out fp32 [B,K+1,D]:
 * tr_soft_dweights: dwt[b,1+p,k] = <g[b,1+k,:], src[b,1+p,:]>  (fp32 [B,N,ldl], CLS rows untouched)
 * tr_soft_dsrc:     dsrc[b,1+p,:] = sum_k wt[b,1+p,k] g[b,1+k,:]  (fp32 [B,N,D], CLS rows untouched)
 * tr_token_softmax_bwd: wt = softmax over the tokens of logits*scale: ds[b,1+p,k] = scale * wt (dwt - sum_p' wt dwt) as bf16 at row stride
 *   ldo (CLS rows zero; columns K..ldo-1 untouched: zero them once), the next GEMMs' operand; dscale (nullable, fp32[1]) (+)= d/d scale
 *   (SiT's learnable temperature, sit.py:34) -- needs the raw logits and tr_token_softmax_bwd_workspace_floats(B,K) floats.
 * tr_sinkhorn_bwd: log_optimal_transport backwards: scores = the raw token.centre products, dplan = d of the transport plan (both fp32
 *   [B,N,ldl]); the iterations are recomputed in LDS (K*(N-1) floats must fit: 224x224 inputs); ds bf16 as above.
 * tr_rownorm_bwd: F.normalize backwards: dx = (d xh - xh <xh, d xh>) / |x| with d xh = da (fp32) + db (bf16, nullable).
 * tr_add_into_bf16: y := bf16(a + y). */
int tr_soft_dweights(const float* g, const float* src, float* dwt, int ldl, int B, int N, int K, int D, tr_stream_t s);
int tr_soft_dsrc(const float* g, const float* wt, int ldl, float* dsrc, int B, int N, int K, int D, tr_stream_t s);
size_t tr_token_softmax_bwd_workspace_floats(int B, int K);
int tr_token_softmax_bwd(const float* wt, const float* dwt, const float* logits, int ldl, float scale, uint16_t* ds, int ldo, float* dscale,
                         int accumulate, float* ws, size_t ws_floats, int B, int N, int K, tr_stream_t s);
int tr_sinkhorn_bwd(const float* scores, const float* dplan, int ldl, float eps, int iters, uint16_t* ds, int ldo, int B, int N, int K,
                    tr_stream_t s);
int tr_rownorm_bwd(const float* x, const float* da, const uint16_t* db, float* dx, int M, int D, tr_stream_t s);
int tr_add_into_bf16(const float* a, uint16_t* y, size_t n, tr_stream_t s);

int tr_rowscale_bf16(const uint16_t* src, uint16_t* dst, const float* scale, int B, int rows, int D, tr_stream_t s);   /* dst[b,r,:] = src[b,r,:] * scale[b] */
int tr_reduce_partials_f32(const float* part, int S, size_t count, float* dst, int accumulate, tr_stream_t s);   /* dst (+)= sum_s part[s] */
/* ---- DyViT training pieces (csrc/tr_dyvit_train.hip, csrc/tr_attention_bwd.hip): see the file headers.  policy / prev / outputs are
 * fp32 [B,N] with entry 0 = the CLS token (always 1); gumbel fp32 [B,N-1,2]. */
int tr_pool_policy(uint16_t* h, const float* policy, int B, int N, int C, float eps, tr_stream_t s);
int tr_pool_policy_bwd(const uint16_t* dcat, const uint16_t* pre0, const uint16_t* cat, const float* policy, uint16_t* dh, float* dpolicy,
                       int B, int N, int C, tr_stream_t s);
int tr_dyvit_decide(const uint16_t* h2, int ldh, const float* w, const float* bias, const float* gumbel, const float* prev,
                    float* policy_out, float* ysoft0, float* sm0, float* hard0, int B, int N, int C, tr_stream_t s);
size_t tr_dyvit_decide_bwd_workspace_floats(int B, int N, int C);
int tr_dyvit_decide_bwd(const float* dkeep, const float* prev, const float* hard0, const float* ysoft0, const float* sm0,
                        const uint16_t* h2, int ldh, const float* w, uint16_t* dh2, float* dprev, float* dw, float* db, int accumulate,
                        float* ws, size_t ws_floats, int B, int N, int C, tr_stream_t s);
int tr_attention_policy_bwd_bf16(const uint16_t* qkv, const uint16_t* dout, const float* policy, uint16_t* dqkv, float* dpol_part, int B,
                                 int N, int H, tr_stream_t s);   /* N <= 224; dpol_part fp32 [B,H,N]: sum the heads with tr_head_sum */
/* The same gradient for any N (the executor uses it beyond 224 tokens: DyViT training at 384 x 384): the key-blocked kernels of
 * tr_attention_bwd_long_bf16 under the keep policy.  ws: tr_attention_bwd_long_workspace_floats(B,N,H) floats. */
int tr_attention_policy_bwd_long_bf16(const uint16_t* qkv, const uint16_t* dout, const float* policy, uint16_t* dqkv, float* dpol_part,
                                      float* ws, size_t ws_floats, int B, int N, int H, tr_stream_t s);
int tr_head_sum(const float* part, float* dst, int B, int H, int N, tr_stream_t s);
int tr_fill_f32(float* p, float v, size_t n, tr_stream_t s);
int tr_add_patch_rows(float* dst, const float* src, int B, int N, tr_stream_t s);
/* DPC-KNN CTM backward (merge_tokens dpcknn.py:103-132 + the score Linear, CTM.forward :155-157): see csrc/tr_backward.hip.
 * ws: at least (B+1)*(D+4) floats; with (8B+1)*(D+4) eight workgroups share an image.  tr_ats_scatter: backward of ATS's row sampling (ats.py:86,157): valid sampled rows t go back to row
 * ids[b,t] of the zero-filled full tensors (g fp32 [B,Ks,D] -> [B,N,D]; d(attn @ v) bf16 likewise). */
int tr_cluster_merge_bwd(const float* g_in, const float* x0, const float* x1, const float* wtok, const int32_t* assign,
                         const float* score_w, float* g_out, uint16_t* gb_out, float* d_sw, float* d_sb, int accumulate, float* ws,
                         size_t ws_floats, int B, int N, int K, int D, tr_stream_t s);
int tr_ats_scatter(const float* g, const uint16_t* dao_s, const int32_t* ids, float* g_full, uint16_t* dao_full, int B, int N, int Ks,
                   int D, tr_stream_t s);

/* ---- whole-model executor: TopKVisionTransformer.forward topk.py:179-212,
 *      EfficientVisionTransformer.forward evit.py:209-244, deit_viz.VisionTransformer.forward :186-212 (eval) ---- */
#define TR_FAMILY_DEIT 0
#define TR_FAMILY_TOPK 1
#define TR_FAMILY_EVIT 2
#define TR_FAMILY_TOME 3   /* keep[blk] = r requested for that block (tome.py:152-155); clamped per call to (N-1)/2 */
#define TR_FAMILY_DYVIT 4  /* eval path of models/dyvit.py: predictor scores -> top-K -> gather BEFORE the block */
#define TR_FAMILY_SIT 5    /* models/sit.py: soft token slimming BEFORE the block */
#define TR_FAMILY_ATS 7    /* models/ats.py: inverse-CDF token sampling inside the attention; keep[blk] = sample_count K (static
                              token bound of the block's output; padded rows are masked keys) */
#define TR_FAMILY_SINKHORN 8 /* models/sinkhorn.py: optimal-transport soft assignment to learned centres BEFORE the block */
#define TR_FAMILY_KMEDOIDS 9 /* models/kmedoids.py: weighted K-Medoids on the patch tokens BEFORE the block, medoids kept */
#define TR_FAMILY_PATCHMERGER 10 /* models/patchmerger.py: K learned queries attend over the normalised tokens BEFORE the block */
#define TR_FAMILY_HEURISTIC 11 /* models/heuristic.py: fixed spatial key/query masks from a block on; no token is removed */
#define TR_FAMILY_DPCKNN 6 /* models/dpcknn.py: DPC-KNN clustering + weighted merge BEFORE the block; keep[blk] = clusters */
#define TR_MAX_DEPTH 32
#define TR_PREC_BF16 0   /* the product path: bf16 MFMA operands, fp32 accumulate / residual / statistics */
#define TR_PREC_FP32 1   /* validation path: the reference's own arithmetic on the GPU (bit-exact indices vs its golden vectors) */
#define TR_PREC_BF16X3 2 /* the fp32 executor with Linears + attention as split-bf16 (hi/lo) products on the matrix cores */

/* Weight MATRICES are bf16 (uint16_t bits) when cfg.precision == TR_PREC_BF16 and fp32 otherwise; vectors are fp32. */
typedef struct {
  const float* ln1_g; const float* ln1_b;
  const void* qkv_w; const float* qkv_b;     /* [3D,D], [3D] */
  const void* proj_w; const float* proj_b;   /* [D,D], [D] */
  const float* ln2_g; const float* ln2_b;
  const void* fc1_w; const float* fc1_b;     /* [Hd,D], [Hd] */
  const void* fc2_w; const float* fc2_b;     /* [D,Hd], [D] */
  const void* mlp_pk;                        /* optional (NULL = none): fc1_w / fc2_w in tr_mlp_pack_bf16's fragment-major order; with it the bf16
                                                EVAL forward may run the block's Mlp as one launch (tr_mlp_fused_bf16: bit-identical to the pair) */
} tr_block_weights;

/* Learned reduction module of one block (families that own one); unused pointers NULL.
 *   DyViT PredictorLG (dyvit.py:96-110): ln = in_conv.0 (eps 1e-5), w0/b0 = in_conv.1 [D,D], w1/b1 = out_conv.0 [D/2,D],
 *     w2/b2 = out_conv.2 [D/4,D/2], w3/b3 = out_conv.4 [2,D/4] (fp32 in both precisions).
 *   SiT TokenSlimmingModule (sit.py:29-34): ln = weight.0 (eps 1e-5), w0/b0 = weight.1 [D/2,D], w1/b1 = weight.3 zero-padded
 *     to n_pad rows ([n_pad, D/2], n_pad = K rounded up to 8), scale = the module's scalar.
 *   PatchMerger (patchmerger.py:32-33): ln = norm (eps 1e-5), w1 = queries zero-padded to n_pad rows [n_pad, D], b1 = zeros,
 *     scale = 1 (embed_dim**-0.5 with scaled_attention).
 *   Sinkhorn (sinkhorn.py:62,73-76): w1 = F.normalize(v) zero-padded to n_pad rows [n_pad, D], b1 = zeros [n_pad].
 *   Heuristic (heuristic.py:247-258): w3 = the block's visibility mask fp32 [N] of 1/0 (CLS 1), n_pad = N; blocks without a
 *     new mask leave w3 NULL and keep the previous one.
 *   ATS (ats.py:48): w3 = sample_steps fp32 [n_pad], n_pad = their count (K-1).
 *   DPC-KNN CTM (dpcknn.py:150-151): w3/b3 = score.weight [1,D] / score.bias [1] (fp32); NULL = args.equal_weight. */
typedef struct {
  const float* ln_g; const float* ln_b;
  const void* w0; const float* b0;
  const void* w1; const float* b1;
  const void* w2; const float* b2;
  const float* w3; const float* b3;
  float scale; int n_pad;
  int h_pad;                   /* DyViT / SiT: width of the D/2 hidden layer as packed (D/2 rounded up to 64 with zero weights, so
                                  the bf16 GEMMs' K %% 64 holds for DeiT-T); 0 = D/2 */
  int reserved_;
} tr_stage_weights;

typedef struct {
  const void* patch_w; const float* patch_b;     /* [D, C*p*p], [D] */
  const float* cls_token; const float* pos_embed;/* [D], [(P+1), D] fp32 */
  const float* norm_g; const float* norm_b;
  const void* head_w; const float* head_b;       /* [classes, D], [classes] */
  tr_block_weights blocks[TR_MAX_DEPTH];
  tr_stage_weights stage[TR_MAX_DEPTH];          /* indexed by BLOCK; read only where keep[blk] > 0 (DyViT, SiT) */
} tr_vit_weights;

typedef struct {
  int family;                 /* TR_FAMILY_* */
  int img_size, patch, in_chans;
  int embed_dim, depth, num_heads, mlp_hidden, num_classes;
  float ln_eps;
  int keep[TR_MAX_DEPTH];     /* per block, 0 = plain block.  Top-K/EViT/DyViT: K patch tokens kept; ToMe: r tokens merged
                                 away; SiT: K output tokens of the slimming module */
  int precision;              /* TR_PREC_* */
  int knn_k;                  /* DPC-KNN: neighbours of the local density (args.k_neighbors, train.py:221 default 5) */
  int cluster_iters;          /* Sinkhorn / K-Medoids iterations (args.cluster_iters, train.py:232 default 3) */
  float sinkhorn_eps;         /* Sinkhorn temperature (args.sinkhorn_eps, train.py:229 default 1.0) */
  int kmed_init[TR_MAX_DEPTH];/* K-Medoids args.equal_weight: per block, 1 + the first medoid id (the host's np.random.choice draw,
                                 kmedoids.py:45); 0 = the attention-weighted branch */
  int ats_dynamic;            /* ATS, eval forward only: 1 = every sampling block shrinks to the batch maximum of unique ids like the reference
                                 (ats.py:77-78) instead of running the static bound keep[blk] with masked rows.  Same valid tokens and logits (to
                                 the attention kernels' summation order); fewer rows downstream.  tr_vit_forward then reads one int back per
                                 sampling block: it SYNCHRONISES the stream there and cannot be captured in a hipGraph.  0 = static (default). */
  int concurrent;             /* eval forward only, a scheduling hint: 1 = the caller runs other forwards BESIDE this one (other streams, other
                                 workspaces: models.py forward_async), so a launch need not fill the chip on its own -- the other forward's
                                 launches take the compute units it leaves idle.  The executor then runs the fused Mlp wherever it is
                                 supported (also where its blocks fill less than 3/4 of a round) and as whole blocks round-robin (no stream-K
                                 hand-over traffic): measured +2 ... +3 % with two forwards in flight, -2 ... -4 % one at a time.  Same bits
                                 either way.  0 = one forward at a time (default). */
} tr_vit_config;

/* Diagnostics (lab, tools/lab/clock_probe.py): the in-kernel clock probes of a library built with -DTR_DIAG_CLOCK -- {shader cycles, 100-MHz
 * ticks, launches} of workgroup 8 of gemm_bf16_pc / mlp_fused_kernel since the last read (read and reset); all zero in a product build. */
int tr_gemm_clock_probe_read(unsigned long long* out3);
int tr_mlp_clock_probe_read(unsigned long long* out3);

/* Bytes of workspace tr_vit_forward needs for batch B (0 on invalid config). */
size_t tr_vit_workspace_bytes(const tr_vit_config* cfg, int B);
/* Status check of the forwards that ran on `workspace` (the same cfg and B): waits for the stream and reports what only the device can
 * know -- today the fused Mlp's stream-K hand-over (tr_mlp_fused_status on the workspace's scratch).  TR_OK, or TR_ERR_LAUNCH when a
 * forward since the last check produced invalid outputs; the record is cleared.  Not part of the forward itself (it synchronises): call
 * it where the logits are consumed. */
int tr_vit_forward_status(const tr_vit_config* cfg, void* workspace, size_t workspace_bytes, int B, tr_stream_t s);

/* img fp32 [B,C,S,S] -> logits fp32 [B,classes].  kept_idx (nullable): device int32 slab of depth*B*(P+1) entries;
 * reduction block blk writes its contiguous [B,K_blk] idx array at offset blk*B*(P+1) (Kept_Tokens, topk.py:196); ToMe
 * writes [unm_idx | src_idx | dst_idx] there ([B,na-r], [B,r], [B,r] back to back).  The slab holds depth*B*(P+1) entries.
 * compl_idx (nullable, EViT): same slab shape, block blk writes [B,P_in-K_blk] at offset blk*B*(P+1)
 * (Fusion_Assign, evit.py:229).  soft_out (nullable, SiT): fp32, the stages' soft assignments [B,K,P_in] back to back in
 * block order (Soft_Assignment_Maps, sit.py:124).  DPC-KNN: kept_idx gets the centres [B,K] (Kept_Tokens), compl_idx the
 * assignment [B,P_in] (Assignment_Maps), K-Medoids likewise (medoid ids, assignment); ATS: kept_idx gets ids [B,K] (CLS id 0 first, 1-based token ids, 0 padding); noise_in (nullable): fp32, the stages' density noise [B,P_in] back to back in block
 * order (dpcknn.py:71-72; NULL = no noise).  features_out (nullable): fp32, the residual stream after EVERY block
 * ([B,N_blk,D] back to back in block order; viz_data["Features"]).  tokens_out (nullable, HOST pointer, int[depth]): token count after each block. */
int tr_vit_forward(const tr_vit_config* cfg, const tr_vit_weights* w, const float* img, float* logits,
                   void* workspace, size_t workspace_bytes, int32_t* kept_idx, int32_t* compl_idx, float* soft_out,
                   const float* noise_in, float* features_out, int* tokens_out, int B, tr_stream_t s);

/* ---- training: forward that keeps its activations + backward executor (csrc/tr_vit.hip, csrc/tr_train.hip) ----------------------
 * engine.py:50-76: `output = model(samples)` in train mode, `loss.backward()`.  Every family (bf16 operands, fp32 master weights and
 * gradients), up to 640 tokens (224 x 224 and 384 x 384 inputs).  DyViT (dyvit.py:221-229): noise_in = the Gumbel noise of every stage, fp32 [B,P,2] back to back (torch's
 * -log(Exp(1)) draws); the stages' policies stay on the tape (tr_vit_tape_layout); features_out (nullable) fp32 [B,N0,D]: the final
 * norm of every row (the distillation features, dyvit.py:252-258); tr_vit_backward takes dpred fp32 [stages,B,P] (gradient wrt each
 * stage's out_pred_prob) and dfeat fp32 [B,N0,D] (gradient wrt features_out), both nullable.
 * tr_vit_forward_train: as tr_vit_forward, and every activation the backward needs is written to `tape` (tr_vit_tape_bytes(cfg, B)
 *   bytes, caller-owned; 0 = family / precision without a training path).  drop_scale (nullable, both calls get the same): DropPath
 *   (timm 0.4.12 drop_path, topk.py:78,87,95) as fp32 [2*depth, B]: entry [2i][b] / [2i+1][b] = the scale (0 or 1/keep_prob) of
 *   image b's attention / MLP branch in block i -- the random draw is the caller's.  Dropout: see below.
 * tr_vit_backward: dlogits fp32 [B,classes] -> parameter gradients.  `w` = the forward's weights; `wt` = same struct with the block
 *   matrices TRANSPOSED (bf16: qkv_w [D,3D], proj_w [D,D], fc1_w [D,Hd], fc2_w [Hd,D]); `grads` = same struct, every pointer an
 *   fp32 buffer of the parameter's shape (written; added to when accumulate != 0).  workspace: tr_vit_backward_workspace_bytes.
 *   [blk_hi .. blk_lo]: the blocks this call walks, in reverse; blk_hi == depth-1 runs the classifier + final norm first, blk_lo == 0
 *   the embedding gradients last; a whole pass is (depth-1, 0) or consecutive ranges in descending order (the stream's gradient
 *   stays in the workspace in between) -- the hook for overlapping the data-parallel gradient reduction (train.py:405-407) with
 *   the backward: reduce one range's gradients on a second stream while the next range runs. */
size_t tr_vit_tape_bytes(const tr_vit_config* cfg, int B);
int tr_vit_forward_train(const tr_vit_config* cfg, const tr_vit_weights* w, const float* img, float* logits, void* workspace,
                         size_t workspace_bytes, void* tape, size_t tape_bytes, const float* noise_in, float* features_out,
                         const float* drop_scale, int* tokens_out, int B, tr_stream_t s, const uint8_t* dropout_keep, float drop_rate);
/* Dropout (timm's drop_rate, train.py:46 --drop: pos_drop topk.py:186, proj_drop :53, the Mlp's two nn.Dropout): dropout_keep (nullable
 * = drop_rate 0; both calls get the same) is the caller's keep mask, 1 byte per element (non-zero = keep), tr_vit_dropout_mask_bytes()
 * bytes in the order the forward consumes them -- the embedded tokens [B,N0,D], then per block proj's output rows, the Mlp's hidden
 * layer and its output: the order in which the reference module draws them.  Survivors are scaled by 1 / (1 - drop_rate).
 * (attn_drop_rate -- dropout on the attention probabilities, topk.py:49 -- is not built; the reference's CLI cannot set it either.) */
size_t tr_vit_dropout_mask_bytes(const tr_vit_config* cfg, int B);
/* fp32 master parameters -> the bf16 operand copies of the GEMMs, every matrix of a model in ONE launch (an optimizer step changes them
 * all: engine.py:76-91).  Item i: src fp32 [rows, cols] contiguous -> dst bf16 [rows, cols] (nullable) and dst_t bf16 [cols, rows]
 * (nullable: the transposed copy the data-gradient GEMMs read).  items / first are DEVICE arrays; first[i] = tiles before item i with
 * ceil(rows/64) * ceil(cols/64) tiles per item, total_tiles = first[n_items]. */
typedef struct tr_cast_item { const void* src; void* dst; void* dst_t; int rows, cols; } tr_cast_item;
int tr_cast_pack_bf16(const tr_cast_item* items, const int* first, int n_items, int total_tiles, tr_stream_t s);
/* One launch for the optimizer step of the fine-tune path (csrc/tr_optim.hip): AdamW on every parameter (torch's fused AdamW
 * arithmetic, bit for bit), the gradient zeroed behind it when `zero_grads` != 0, and the bf16 (+ transposed) operand copies rewritten
 * in the same pass.
 * Item i: p, g, m (exp_avg), v (exp_avg_sq) fp32 [rows, cols] contiguous; dst bf16 [rows, cols] / dst_t bf16 [cols, rows] nullable;
 * group selects lr8[group] / wd8[group] (up to 8 parameter groups per launch).  items / first: DEVICE arrays, first[i] = tiles before
 * item i (ceil(rows/64) * ceil(cols/64) per item), total_tiles = first[n_items].  bias_correction1 = 1 - beta1^step,
 * bias_correction2_sqrt = sqrt(1 - beta2^step), computed by the caller in double and rounded to float. */
typedef struct tr_adamw_item { void* p; void* g; void* m; void* v; void* dst; void* dst_t; int rows, cols, group, pad_; } tr_adamw_item;
int tr_adamw_step(const tr_adamw_item* items, const int* first, int n_items, int total_tiles, double beta1, double beta2, double eps,
                  float bias_correction1, float bias_correction2_sqrt, const double* lr8, const double* wd8, int zero_grads,
                  tr_stream_t s);
int tr_dropout_bf16(const uint16_t* src, uint16_t* dst, const uint8_t* keep, float mul, size_t n, tr_stream_t s);
int tr_dropout_f32(const float* src, float* dst, const uint8_t* keep, float mul, size_t n, tr_stream_t s);
/* Byte offsets of block blk's tape slots (x0,x1,xn1,qkv,ao,dattn,x2,xn2,pre,h,idx,idx2,scores,size) followed by its token counts
 * (entering, in attention, in the MLP) and its reduction count: lets a host read the decisions of a training forward. */
int tr_vit_tape_layout(const tr_vit_config* cfg, int B, int blk, size_t* out18);
size_t tr_vit_backward_workspace_bytes(const tr_vit_config* cfg, int B);
int tr_vit_backward(const tr_vit_config* cfg, const tr_vit_weights* w, const tr_vit_weights* wt, const tr_vit_weights* grads,
                    const float* dlogits, const float* dpred, const float* dfeat, const float* drop_scale, const void* tape,
                    size_t tape_bytes, void* workspace, size_t workspace_bytes, int accumulate, int blk_hi, int blk_lo, int B,
                    tr_stream_t s, const uint8_t* dropout_keep, float drop_rate);

#ifdef __cplusplus
}
#endif
#endif /* TOKENREDUCTION_HIP_H */
