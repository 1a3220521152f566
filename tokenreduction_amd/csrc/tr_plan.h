// Static shape bookkeeping shared by the forward executor (tr_vit.hip) and the backward executor (tr_train.hip).
// Token counts are compile-time per (model, keep_rate, reduction_loc) -- topk.py:56 int(ratio*196) -- so the size and the place of
// every saved activation (the "tape" of a training forward) is known before anything is launched.
#pragma once
#include "tr_common.h"

namespace trplan {

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// families whose training path (saved activations + backward) is built
inline bool trainable_family(int family) {
  return family == TR_FAMILY_DEIT || family == TR_FAMILY_TOPK || family == TR_FAMILY_EVIT || family == TR_FAMILY_TOME ||
         family == TR_FAMILY_DPCKNN || family == TR_FAMILY_ATS || family == TR_FAMILY_DYVIT || family == TR_FAMILY_KMEDOIDS ||
         family == TR_FAMILY_HEURISTIC || family == TR_FAMILY_SIT || family == TR_FAMILY_PATCHMERGER || family == TR_FAMILY_SINKHORN;
}
inline bool soft_family(int family) { return family == TR_FAMILY_SIT || family == TR_FAMILY_PATCHMERGER || family == TR_FAMILY_SINKHORN; }
inline int soft_ld(int K) { return (K + 7) / 8 * 8; }          // row stride of the token-major weight matrices (= tr_stage_weights.n_pad)
inline int soft_ld64(int K) { return (K + 63) / 64 * 64; }     // row stride of their bf16 gradient (a GEMM contraction dimension)

// tokens (incl. CLS) entering block i, inside its attention, and inside its MLP: the rules of tr_vit_forward
struct TokenPlan {
  int n_pre[TR_MAX_DEPTH], n_att[TR_MAX_DEPTH], n_mlp[TR_MAX_DEPTH];
  int kk[TR_MAX_DEPTH];     // Top-K / EViT: K kept (0 = plain block); ToMe: r merged (after the 50 % cap); ATS: sample bound; pre-block: clusters
  int N0, P;
};

// rows of block i that go through attn.proj: the attended tokens, except ATS after a sampling stage, where only the sampled rows do (ats.py:86,129)
inline size_t proj_rows(const tr_vit_config* c, const TokenPlan& t, int i) {
  return (size_t)((c->family == TR_FAMILY_ATS && t.kk[i] > 0) ? t.n_mlp[i] : t.n_att[i]);
}

inline bool make_token_plan(const tr_vit_config* c, TokenPlan* t) {
  const int g = c->img_size / c->patch;
  t->P = g * g;
  t->N0 = t->P + 1;
  int N = t->N0;
  for (int i = 0; i < c->depth; ++i) {
    t->n_pre[i] = N;
    t->kk[i] = 0;
    const int f = c->family;
    const bool pre = f == TR_FAMILY_DPCKNN || f == TR_FAMILY_KMEDOIDS || f == TR_FAMILY_PATCHMERGER || f == TR_FAMILY_SINKHORN ||
                     f == TR_FAMILY_SIT;
    if (f == TR_FAMILY_DYVIT && c->keep[i] > 0) {
      // DyViT TRAINS without removing tokens (dyvit.py:223-229: policy masks instead); kk marks the predictor stages
      if (c->keep[i] > N - 1) return false;
      t->kk[i] = c->keep[i];
    }
    if (pre && c->keep[i] > 0) {
      if (c->keep[i] > N - 1) return false;
      t->kk[i] = c->keep[i];
      N = c->keep[i] + 1;
    }
    t->n_att[i] = N;
    if (f == TR_FAMILY_TOPK || f == TR_FAMILY_EVIT) {
      int K = c->keep[i];
      if (K < 0 || K > N - 1) return false;
      if (K == N - 1) K = 0;                         // topk.py:57: left_tokens == N-1 -> plain block
      t->kk[i] = K;
      if (K > 0) N = K + 1 + (f == TR_FAMILY_EVIT ? 1 : 0);
    } else if (f == TR_FAMILY_TOME) {
      if (c->keep[i] < 0) return false;
      const int r = c->keep[i] < (N - 1) / 2 ? c->keep[i] : (N - 1) / 2;   // tome.py:253
      t->kk[i] = r;
      N -= r;
    } else if (f == TR_FAMILY_ATS) {
      const int Ks = c->keep[i];
      if (Ks > 0) {
        if (Ks < 2 || Ks > N) return false;
        t->kk[i] = Ks;
        N = Ks;
      }
    }
    t->n_mlp[i] = N;
  }
  return true;
}

// ---- the tape: activations a training forward keeps for the backward pass, one slot set per block
struct BlockTape {
  size_t x0;      // fp32 [B, n_pre, D]   stream entering a pre-block reducer, pending residual added (DPC-KNN);
                  //                      ATS sampling blocks: the sampled rows of the stream [B, n_mlp, D] (x before the attention residual)
  size_t x1;      // fp32 [B, n_att, D]   input of norm1
  size_t xn1;     // bf16 [B, n_att, D]   norm1 output (qkv's operand)
  size_t qkv;     // bf16 [B, n_att, 3D]
  size_t ao;      // bf16 [B, n_att, D]   attention output = proj's operand (ATS sampling blocks: only the sampled rows [B, n_mlp, D])
  size_t dattn;   // bf16 [B, n_att, D]   proj output (EViT reduction blocks: the fused token reads x + this at the dropped rows)
  size_t x2;      // fp32 [B, n_mlp, D]   input of norm2
  size_t xn2;     // bf16 [B, n_mlp, D]
  size_t pre;     // bf16 [B, n_mlp, Hd]  fc1 pre-activation
  size_t h;       // bf16 [B, n_mlp, Hd]  gelu(pre)
  size_t idx;     // int32 [B, N0]        kept ids / ToMe [unm|src|dst] / cluster centres / ATS ids
  size_t idx2;    // int32 [B, N0]        complement ids / cluster assignment
  size_t scores;  // fp32 [B, N0]         Top-K scores (EViT fuse weights) / DPC-KNN token weights
  size_t size;    // fp32 [B, N0]         ToMe token sizes AFTER this block's merge / ATS key mask after this block
  size_t xa;      // reserved
  // DyViT predictor stage (blocks with kk > 0, family DYVIT): the PredictorLG activations of dyvit.py:113-119 over all B*N rows
  size_t pu;      // bf16 [B*N, D]    in_conv.0 (LayerNorm) output
  size_t ppre0;   // bf16 [B*N, D]    in_conv.1 pre-activation
  size_t pcat;    // bf16 [B*N, D]    [local | policy-weighted global mean]: out_conv.0's operand
  size_t ppre1;   // bf16 [B*N, Hh]   out_conv.0 pre-activation      ph1: its GELU
  size_t ph1;
  size_t ppre2;   // bf16 [B*N, Q]    out_conv.2 pre-activation (Q = D/4 padded to 64)      ph2: its GELU
  size_t ph2;
  size_t pol;     // fp32 [B, N]      the stage's policy = [1, hard_keep]  (entry 0 = CLS)
  size_t ysoft;   // fp32 [B, N]      softmax(score + gumbel)[..., 0];  sm: softmax(z)[..., 0];  hard: the one-hot's first entry
  size_t sm;
  size_t hard;
  // soft-assignment stage (blocks with kk > 0, families SIT / PATCHMERGER / SINKHORN), rows = B * n_pre; x0 holds the stream
  // entering the stage, pu the GEMM operand (LayerNorm / unit-norm rows, bf16), ppre0 / pcat SiT's hidden layer [rows, D/2]
  size_t sxh;     // fp32 [rows, D]         the rows that are summed when they are not x0: LayerNorm (PatchMerger), unit norm (Sinkhorn)
  size_t slog;    // fp32 [rows, soft_ld]   raw logits / scores
  size_t swt;     // fp32 [rows, soft_ld]   token-axis softmax / transport plan
};

struct TapePlan {
  BlockTape blk[TR_MAX_DEPTH];
  size_t cols;     // bf16 [B*P, C*p*p]  im2col of the images (PatchEmbed's weight-gradient operand)
  size_t xfinal;   // fp32 [B, D]        CLS rows entering the final norm
  size_t xcls;     // bf16 [B, D]        final norm output (the classifier's operand)
  size_t ones;     // fp32 [B, N0]       DyViT: the all-ones policy of the blocks before the first predictor stage
  size_t xfin_all; // fp32 [B, N0, D]    DyViT: the whole stream entering the final norm (the distillation features need every row)
  size_t total;
};

inline bool make_tape_plan(const tr_vit_config* c, int B, const TokenPlan& t, TapePlan* p) {
  const size_t D = c->embed_dim, Hd = c->mlp_hidden;
  const size_t kcols = (size_t)c->in_chans * c->patch * c->patch;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes); return at; };
  p->cols = take((size_t)B * t.P * kcols * 2);
  p->xfinal = take((size_t)B * D * 4);
  p->xcls = take((size_t)B * D * 2);
  const bool soft = soft_family(c->family);
  const bool pre = c->family == TR_FAMILY_DPCKNN || c->family == TR_FAMILY_KMEDOIDS || soft;
  const bool dyvit = c->family == TR_FAMILY_DYVIT;
  p->ones = dyvit ? take((size_t)B * t.N0 * 4) : 0;
  p->xfin_all = dyvit ? take((size_t)B * t.N0 * D * 4) : 0;
  const size_t Hh = (D / 2 + 63) / 64 * 64, Q = (D / 4 + 63) / 64 * 64;
  for (int i = 0; i < c->depth; ++i) {
    BlockTape& b = p->blk[i];
    const size_t Tp = (size_t)B * t.n_pre[i], Ta = (size_t)B * t.n_att[i], Tm = (size_t)B * t.n_mlp[i];
    b.x0 = (pre && t.kk[i] > 0) ? take(Tp * D * 4) : ((c->family == TR_FAMILY_ATS && t.kk[i] > 0) ? take(Tm * D * 4) : 0);
    b.x1 = take(Ta * D * 4);
    if (dyvit && t.kk[i] > 0) b.x0 = b.x1;       // the predictor reads the stream norm1 reads: one slot (norm1 runs in place on it)
    b.xn1 = take(Ta * D * 2);
    b.qkv = take(Ta * 3 * D * 2);
    b.ao = take(Ta * D * 2);
    b.dattn = take(Ta * D * 2);
    b.x2 = take(Tm * D * 4);
    b.xn2 = take(Tm * D * 2);
    b.pre = take(Tm * Hd * 2);
    b.h = take(Tm * Hd * 2);
    b.idx = take((size_t)B * t.N0 * 4);
    b.idx2 = take((size_t)B * t.N0 * 4);
    b.scores = take((size_t)B * t.N0 * 4);
    b.size = take((size_t)B * t.N0 * 4);
    b.xa = 0;
    b.pu = b.ppre0 = b.pcat = b.ppre1 = b.ph1 = b.ppre2 = b.ph2 = b.pol = b.ysoft = b.sm = b.hard = 0;
    if (dyvit && t.kk[i] > 0) {
      b.pu = take(Ta * D * 2); b.ppre0 = take(Ta * D * 2); b.pcat = take(Ta * D * 2);
      b.ppre1 = take(Ta * Hh * 2); b.ph1 = take(Ta * Hh * 2);
      b.ppre2 = take(Ta * Q * 2); b.ph2 = take(Ta * Q * 2);
      b.pol = take(Ta * 4); b.ysoft = take(Ta * 4); b.sm = take(Ta * 4); b.hard = take(Ta * 4);
    }
    b.sxh = b.slog = b.swt = 0;
    if (soft && t.kk[i] > 0) {
      const size_t ld = soft_ld(t.kk[i]);
      b.pu = take(Tp * D * 2);
      if (c->family == TR_FAMILY_SIT) { b.ppre0 = take(Tp * Hh * 2); b.pcat = take(Tp * Hh * 2); }     // hidden layer as packed (D/2 padded to 64)
      else b.sxh = take(Tp * D * 4);
      b.slog = take(Tp * ld * 4);
      b.swt = take(Tp * ld * 4);
    }
  }
  p->total = o;
  return true;
}

}  // namespace trplan
