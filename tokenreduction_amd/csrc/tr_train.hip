// Backward executor: enqueues the gradient pass of a training forward (tr_vit_forward_train) on the caller's stream.
//
// The reference's training step is `output = model(samples); loss = criterion(...); loss.backward()` (engine.py:50-76) with
// torch.autograd deriving the backward of the eager ops; under DistributedDataParallel (train.py:405-407) the parameter
// gradients are all-reduced in buckets while the backward is still running.  Here the backward is one fixed launch sequence over
// the tape the forward left (tr_plan.h): block by block in reverse, for each nn.Linear a weight-gradient GEMM that also sums the bias
// gradient (tr_linear_bwd_params) and a data-gradient GEMM (tr_gemm_bf16 on the transposed weight), the attention / LayerNorm / GELU backward
// kernels, and the backward of the block's token reduction (Top-K scatter topk.py:89-93, EViT fused token evit.py:111-123, ToMe
// merge tome.py:309-323).  A call may cover a RANGE of blocks [blk_hi .. blk_lo]: the caller walks the model in a few ranges and,
// after each, starts that range's gradient bucket on RCCL from a second stream while the next range runs (the DDP overlap).
// Host-side only: no allocation, no synchronisation -- capturable in a hipGraph.
#include "tr_common.h"
#include "tr_plan.h"

using trplan::align_up;

// tr_backward.hip: deferred reduction of the LayerNorm parameter gradients (one launch per backward call instead of one per norm)
void tr_ln_defer_begin(float* region, size_t floats, hipStream_t st);
int tr_ln_defer_flush();
void tr_ln_defer_end();

namespace {

struct BwdPlan {
  size_t g0, g1, gb0, gb1, gb2, dxn, dqkv, dao, dh, zeros, wsf, dscore, gfused, invmap, dxcls, dl16, dpol, dprev, dpolpart, soft_dp, soft_ds, soft_s, attn_stats, lnpart, total;
  size_t wsf_floats, lnpart_floats;
};

bool make_bwd_plan(const tr_vit_config* c, int B, const trplan::TokenPlan& t, BwdPlan* p) {
  const size_t D = c->embed_dim, Hd = c->mlp_hidden, T = (size_t)B * t.N0;
  const size_t kcols = (size_t)c->in_chans * c->patch * c->patch;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes); return at; };
  p->g0 = take(T * D * 4);
  p->g1 = take(T * D * 4);
  p->gb0 = take(T * D * 2);
  p->gb1 = take(T * D * 2);
  p->gb2 = take(T * D * 2);          // spare: keeps fc2's dY alive past norm2's backward (the block's four weight gradients run as one launch)
  p->dxn = take(T * D * 2);
  p->dqkv = take(T * 3 * D * 2);
  p->dao = take(T * D * 2);
  p->dh = take(T * Hd * 2);
  size_t zmax = 3 * D > Hd ? 3 * D : Hd;
  if (kcols > zmax) zmax = kcols;
  p->zeros = take(zmax * 4);
  // scratch of the two-stage reductions: the largest request of any call below
  size_t f = tr_wgrad_workspace_floats((int)T, (int)(3 * D), (int)D);
  auto upd = [&](size_t v) { if (v > f) f = v; };
  upd(tr_wgrad_workspace_floats((int)T, (int)Hd, (int)D));
  upd(tr_wgrad_workspace_floats((int)T, (int)D, (int)Hd));
  upd(tr_wgrad_workspace_floats((int)T, (int)D, (int)D));
  upd(tr_wgrad_workspace_floats(B * t.P, (int)D, (int)kcols));
  upd(tr_colsum_workspace_floats((int)T, (int)(3 * D)));
  upd(tr_colsum_workspace_floats((int)T, (int)Hd));
  upd(tr_layernorm_bwd_workspace_floats((int)T, (int)D));
  upd(tr_wgrad_workspace_floats(B, c->num_classes, (int)D));
  upd((size_t)(8 * B + 1) * (D + 4));          // tr_cluster_merge_bwd: eight workgroups per image
  upd(tr_dyvit_decide_bwd_workspace_floats(B, t.N0, (int)(D / 4)));
  upd(tr_wgrad_workspace_floats((int)T, (int)(D / 2), (int)D));
  int soft_k = 0;                    // soft-assignment families: the widest stage
  if (trplan::soft_family(c->family))
    for (int i = 0; i < c->depth; ++i) soft_k = t.kk[i] > soft_k ? t.kk[i] : soft_k;
  if (soft_k > 0) {
    upd(tr_wgrad_workspace_floats((int)T, trplan::soft_ld(soft_k), (int)D));
    upd(tr_token_softmax_bwd_workspace_floats(B, soft_k));
  }
  for (int i = 0; i < c->depth; ++i) {          // the paired weight-gradient launches of every block, at the block's own token counts
    const int M1 = B * t.n_att[i], M2 = B * t.n_mlp[i];
    const int Mp = (c->family == TR_FAMILY_ATS && t.kk[i] > 0) ? M2 : M1;
    upd(tr_linear_bwd_params2_workspace_floats(M2, (int)D, (int)Hd, M2, (int)Hd, (int)D));
    upd(tr_linear_bwd_params2_workspace_floats(Mp, (int)D, (int)D, M1, (int)(3 * D), (int)D));
    const tr_linear_grad four[4] = {{nullptr, (long)D, nullptr, (long)Hd, nullptr, nullptr, M2, (int)D, (int)Hd},
                                    {nullptr, (long)Hd, nullptr, (long)D, nullptr, nullptr, M2, (int)Hd, (int)D},
                                    {nullptr, (long)D, nullptr, (long)D, nullptr, nullptr, Mp, (int)D, (int)D},
                                    {nullptr, (long)(3 * D), nullptr, (long)D, nullptr, nullptr, M1, (int)(3 * D), (int)D}};
    upd(tr_linear_bwd_group_workspace_floats(four, 4));
  }
  p->wsf_floats = f;
  p->wsf = take(f * 4);
  p->dscore = take(T * 4);
  p->gfused = take((size_t)B * D * 4);
  p->invmap = take(T * 4);
  p->dxcls = take((size_t)B * D * 2);
  p->dl16 = take((size_t)B * c->num_classes * 2);
  p->dpol = take(T * 4);
  p->dprev = take(T * 4);
  p->dpolpart = take(T * c->num_heads * 4);
  p->attn_stats = t.N0 > 224 ? take(tr_attention_bwd_long_workspace_floats(B, t.N0, c->num_heads) * 4) : 0;
  p->soft_dp = p->soft_ds = p->soft_s = 0;
  if (soft_k > 0) {
    p->soft_dp = take(T * trplan::soft_ld(soft_k) * 4);
    p->soft_ds = take(T * trplan::soft_ld64(soft_k) * 2);
    p->soft_s = take(T * D * 4);
  }
  // the LayerNorm parameter-gradient partials of a pass (reduced by one launch at its end; a flush after 16 norms reuses the region from its
  // start): two norms per block, one per reduction stage that owns a norm, the final norm -- never more than 16 slices in flight
  int ln_norms = 2 * c->depth + 2;
  for (int i = 0; i < c->depth; ++i) ln_norms += c->keep[i] > 0 ? 1 : 0;
  p->lnpart_floats = (size_t)(ln_norms < 16 ? ln_norms : 16) * tr_layernorm_bwd_workspace_floats((int)T, (int)D);
  p->lnpart = take(p->lnpart_floats * 4);
  p->total = o;
  return true;
}

#define TR_TRY(call)                \
  do {                              \
    int rc__ = (call);              \
    if (rc__ != TR_OK) return rc__; \
  } while (0)

inline float* F(const void* p) { return const_cast<float*>(static_cast<const float*>(p)); }
inline const uint16_t* U(const void* p) { return static_cast<const uint16_t*>(p); }

}  // namespace

extern "C" size_t tr_vit_backward_workspace_bytes(const tr_vit_config* cfg, int B) {
  trplan::TokenPlan t;
  BwdPlan p;
  if (!cfg || B <= 0 || !trplan::trainable_family(cfg->family) || !trplan::make_token_plan(cfg, &t) || !make_bwd_plan(cfg, B, t, &p)) return 0;
  return p.total;
}

// wt: the weight MATRICES transposed (bf16): blocks[i].qkv_w = qkv.weight^T [D,3D], proj_w = proj.weight^T [D,D], fc1_w = fc1.weight^T
//     [D,Hd], fc2_w = fc2.weight^T [Hd,D]; other fields unused.  w: the forward's weights (head_w and the LayerNorm gammas are read).
// grads: same layout as tr_vit_weights, every pointer an fp32 gradient buffer of the parameter's shape (matrices included).
// accumulate != 0: gradients are added to the buffers (engine.py:41 grad accumulation), else overwritten.
// [blk_hi .. blk_lo] (blk_hi >= blk_lo): the blocks this call walks, in reverse.  blk_hi == depth-1 also runs the classifier and
// the final norm first; blk_lo == 0 also runs the embedding gradients last.
// DyViT only: dpred (nullable) fp32 [stages, B, P]: gradient wrt each stage's out_pred_prob (the ratio loss, losses.py:113-118), stages in
// block order; dfeat (nullable) fp32 [B, N0, D]: gradient wrt the final-norm token features (distillation, losses.py:134-156; row 0 = 0).  The gradient of the residual stream stays in the
// workspace between calls, so a backward pass is the calls (depth-1 .. a), (a-1 .. b), ..., (c .. 0) in this order.
extern "C" int tr_vit_backward(const tr_vit_config* cfg, const tr_vit_weights* w, const tr_vit_weights* wt, const tr_vit_weights* grads,
                               const float* dlogits, const float* dpred, const float* dfeat, const float* drop_scale, const void* tape_,
                               size_t tape_bytes, void* workspace, size_t workspace_bytes, int accumulate, int blk_hi, int blk_lo, int B,
                               tr_stream_t s, const uint8_t* dropout_keep, float drop_rate) {
  TR_REQUIRE(cfg && w && wt && grads && dlogits && tape_ && workspace, TR_ERR_NULL, "tr_vit_backward: null pointer");
  TR_REQUIRE((dropout_keep == nullptr) == (drop_rate == 0.f) && drop_rate >= 0.f && drop_rate < 1.f, TR_ERR_CONFIG,
             "tr_vit_backward: dropout needs the forward's keep mask AND its drop_rate (got mask %p, rate %g)", (const void*)dropout_keep, (double)drop_rate);
  const float drop_mul = dropout_keep != nullptr ? 1.0f / (1.0f - drop_rate) : 1.0f;
  TR_REQUIRE(cfg->precision == TR_PREC_BF16 && trplan::trainable_family(cfg->family), TR_ERR_CONFIG,
             "tr_vit_backward: family %d / precision %d has no training path", cfg->family, cfg->precision);
  trplan::TokenPlan t;
  trplan::TapePlan tp;
  BwdPlan bp;
  TR_REQUIRE(trplan::make_token_plan(cfg, &t) && trplan::make_tape_plan(cfg, B, t, &tp) && make_bwd_plan(cfg, B, t, &bp), TR_ERR_CONFIG,
             "tr_vit_backward: invalid config");
  TR_REQUIRE(tape_bytes >= tp.total && workspace_bytes >= bp.total, TR_ERR_SHAPE, "tr_vit_backward: tape (%zu < %zu) or workspace (%zu < %zu) too small",
             tape_bytes, tp.total, workspace_bytes, bp.total);
  TR_REQUIRE(tr_aligned16(tape_) && tr_aligned16(workspace), TR_ERR_ALIGN, "tr_vit_backward: tape / workspace must be 16-byte aligned");
  TR_REQUIRE(blk_lo >= 0 && blk_hi >= blk_lo && blk_hi < cfg->depth, TR_ERR_CONFIG, "tr_vit_backward: bad block range [%d .. %d]", blk_hi, blk_lo);
  const char* tape = static_cast<const char*>(tape_);
  char* ws = static_cast<char*>(workspace);
  hipStream_t st = static_cast<hipStream_t>(s);
  const int D = cfg->embed_dim, H = cfg->num_heads, Hd = cfg->mlp_hidden, C = cfg->num_classes;
  const int kcols = cfg->in_chans * cfg->patch * cfg->patch;
  float* g = reinterpret_cast<float*>(ws + bp.g0);
  float* g_alt = reinterpret_cast<float*>(ws + bp.g1);
  uint16_t* gb = reinterpret_cast<uint16_t*>(ws + bp.gb0);
  uint16_t* gb_alt = reinterpret_cast<uint16_t*>(ws + bp.gb1);
  uint16_t* dxn = reinterpret_cast<uint16_t*>(ws + bp.dxn);
  uint16_t* dqkv = reinterpret_cast<uint16_t*>(ws + bp.dqkv);
  uint16_t* dao = reinterpret_cast<uint16_t*>(ws + bp.dao);
  uint16_t* dh = reinterpret_cast<uint16_t*>(ws + bp.dh);
  float* zeros = reinterpret_cast<float*>(ws + bp.zeros);
  float* wsf = reinterpret_cast<float*>(ws + bp.wsf);
  float* dscore = reinterpret_cast<float*>(ws + bp.dscore);
  float* gfused = reinterpret_cast<float*>(ws + bp.gfused);
  int32_t* invmap = reinterpret_cast<int32_t*>(ws + bp.invmap);
  uint16_t* dxcls = reinterpret_cast<uint16_t*>(ws + bp.dxcls);
  const size_t wsn = bp.wsf_floats;
  const int acc = accumulate ? 1 : 0;

  uint16_t* gb_spare = reinterpret_cast<uint16_t*>(ws + bp.gb2);
  // The four parameter-gradient products of a block (fc2, fc1, proj, qkv) run as ONE launch after the attention backward, when all four dY
  // exist -- provided nothing has rewritten fc2's dY (the bf16 stream gradient gb) by then: norm2's backward writes its bf16 output into the
  // spare buffer instead and the two trade places.  With DropPath the scaled dY copies live in scratch that is reused: pairs then.
  const bool four = drop_scale == nullptr && dropout_keep == nullptr;    // the branches' dY are the stream gradient itself (no scaled / masked copies)
  auto plain_ln2 = [&](int j) {      // blocks whose norm2 backward writes gb in place (no gather / merge between norm2 and the stream)
    const bool gathers = (cfg->family == TR_FAMILY_TOPK || cfg->family == TR_FAMILY_EVIT || cfg->family == TR_FAMILY_TOME) && t.kk[j] > 0;
    return !gathers;
  };
  // which stream-gradient buffers are current at block blk_hi: replay the pointer moves of the blocks above (per block: the spare rotation
  // of norm2's backward first, then the swap of a token-reducing block)
  for (int j = cfg->depth - 1; j > blk_hi; --j) {
    if (four && plain_ln2(j)) { uint16_t* tb = gb; gb = gb_spare; gb_spare = tb; }
    if (t.kk[j] > 0 && cfg->family != TR_FAMILY_DYVIT) {          // every token-reducing block of the built families swaps once
      float* tg = g; g = g_alt; g_alt = tg;
      uint16_t* tb = gb; gb = gb_alt; gb_alt = tb;
    }
  }
  if (blk_hi == cfg->depth - 1) {
    size_t zmax = (size_t)(3 * D > Hd ? 3 * D : Hd);
    if ((size_t)kcols > zmax) zmax = kcols;
    TR_REQUIRE(hipMemsetAsync(zeros, 0, zmax * 4, st) == hipSuccess, TR_ERR_LAUNCH, "tr_vit_backward: memset failed");
    // ---- classifier + final norm (topk.py:201-203): gradient enters the CLS rows of the last block's output stream
    const int Nl = t.n_mlp[cfg->depth - 1];
    TR_TRY(tr_head_bwd(dlogits, U(w->head_w), U(tape + tp.xcls), dxcls, F(grads->head_w), F(grads->head_b), acc,
                       reinterpret_cast<uint16_t*>(ws + bp.dl16), wsf, wsn, B, C, D, s));
    TR_REQUIRE(hipMemsetAsync(g, 0, (size_t)B * Nl * D * 4, st) == hipSuccess, TR_ERR_LAUNCH, "tr_vit_backward: memset failed");
    TR_TRY(tr_layernorm_bwd(dxcls, reinterpret_cast<const float*>(tape + tp.xfinal), D, w->norm_g, nullptr, 0, g, (long)Nl * D, nullptr, nullptr, 0, 0,
                            0, nullptr, F(grads->norm_g), F(grads->norm_b), acc, wsf, wsn, B, D, cfg->ln_eps, s));
    if (dfeat != nullptr) {
      // distillation: gradient wrt the final norm of every row (dyvit.py:252): a second pass of the norm's backward over all rows
      TR_REQUIRE(cfg->family == TR_FAMILY_DYVIT, TR_ERR_CONFIG, "tr_vit_backward: dfeat is DyViT's distillation gradient");
      TR_TRY(tr_f32_to_bf16(dfeat, dxn, (size_t)B * Nl * D, s));
      TR_TRY(tr_layernorm_bwd(dxn, reinterpret_cast<const float*>(tape + tp.xfin_all), D, w->norm_g, g, D, g, D, nullptr, nullptr, 0, 0, 0, nullptr,
                              F(grads->norm_g), F(grads->norm_b), 1, wsf, wsn, B * Nl, D, cfg->ln_eps, s));
    }
    TR_TRY(tr_f32_to_bf16(g, gb, (size_t)B * Nl * D, s));
    if (cfg->family == TR_FAMILY_DYVIT)
      TR_REQUIRE(hipMemsetAsync(ws + bp.dpol, 0, (size_t)B * t.N0 * 4, st) == hipSuccess, TR_ERR_LAUNCH, "tr_vit_backward: memset failed");
  }

  // dropout keep masks: the forward consumed them in order (tr_vit_dropout_mask_bytes): [pos | block 0: proj, hidden, fc2 | block 1: ...]
  size_t drop_off[TR_MAX_DEPTH + 1];
  drop_off[0] = (size_t)B * t.N0 * D;
  for (int i = 0; i < cfg->depth; ++i) drop_off[i + 1] = drop_off[i] + (size_t)B * (trplan::proj_rows(cfg, t, i) * D + (size_t)t.n_mlp[i] * (Hd + D));
  struct LnDeferScope {       // every norm of the block loop leaves its d gamma / d beta partials in bp.lnpart; one reduce launch after the loop
    LnDeferScope(float* r, size_t n, hipStream_t st) { tr_ln_defer_begin(r, n, st); }
    ~LnDeferScope() { tr_ln_defer_end(); }
  } ln_scope(reinterpret_cast<float*>(ws + bp.lnpart), bp.lnpart_floats, st);
  for (int i = blk_hi; i >= blk_lo; --i) {
    const trplan::BlockTape& bt = tp.blk[i];
    const tr_block_weights* bw = &w->blocks[i];
    const tr_block_weights* bwt = &wt->blocks[i];
    const tr_block_weights* bg = &grads->blocks[i];
    const int Na = t.n_att[i], Nm = t.n_mlp[i];
    const int M2 = B * Nm, M1 = B * Na;
    // ---- mlp: x2 -> norm2 -> fc1 -> gelu -> fc2 -> (+ residual)
    const uint16_t* gy = gb;          // the branch's dY: the stream gradient, times the branch's DropPath scale when there is one
    if (drop_scale != nullptr) {
      TR_TRY(tr_rowscale_bf16(gb, dao, drop_scale + (size_t)(2 * i + 1) * B, B, Nm, D, s));
      gy = dao;
    }
    const uint8_t* keep_proj = dropout_keep ? dropout_keep + drop_off[i] : nullptr;                                  // [B*n_proj, D]
    const uint8_t* keep_h = dropout_keep ? keep_proj + (size_t)B * trplan::proj_rows(cfg, t, i) * D : nullptr;        // [M2, Hd]
    const uint8_t* keep_fc2 = dropout_keep ? keep_h + (size_t)M2 * Hd : nullptr;                                      // [M2, D]
    if (dropout_keep != nullptr) {    // the Mlp's second dropout (after fc2): the same mask on fc2's dY
      TR_TRY(tr_dropout_bf16(gy, dao, keep_fc2, drop_mul, (size_t)M2 * D, s));
      gy = dao;
    }
    TR_TRY(tr_gemm_dgelu_bf16(gy, U(bwt->fc2_w), U(tape + bt.pre), dh, M2, Hd, D, s));          // d fc2 input, times gelu'(pre): d pre
    if (dropout_keep != nullptr)      // ... and the first one (after the activation): d pre = (dY W) * keep/(1-p) * gelu'(pre), the factors commute
      TR_TRY(tr_dropout_bf16(dh, dh, keep_h, drop_mul, (size_t)M2 * Hd, s));
    // fc2's and fc1's parameter gradients: both dY (gy, dh) exist now; launched here as a pair, or with proj's and qkv's further down
    tr_linear_grad LG[4];
    LG[0] = {gy, (long)D, U(tape + bt.h), (long)Hd, F(bg->fc2_w), F(bg->fc2_b), M2, D, Hd};
    LG[1] = {dh, (long)Hd, U(tape + bt.xn2), (long)D, F(bg->fc1_w), F(bg->fc1_b), M2, Hd, D};
    if (!four) TR_TRY(tr_linear_bwd_group(LG, 2, acc, wsf, wsn, s));
    TR_TRY(tr_gemm_bf16(dh, U(bwt->fc1_w), zeros, dxn, nullptr, 0, M2, D, Hd, TR_EPI_BF16, s));
    // ---- norm2 (+ the block's in-block token reduction)
    const float* x2 = reinterpret_cast<const float*>(tape + bt.x2);
    const float* x1 = reinterpret_cast<const float*>(tape + bt.x1);
    const float* dcls = nullptr;
    const int K = t.kk[i];
    if ((cfg->family == TR_FAMILY_TOPK || cfg->family == TR_FAMILY_EVIT) && K > 0) {
      const bool fuse = cfg->family == TR_FAMILY_EVIT;
      const int32_t* idx = reinterpret_cast<const int32_t*>(tape + bt.idx);
      TR_REQUIRE(hipMemsetAsync(g_alt, 0, (size_t)M1 * D * 4, st) == hipSuccess && hipMemsetAsync(gb_alt, 0, (size_t)M1 * D * 2, st) == hipSuccess,
                 TR_ERR_LAUNCH, "tr_vit_backward: memset failed");
      TR_TRY(tr_layernorm_bwd(dxn, x2, D, bw->ln2_g, g, D, g_alt, D, gb_alt, idx, K, Nm, Na, fuse ? gfused : nullptr, F(bg->ln2_g), F(bg->ln2_b),
                              acc, wsf, wsn, M2, D, cfg->ln_eps, s));
      if (fuse) {
        TR_REQUIRE(hipMemsetAsync(dscore, 0, (size_t)M1 * 4, st) == hipSuccess, TR_ERR_LAUNCH, "tr_vit_backward: memset failed");
        TR_TRY(tr_evit_fuse_bwd(x1, U(tape + bt.dattn), reinterpret_cast<const int32_t*>(tape + bt.idx2),
                                reinterpret_cast<const float*>(tape + bt.scores), gfused, g_alt, gb_alt, dscore, B, Na, K, D, s));
        dcls = dscore;
      }
      float* tg = g; g = g_alt; g_alt = tg;
      uint16_t* tb = gb; gb = gb_alt; gb_alt = tb;
    } else if (cfg->family == TR_FAMILY_TOME && K > 0) {
      TR_TRY(tr_layernorm_bwd(dxn, x2, D, bw->ln2_g, g, D, g, D, nullptr, nullptr, 0, 0, 0, nullptr, F(bg->ln2_g), F(bg->ln2_b), acc, wsf, wsn, M2, D,
                              cfg->ln_eps, s));
      const int na = (Na + 1) / 2;
      const int32_t* unm = reinterpret_cast<const int32_t*>(tape + bt.idx);
      const int32_t* src = unm + (size_t)B * (na - K);
      const int32_t* dst = src + (size_t)B * K;
      const float* size_in = nullptr;                       // sizes entering this block's merge: after the previous merging block
      for (int j = i - 1; j >= 0; --j)
        if (t.kk[j] > 0) { size_in = reinterpret_cast<const float*>(tape + tp.blk[j].size); break; }
      TR_TRY(tr_tome_merge_bwd(g, size_in, reinterpret_cast<const float*>(tape + bt.size), unm, src, dst, invmap, g_alt, gb_alt, B, Na, K, D, s));
      float* tg = g; g = g_alt; g_alt = tg;
      uint16_t* tb = gb; gb = gb_alt; gb_alt = tb;
    } else {
      TR_TRY(tr_layernorm_bwd(dxn, x2, D, bw->ln2_g, g, D, g, D, four ? gb_spare : gb, nullptr, 0, 0, 0, nullptr, F(bg->ln2_g), F(bg->ln2_b), acc, wsf,
                              wsn, M2, D, cfg->ln_eps, s));
      if (four) { uint16_t* tb = gb; gb = gb_spare; gb_spare = tb; }       // gb_spare: fc2's dY, until this block's weight-gradient launch
    }
    // ---- attention: x1 -> norm1 -> qkv -> softmax(q k^T) v -> proj -> (+ residual)
    const bool ats_sampled = cfg->family == TR_FAMILY_ATS && K > 0;
    const int Mp = ats_sampled ? M2 : M1;                   // rows that went through proj (ATS: only the sampled ones, ats.py:86,129)
    gy = gb;
    if (drop_scale != nullptr) {
      TR_TRY(tr_rowscale_bf16(gb, dh, drop_scale + (size_t)(2 * i) * B, B, Mp / B, D, s));
      gy = dh;
    }
    if (dropout_keep != nullptr) {    // proj_drop (topk.py:53): the same mask on proj's dY
      TR_TRY(tr_dropout_bf16(gy, dh, keep_proj, drop_mul, (size_t)Mp * D, s));
      gy = dh;
    }
    const uint16_t* gy_proj = gy;           // proj's dY: stays valid until norm1's backward rewrites gb (ATS: the swapped-out buffer)
    if (ats_sampled) {
      // d(attn @ v) of the sampled rows and the stream's gradient go back to the rows they were sampled from (ats.py:86,157)
      TR_TRY(tr_gemm_bf16(gy, U(bwt->proj_w), zeros, dxn, nullptr, 0, Mp, D, D, TR_EPI_BF16, s));
      TR_REQUIRE(hipMemsetAsync(g_alt, 0, (size_t)M1 * D * 4, st) == hipSuccess && hipMemsetAsync(dao, 0, (size_t)M1 * D * 2, st) == hipSuccess,
                 TR_ERR_LAUNCH, "tr_vit_backward: memset failed");
      TR_TRY(tr_ats_scatter(g, dxn, reinterpret_cast<const int32_t*>(tape + bt.idx), g_alt, dao, B, Na, Nm, D, s));
      float* tg = g; g = g_alt; g_alt = tg;
      uint16_t* tb = gb; gb = gb_alt; gb_alt = tb;           // gb is rewritten by norm1's backward below
    } else {
      TR_TRY(tr_gemm_bf16(gy, U(bwt->proj_w), zeros, dao, nullptr, 0, M1, D, D, TR_EPI_BF16, s));
    }
    const float* size_att = nullptr;       // ToMe: log(size) bias of this block's keys (tome.py:48-49); ATS: the key mask (ats.py:117-120)
    if (cfg->family == TR_FAMILY_TOME || cfg->family == TR_FAMILY_ATS)
      for (int j = i - 1; j >= 0; --j)
        if (t.kk[j] > 0) { size_att = reinterpret_cast<const float*>(tape + tp.blk[j].size); break; }
    if (cfg->family == TR_FAMILY_HEURISTIC)          // the spatial key mask in force at this block: the last one set at or before it
      for (int j = i; j >= 0; --j)
        if (w->stage[j].w3 != nullptr) { size_att = reinterpret_cast<const float*>(tape + tp.blk[j].size); break; }
    if (cfg->family == TR_FAMILY_DYVIT) {
      // the policy this block attended under: the last predictor stage at or before it, all ones before the first
      const float* pol = reinterpret_cast<const float*>(tape + tp.ones);
      for (int j = i; j >= 0; --j)
        if (t.kk[j] > 0) { pol = reinterpret_cast<const float*>(tape + tp.blk[j].pol); break; }
      float* dpart = reinterpret_cast<float*>(ws + bp.dpolpart);
      if (Na > 224)      // 384 x 384 inputs: the key-blocked kernels' policy variant
        TR_TRY(tr_attention_policy_bwd_long_bf16(U(tape + bt.qkv), dao, pol, dqkv, dpart, reinterpret_cast<float*>(ws + bp.attn_stats),
                                                 tr_attention_bwd_long_workspace_floats(B, t.N0, H), B, Na, H, s));
      else
        TR_TRY(tr_attention_policy_bwd_bf16(U(tape + bt.qkv), dao, pol, dqkv, dpart, B, Na, H, s));
      TR_TRY(tr_head_sum(dpart, reinterpret_cast<float*>(ws + bp.dpol), B, H, Na, s));
    } else if (Na > 224) {       // 384 x 384 inputs: key-blocked kernels (tr_attention_bwd_long.hip)
      TR_TRY(tr_attention_bwd_long_bf16(U(tape + bt.qkv), dao, size_att, dcls, dqkv, reinterpret_cast<float*>(ws + bp.attn_stats),
                                        tr_attention_bwd_long_workspace_floats(B, t.N0, H), B, Na, H, s));
    } else {
      TR_TRY(tr_attention_bwd_bf16(U(tape + bt.qkv), dao, size_att, dcls, dqkv, B, Na, H, s));
    }
    // the block's parameter gradients (dY: the branch gradients kept above, dh and dqkv): one launch for all four, or the second pair
    LG[2] = {gy_proj, (long)D, U(tape + bt.ao), (long)D, F(bg->proj_w), F(bg->proj_b), Mp, D, D};
    LG[3] = {dqkv, (long)(3 * D), U(tape + bt.xn1), (long)D, F(bg->qkv_w), F(bg->qkv_b), M1, 3 * D, D};
    if (four) TR_TRY(tr_linear_bwd_group(LG, 4, acc, wsf, wsn, s));
    else TR_TRY(tr_linear_bwd_group(LG + 2, 2, acc, wsf, wsn, s));
    TR_TRY(tr_gemm_bf16(dqkv, U(bwt->qkv_w), zeros, dxn, nullptr, 0, M1, D, 3 * D, TR_EPI_BF16, s));
    if (cfg->family == TR_FAMILY_KMEDOIDS && K > 0) {
      // norm1 ran on the gathered medoid rows (kmedoids.py:243-248): its backward scatter-ADDS into the pre-reduction stream's gradient
      const size_t nfull = (size_t)B * t.n_pre[i] * D;
      TR_REQUIRE(hipMemsetAsync(g_alt, 0, nfull * 4, st) == hipSuccess, TR_ERR_LAUNCH, "tr_vit_backward: memset failed");
      TR_TRY(tr_layernorm_bwd_scatter_add(dxn, x1, bw->ln1_g, g, g_alt, reinterpret_cast<const int32_t*>(tape + bt.idx), K, t.n_pre[i], F(bg->ln1_g),
                                          F(bg->ln1_b), acc, wsf, wsn, M1, D, cfg->ln_eps, s));
      TR_TRY(tr_f32_to_bf16(g_alt, gb_alt, nfull, s));
      float* tg = g; g = g_alt; g_alt = tg;
      uint16_t* tb = gb; gb = gb_alt; gb_alt = tb;
    } else {
      TR_TRY(tr_layernorm_bwd(dxn, x1, D, bw->ln1_g, g, D, g, D, gb, nullptr, 0, 0, 0, nullptr, F(bg->ln1_g), F(bg->ln1_b), acc, wsf, wsn, M1, D,
                              cfg->ln_eps, s));
    }
    if (cfg->family == TR_FAMILY_DYVIT && K > 0) {
      // PredictorLG + Gumbel straight-through of this stage (dyvit.py:221-224), backwards.  d keep = the policy gradient collected
      // from the blocks that attended under this stage's policy (+ the later stage's d prev_decision) + d out_pred_prob
      const tr_stage_weights* sw = &w->stage[i];
      const tr_stage_weights* swt = &wt->stage[i];
      const tr_stage_weights* sg = &grads->stage[i];
      // Hr: the predictor's real hidden width (the shapes of the parameter gradients); Hh: as packed and as laid out on the tape
      // (DeiT-T: 96 -> 128 with zero weights: the padded columns of every activation and gradient are zero)
      const int Hr = D / 2, Hh = (D / 2 + 63) / 64 * 64, Q = (D / 4 + 63) / 64 * 64, Cq = D / 4;
      float* dpol = reinterpret_cast<float*>(ws + bp.dpol);
      float* dprev = reinterpret_cast<float*>(ws + bp.dprev);
      int stage = 0;
      for (int j = 0; j < i; ++j) stage += t.kk[j] > 0 ? 1 : 0;
      if (dpred != nullptr) TR_TRY(tr_add_patch_rows(dpol, dpred + (size_t)stage * B * t.P, B, Na, s));
      const float* prev = reinterpret_cast<const float*>(tape + tp.ones);       // prev_decision entering this stage, [B,N] layout
      for (int j = i - 1; j >= 0; --j)
        if (t.kk[j] > 0) { prev = reinterpret_cast<const float*>(tape + tp.blk[j].pol); break; }
      TR_REQUIRE(hipMemsetAsync(dprev, 0, (size_t)M1 * 4, st) == hipSuccess, TR_ERR_LAUNCH, "tr_vit_backward: memset failed");
      uint16_t* d2 = dqkv;                    // scratch: [M1, Q] then reused
      TR_TRY(tr_dyvit_decide_bwd(dpol, prev, reinterpret_cast<const float*>(tape + bt.hard), reinterpret_cast<const float*>(tape + bt.ysoft),
                                 reinterpret_cast<const float*>(tape + bt.sm), U(tape + bt.ph2), Q, sw->w3, d2, dprev, F(sg->w3), F(sg->b3), acc, wsf,
                                 wsn, B, Na, Cq, s));
      TR_TRY(tr_gelu_bwd_bf16(U(tape + bt.ppre2), d2, (size_t)M1 * Q, s));
      TR_TRY(tr_linear_bwd_params(d2, Q, 0, U(tape + bt.ph1), Hh, F(sg->w2), F(sg->b2), acc, wsf, wsn, M1, Cq, Hr, s));
      uint16_t* d1 = dao;                     // [M1, Hh]
      TR_TRY(tr_gemm_dgelu_bf16(d2, U(swt->w2), U(tape + bt.ppre1), d1, M1, Hh, Q, s));
      TR_TRY(tr_linear_bwd_params(d1, Hh, 0, U(tape + bt.pcat), D, F(sg->w1), F(sg->b1), acc, wsf, wsn, M1, Hr, D, s));
      TR_TRY(tr_gemm_bf16(d1, U(swt->w1), zeros, dxn, nullptr, 0, M1, D, Hh, TR_EPI_BF16, s));          // d [local | global]
      uint16_t* d0 = dh;                      // [M1, D]
      TR_TRY(tr_pool_policy_bwd(dxn, U(tape + bt.ppre0), U(tape + bt.pcat), prev, d0, dprev, B, Na, D, s));
      TR_TRY(tr_gelu_bwd_bf16(U(tape + bt.ppre0), d0, (size_t)M1 * D, s));
      TR_TRY(tr_linear_bwd_params(d0, D, 0, U(tape + bt.pu), D, F(sg->w0), F(sg->b0), acc, wsf, wsn, M1, D, D, s));
      TR_TRY(tr_gemm_bf16(d0, U(swt->w0), zeros, dxn, nullptr, 0, M1, D, D, TR_EPI_BF16, s));
      TR_TRY(tr_layernorm_bwd(dxn, reinterpret_cast<const float*>(tape + bt.x0), D, sw->ln_g, g, D, g, D, gb, nullptr, 0, 0, 0, nullptr, F(sg->ln_g),
                              F(sg->ln_b), acc, wsf, wsn, M1, D, 1e-5f, s));
      // what is left of the policy gradient belongs to the previous stage's decision
      TR_REQUIRE(hipMemcpyAsync(dpol, dprev, (size_t)M1 * 4, hipMemcpyDeviceToDevice, st) == hipSuccess, TR_ERR_LAUNCH, "tr_vit_backward: copy failed");
    }
    if (trplan::soft_family(cfg->family) && K > 0) {
      // soft-assignment stage before the block (sit.py:36-40, patchmerger.py:35-39, sinkhorn.py:66-86): g = d of the merged stream
      // [B, K+1, D] -> the stream entering the stage [B, n_pre, D] + the stage's parameters (tr_soft_bwd.hip)
      const tr_stage_weights* sw = &w->stage[i];
      const tr_stage_weights* swt = &wt->stage[i];
      const tr_stage_weights* sg = &grads->stage[i];
      const int Np = t.n_pre[i], Mp = B * Np, ld = trplan::soft_ld(K), ld64 = trplan::soft_ld64(K);
      const bool sit = cfg->family == TR_FAMILY_SIT, sink = cfg->family == TR_FAMILY_SINKHORN;
      const float* x0 = reinterpret_cast<const float*>(tape + bt.x0);
      const float* wts = reinterpret_cast<const float*>(tape + bt.swt);
      const float* slog = reinterpret_cast<const float*>(tape + bt.slog);
      const float* src = sit ? x0 : reinterpret_cast<const float*>(tape + bt.sxh);
      float* dwt = reinterpret_cast<float*>(ws + bp.soft_dp);
      uint16_t* ds = reinterpret_cast<uint16_t*>(ws + bp.soft_ds);
      float* dsrc = sit ? g_alt : reinterpret_cast<float*>(ws + bp.soft_s);       // SiT sums the stream's own rows: d src IS a stream gradient
      TR_REQUIRE(hipMemsetAsync(ds, 0, (size_t)Mp * ld64 * 2, st) == hipSuccess && hipMemsetAsync(dsrc, 0, (size_t)Mp * D * 4, st) == hipSuccess,
                 TR_ERR_LAUNCH, "tr_vit_backward: memset failed");
      TR_TRY(tr_soft_dweights(g, src, dwt, ld, B, Np, K, D, s));
      TR_TRY(tr_soft_dsrc(g, wts, ld, dsrc, B, Np, K, D, s));
      if (sink)
        TR_TRY(tr_sinkhorn_bwd(slog, dwt, ld, cfg->sinkhorn_eps > 0.f ? cfg->sinkhorn_eps : 1.0f, cfg->cluster_iters, ds, ld64, B, Np, K, s));
      else
        TR_TRY(tr_token_softmax_bwd(wts, dwt, slog, ld, sw->scale, ds, ld64, sit ? F(sg->b2) : nullptr, acc, wsf, wsn, B, Np, K, s));
      if (sit) {
        const int Hr = D / 2, Hh = (D / 2 + 63) / 64 * 64;     // real / packed hidden width (DeiT-T: 96 / 128)
        uint16_t* d1 = dh;                    // [Mp, Hh]
        TR_TRY(tr_linear_bwd_params(ds, ld64, 0, U(tape + bt.pcat), Hh, F(sg->w1), F(sg->b1), acc, wsf, wsn, Mp, ld, Hr, s));
        TR_TRY(tr_gemm_dgelu_bf16(ds, U(swt->w1), U(tape + bt.ppre0), d1, Mp, Hh, ld64, s));
        TR_TRY(tr_linear_bwd_params(d1, Hh, 0, U(tape + bt.pu), D, F(sg->w0), F(sg->b0), acc, wsf, wsn, Mp, Hr, D, s));
        TR_TRY(tr_gemm_bf16(d1, U(swt->w0), zeros, dxn, nullptr, 0, Mp, D, Hh, TR_EPI_BF16, s));
      } else {
        // the similarity / score product: d queries (d centres) and d of its token operand
        TR_TRY(tr_wgrad_bf16(ds, ld64, 0, U(tape + bt.pu), D, F(sg->w1), acc, wsf, wsn, Mp, ld, D, s));
        TR_TRY(tr_gemm_bf16(ds, U(swt->w1), zeros, dxn, nullptr, 0, Mp, D, ld64, TR_EPI_BF16, s));
      }
      // the CLS row passes the stage untouched
      if (!sit) TR_REQUIRE(hipMemsetAsync(g_alt, 0, (size_t)Mp * D * 4, st) == hipSuccess, TR_ERR_LAUNCH, "tr_vit_backward: memset failed");
      if (sink) {
        TR_TRY(tr_rownorm_bwd(x0, dsrc, dxn, g_alt, Mp, D, s));                  // d unit-norm rows = merge part + score-product part
        TR_REQUIRE(hipMemcpy2DAsync(g_alt, (size_t)Np * D * 4, g, (size_t)(K + 1) * D * 4, (size_t)D * 4, B, hipMemcpyDeviceToDevice, st) == hipSuccess,
                   TR_ERR_LAUNCH, "tr_vit_backward: copy failed");
        TR_TRY(tr_f32_to_bf16(g_alt, gb_alt, (size_t)Mp * D, s));
      } else {
        TR_REQUIRE(hipMemcpy2DAsync(g_alt, (size_t)Np * D * 4, g, (size_t)(K + 1) * D * 4, (size_t)D * 4, B, hipMemcpyDeviceToDevice, st) == hipSuccess,
                   TR_ERR_LAUNCH, "tr_vit_backward: copy failed");
        if (!sit) TR_TRY(tr_add_into_bf16(dsrc, dxn, (size_t)Mp * D, s));        // PatchMerger sums the LayerNorm output: both uses meet here
        TR_TRY(tr_layernorm_bwd(dxn, x0, D, sw->ln_g, g_alt, D, g_alt, D, gb_alt, nullptr, 0, 0, 0, nullptr, F(sg->ln_g), F(sg->ln_b), acc, wsf,
                                wsn, Mp, D, 1e-5f, s));
      }
      float* tg = g; g = g_alt; g_alt = tg;
      uint16_t* tb = gb; gb = gb_alt; gb_alt = tb;
    }
    if (cfg->family == TR_FAMILY_DPCKNN && K > 0) {
      // CTM before the block (dpcknn.py:257-260): gradient of the merged tokens -> the tokens they were merged from + the score Linear
      const tr_stage_weights* sw = &w->stage[i];
      const tr_stage_weights* sg = &grads->stage[i];
      TR_TRY(tr_cluster_merge_bwd(g, reinterpret_cast<const float*>(tape + bt.x0), x1, reinterpret_cast<const float*>(tape + bt.scores),
                                  reinterpret_cast<const int32_t*>(tape + bt.idx2), sw->w3, g_alt, gb_alt, F(sg->w3), F(sg->b3), acc, wsf, wsn, B,
                                  t.n_pre[i], K, D, s));
      float* tg = g; g = g_alt; g_alt = tg;
      uint16_t* tb = gb; gb = gb_alt; gb_alt = tb;
    }
  }
  TR_TRY(tr_ln_defer_flush());
  if (blk_lo > 0) return TR_OK;
  // ---- embedding (topk.py:181-186): g is d x0 [B, N0, D]
  if (dropout_keep != nullptr) {      // pos_drop (topk.py:186): on the stream gradient and on its bf16 copy (the patch projection's dY)
    TR_TRY(tr_dropout_f32(g, g, dropout_keep, drop_mul, (size_t)B * t.N0 * D, s));
    TR_TRY(tr_dropout_bf16(gb, gb, dropout_keep, drop_mul, (size_t)B * t.N0 * D, s));
  }
  TR_TRY(tr_embed_bwd(g, F(grads->pos_embed), F(grads->cls_token), acc, B, t.N0, D, s));
  TR_TRY(tr_linear_bwd_params(gb, D, t.P, U(tape + tp.cols), kcols, F(grads->patch_w), F(grads->patch_b), acc, wsf, wsn, B * t.P, D, kcols, s));
  return TR_OK;
}
