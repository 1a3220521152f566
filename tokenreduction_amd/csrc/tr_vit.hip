// Whole-model executor: enqueues the DeiT / Top-K / EViT forward pass on the caller's stream.
//
// Follows TopKVisionTransformer.forward (topk.py:179-212), EfficientVisionTransformer.forward
// (evit.py:209-244) and deit_viz.VisionTransformer.forward (:186-212), eval mode:
//   patch_embed -> cat(cls) + pos_embed -> 12 x Block -> norm -> x[:,0] -> head
// with Block_TopK.forward (topk.py:83-99):  x += attn(norm1(x)); [Top-K gather]; x += mlp(norm2(x)).
//
// Host-side only: shape bookkeeping and kernel launches (no allocation, no synchronisation, no
// device->host copies), so one call is a fixed launch sequence that a caller may capture in a hipGraph.
// Token counts are static per (config) -- topk.py:56 int(ratio*196) -- so every buffer size is known up front.
#include <stdarg.h>
#include <string.h>
#include <vector>
#include "tr_common.h"
#include <mutex>
#include "tr_plan.h"

static thread_local char g_err[512] = "";

void tr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tr_last_error(void) { return g_err; }

// ---- launch profiler (see tr_common.h) ---------------------------------------------------------------------------------
namespace {
struct ProfRec { char label[48]; double flops, bytes; };
struct Prof {
  std::atomic<bool> on{false};
  hipStream_t st = nullptr;
  std::vector<hipEvent_t> ev;       // ev[0] = begin, ev[i+1] = after mark i
  std::vector<ProfRec> recs;
  size_t used = 0;
};
// ONE recording per process, whatever thread launches: a training step's forward runs on the caller's thread, its backward on the
// autograd engine's device thread, and a recording started by the caller must see both (round 3: per-thread state recorded a third of
// the step).  The launches of a recording are sequential on one stream; the mutex only keeps the vectors consistent.
Prof g_prof;
std::mutex g_prof_mu;
thread_local ProfRec t_note;        // the pending note -> mark pair of this thread's current launch
thread_local bool t_noted = false;
}  // namespace

void tr_prof_note(const char* label, double flops, double bytes) {
  if (!g_prof.on.load(std::memory_order_relaxed)) return;
  snprintf(t_note.label, sizeof(t_note.label), "%s", label);
  t_note.flops = flops;
  t_note.bytes = bytes;
  t_noted = true;
}

void tr_prof_mark(const char* label) {
  Prof& p = g_prof;
  if (!p.on.load(std::memory_order_relaxed)) return;
  ProfRec r;
  if (t_noted) r = t_note;
  else { snprintf(r.label, sizeof(r.label), "%s", label); r.flops = 0; r.bytes = 0; }
  t_noted = false;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!p.on.load(std::memory_order_relaxed)) return;
  if (p.used + 1 >= p.ev.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    p.ev.push_back(e);
  }
  (void)hipEventRecord(p.ev[p.used + 1], p.st);
  ++p.used;
  p.recs.push_back(r);
}

// Called at the top of the executors: when a recording is active and nothing has been marked yet, the opening event is taken again
// HERE, so the first mark does not include the host time between tr_profile_begin and the executor's first launch.
void tr_prof_restart() {
  Prof& p = g_prof;
  if (!p.on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (p.on.load(std::memory_order_relaxed) && p.used == 0) (void)hipEventRecord(p.ev[0], p.st);
}

// Start recording the launches the process enqueues on stream s through this library (must not be capturing).
extern "C" int tr_profile_begin(tr_stream_t s) {
  Prof& p = g_prof;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (p.ev.empty()) {
    hipEvent_t e;
    TR_REQUIRE(hipEventCreate(&e) == hipSuccess, TR_ERR_LAUNCH, "tr_profile_begin: cannot create an event");
    p.ev.push_back(e);
  }
  p.st = static_cast<hipStream_t>(s);
  p.recs.clear();
  p.used = 0;
  t_noted = false;
  TR_REQUIRE(hipEventRecord(p.ev[0], p.st) == hipSuccess, TR_ERR_LAUNCH, "tr_profile_begin: event record failed");
  p.on.store(true);
  return TR_OK;
}

// Stop, wait for the stream, and return up to `max` marks: label (48 chars each), ms since the previous mark, FLOPs, bytes.
// Returns the number of marks recorded (may exceed max; only max are written), or < 0 on error.
extern "C" int tr_profile_end(int max, char* labels, float* ms, double* flops, double* bytes) {
  Prof& p = g_prof;
  TR_REQUIRE(p.on.load(), TR_ERR_CONFIG, "tr_profile_end: no recording is active");
  p.on.store(false);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  TR_REQUIRE(hipStreamSynchronize(p.st) == hipSuccess, TR_ERR_LAUNCH, "tr_profile_end: stream synchronize failed");
  const int n = (int)p.recs.size();
  for (int i = 0; i < n && i < max; ++i) {
    float t = 0.f;
    (void)hipEventElapsedTime(&t, p.ev[i], p.ev[i + 1]);
    if (labels) memcpy(labels + (size_t)i * 48, p.recs[i].label, 48);
    if (ms) ms[i] = t;
    if (flops) flops[i] = p.recs[i].flops;
    if (bytes) bytes[i] = p.recs[i].bytes;
  }
  return n;
}
extern "C" int tr_version(void) { return 100; }

int tr_mlp_fused_wanted(int M, int D, int Hd, int have_scratch, int concurrent);      // tr_mlp_fused.hip: the schedule policy behind tr_set_mlp_fused
int tr_mlp_resid_ln_enabled();                                        // tr_mlp_fused.hip: tr_set_mlp_resid_ln's switch
// ... and its launches on a counter set that ONE memset in front of the forward has zeroed (block i uses set i)
int tr_mlp_fused_zero_counters(void* scratch, size_t scratch_bytes, int D, int Hd, int nsets, tr_stream_t s);
int tr_mlp_fused_bf16_set(const uint16_t* xn, const void* packed, const float* fc1_b, uint16_t* out, void* scratch, size_t scratch_bytes, int M, int D,
                          int Hd, int cset, tr_stream_t s);
int tr_mlp_fused_ln_bf16_set(const float* x, const uint16_t* delta, const float* g, const float* b, float eps, const void* packed, const float* fc1_b,
                             uint16_t* out, void* scratch, size_t scratch_bytes, int M, int D, int Hd, int cset, tr_stream_t s);
int tr_mlp_fused_resid_ln_bf16_set(const uint16_t* xn, const void* packed, const float* fc1_b, const float* fc2_b, float* x, const float* next_g,
                                   const float* next_b, float eps, uint16_t* xn_next, void* scratch, size_t scratch_bytes, int M, int D, int Hd,
                                   int cset, tr_stream_t s);
int tr_mlp_ln_wanted(int M, int D, int Hd, int have_scratch, int concurrent);         // tr_mlp_fused.hip: ... with the norm2 in front of it inside the launch (tr_set_mlp_ln)

namespace {

using trplan::align_up;

struct Plan {
  int P, N0, D, H, Hd, C, kcols;
  size_t off_x0, off_x1, off_xn, off_qkv, off_ao, off_h, off_d, off_d2, off_cols, off_cls, off_scores, off_idx, off_compl, off_xcls, off_size0, off_size1, off_cluster, off_soft, off_mlp_sk, mlp_sk_bytes, off_misc, total;
};

bool make_plan(const tr_vit_config* c, int B, Plan* p) {
  if (!c || B <= 0) return false;
  if (c->precision != TR_PREC_BF16 && c->precision != TR_PREC_FP32 && c->precision != TR_PREC_BF16X3) return false;
  const size_t es = c->precision != TR_PREC_BF16 ? 4 : 2;   // activation element size
  if (c->patch <= 0 || c->img_size <= 0 || c->img_size % c->patch != 0) return false;
  if (c->depth <= 0 || c->depth > TR_MAX_DEPTH) return false;
  if (c->num_heads <= 0 || c->embed_dim != c->num_heads * 64) return false;
  if (c->family < TR_FAMILY_DEIT || c->family > TR_FAMILY_HEURISTIC) return false;
  const int g = c->img_size / c->patch;
  p->P = g * g;
  p->N0 = p->P + 1;
  p->D = c->embed_dim;
  p->H = c->num_heads;
  p->Hd = c->mlp_hidden;
  p->C = c->num_classes;
  p->kcols = c->in_chans * c->patch * c->patch;
  if (p->kcols % 64 || p->D % 64 || p->Hd % 64 || p->C % 4 || p->Hd <= 0 || p->C <= 0) return false;
  const size_t T = (size_t)B * p->N0;  // max tokens
  size_t o = 0;
  p->off_x0 = o;     o += align_up(T * p->D * 4);
  p->off_x1 = o;     o += align_up(T * p->D * 4);
  p->off_xn = o;     o += align_up(T * p->D * es);
  p->off_qkv = o;    o += align_up(T * 3 * p->D * es);
  p->off_ao = o;     o += align_up(T * p->D * es);
  p->off_h = o;      o += align_up(T * p->Hd * es);
  p->off_d = o;      o += align_up(T * p->D * es);
  p->off_d2 = o;     o += align_up(T * p->D * es);      // second residual buffer (norm2 without a stream write, see vit_forward_impl)
  p->off_cols = o;   o += align_up((size_t)B * p->P * p->kcols * es);
  p->off_cls = o;    o += align_up((size_t)B * p->H * p->N0 * 4);
  p->off_scores = o; o += align_up((size_t)B * p->N0 * 4);
  p->off_idx = o;    o += align_up((size_t)B * p->N0 * 4);
  p->off_compl = o;  o += align_up((size_t)B * p->N0 * 4);
  p->off_size0 = o;  o += align_up((size_t)B * p->N0 * 4);
  p->off_size1 = o;  o += align_up((size_t)B * p->N0 * 4);
  p->off_xcls = o;   o += align_up((size_t)B * p->D * es);
  p->off_cluster = o;
  if (c->family == TR_FAMILY_DPCKNN) o += align_up(tr_dpcknn_workspace_floats(B, p->N0) * 4 + (size_t)B * p->N0 * 4);
  if (c->family == TR_FAMILY_KMEDOIDS)
    o += align_up(tr_dpcknn_workspace_floats(B, p->N0) * 4) + align_up((size_t)B * p->H * 4 * p->N0 * 4);
  // soft-assignment families: token-major logits / scores / transport plan of a stage, fp32 [B*N0, soft_ld(K)]
  p->off_soft = o;
  if (trplan::soft_family(c->family)) {
    int kmax = 0;
    for (int i = 0; i < c->depth; ++i) kmax = c->keep[i] > kmax ? c->keep[i] : kmax;
    o += align_up(T * (size_t)trplan::soft_ld(kmax) * 4);
  }
  // stream-K scratch of the fused eval Mlp (tr_mlp_fused.hip): one accumulator slot of 192 KiB + a counter per compute unit, bf16 executor only
  p->off_mlp_sk = o;
  p->mlp_sk_bytes = c->precision == TR_PREC_BF16 ? tr_mlp_fused_scratch_bytes(p->D, p->Hd) : 0;
  o += align_up(p->mlp_sk_bytes);
  p->off_misc = o;          // device words read back by the host (ATS dynamic width)
  o += align_up(256);
  p->total = o;
  return true;
}

// ---- precision dispatch: the executor is one launch sequence; TR_PREC_FP32 swaps every op for its fp32 validation twin;
// TR_PREC_BF16X3 is that fp32 executor with the Linears and the attention on the matrix cores as split-bf16 products (tr_split.hip)
inline int op_im2col(bool f32, const float* img, void* cols, int B, int C, int H, int W, int patch, tr_stream_t s) {
  return f32 ? tr_im2col_f32(img, static_cast<float*>(cols), B, C, H, W, patch, s)
             : tr_im2col_bf16(img, static_cast<uint16_t*>(cols), B, C, H, W, patch, s);
}
inline int op_gemm(int prec, const void* A, const void* W, const float* bias, void* out, const float* aux, int aux_i, int M, int N,
                   int K, int epi, tr_stream_t s) {
  if (prec == TR_PREC_BF16)
    return tr_gemm_bf16(static_cast<const uint16_t*>(A), static_cast<const uint16_t*>(W), bias, out, aux, aux_i, M, N, K, epi, s);
  const int e32 = (epi == TR_EPI_BF16) ? TR_EPI_F32 : epi;      // "store bf16" becomes "store fp32"; GELU / PATCH / F32 keep their meaning
  if (prec == TR_PREC_BF16X3)
    return tr_gemm_split(static_cast<const float*>(A), static_cast<const float*>(W), bias, static_cast<float*>(out), aux, aux_i, M, N, K,
                         e32, s);
  return tr_gemm_f32(static_cast<const float*>(A), static_cast<const float*>(W), bias, static_cast<float*>(out), aux, aux_i, M, N, K,
                     e32, s);
}
inline int op_ln(bool f32, float* x, long ldx, const void* d, long ldd, const float* g, const float* b, void* y, int M, int D,
                 float eps, tr_stream_t s) {
  return f32 ? tr_layernorm_f32(x, ldx, static_cast<const float*>(d), ldd, g, b, static_cast<float*>(y), M, D, eps, s)
             : tr_layernorm_bf16(x, ldx, static_cast<const uint16_t*>(d), ldd, g, b, static_cast<uint16_t*>(y), M, D, eps, s);
}
// norm over x + pending residual(s): the plain norm1 of a block and the final norm.  d_attn != nullptr: the previous block's norm2 did
// not write the stream back (lazy norm2 below), so BOTH of its residuals are still pending -- (x + d_attn) + d, the reference's order
inline int op_ln_pending(bool f32, float* x, long ldx, const void* d, const void* d_attn, long ldd, const float* g, const float* b, void* y,
                         int M, int D, float eps, tr_stream_t s) {
  if (d_attn == nullptr) return op_ln(f32, x, ldx, d, ldd, g, b, y, M, D, eps, s);
  return tr_layernorm2_bf16(x, ldx, x, ldx, static_cast<const uint16_t*>(d_attn), ldd, static_cast<const uint16_t*>(d), ldd, g, b,
                            static_cast<uint16_t*>(y), M, D, eps, s);
}
inline int op_attn(int prec, const void* qkv, void* out, float* cls_rows, const float* size, float* colsum, int B, int N, int H,
                   tr_stream_t s) {
  if (prec == TR_PREC_BF16X3)
    return tr_attention_split(static_cast<const float*>(qkv), static_cast<float*>(out), cls_rows, size, colsum, B, N, H, s);
  return prec == TR_PREC_FP32 ? tr_attention_f32(static_cast<const float*>(qkv), static_cast<float*>(out), cls_rows, size, colsum, B, N, H, s)
             : tr_attention_bf16(static_cast<const uint16_t*>(qkv), static_cast<uint16_t*>(out), cls_rows, size, colsum, B, N, H, s);
}
inline int op_gather(bool f32, const float* x, const void* d, const int32_t* idx, const int32_t* cidx, const float* scores,
                     const float* g, const float* b, float* x_out, void* y, int B, int N, int K, int D, float eps, tr_stream_t s) {
  return f32 ? tr_gather_layernorm_f32(x, static_cast<const float*>(d), idx, cidx, scores, g, b, x_out, static_cast<float*>(y), B, N, K,
                                       D, eps, s)
             : tr_gather_layernorm_bf16(x, static_cast<const uint16_t*>(d), idx, cidx, scores, g, b, x_out, static_cast<uint16_t*>(y),
                                        B, N, K, D, eps, s);
}

}  // namespace

extern "C" size_t tr_vit_workspace_bytes(const tr_vit_config* cfg, int B) {
  Plan p;
  if (!make_plan(cfg, B, &p)) return 0;
  return p.total;
}

#define TR_TRY(call)            \
  do {                          \
    int rc__ = (call);          \
    if (rc__ != TR_OK) return rc__; \
  } while (0)

// tape != nullptr: TRAINING forward -- every activation the backward pass needs goes to its own slot of the tape (tr_plan.h)
// instead of the shared scratch, the residual stream is written out of place (the inputs of norm1 / norm2 of every block stay),
// fc1 keeps its pre-activation (GELU as a separate kernel), and the decisions (kept ids, sizes) are kept per block.
static int vit_forward_impl(const tr_vit_config* cfg, const tr_vit_weights* w, const float* img, float* logits, void* workspace,
                            size_t workspace_bytes, int32_t* kept_idx, int32_t* compl_idx, float* soft_out,
                            const float* noise_in, float* features_out, int* tokens_out, int B, tr_stream_t s, char* tape,
                            const trplan::TapePlan* tp, const float* drop_scale = nullptr, const uint8_t* drop_keep = nullptr, float drop_rate = 0.f) {
  Plan p;
  const bool train = tape != nullptr;
  TR_REQUIRE(cfg && w && img && logits && workspace, TR_ERR_NULL, "tr_vit_forward: null pointer");
  TR_REQUIRE(make_plan(cfg, B, &p), TR_ERR_CONFIG,
             "tr_vit_forward: invalid config (need embed_dim == 64*heads, dims %% 64 == 0, classes %% 4 == 0, depth <= %d)",
             TR_MAX_DEPTH);
  TR_REQUIRE(workspace_bytes >= p.total, TR_ERR_SHAPE, "tr_vit_forward: workspace too small (%zu < %zu)", workspace_bytes, p.total);
  TR_REQUIRE(tr_aligned16(workspace), TR_ERR_ALIGN, "tr_vit_forward: workspace must be 16-byte aligned");

  tr_prof_restart();
  char* ws = static_cast<char*>(workspace);
  float* x = reinterpret_cast<float*>(ws + p.off_x0);
  float* x_alt = reinterpret_cast<float*>(ws + p.off_x1);
  void* xn = static_cast<void*>(ws + p.off_xn);
  void* const xn_shared = xn;
  void* qkv = static_cast<void*>(ws + p.off_qkv);
  void* ao = static_cast<void*>(ws + p.off_ao);
  void* const ao_shared = ao;
  void* hbuf = static_cast<void*>(ws + p.off_h);
  void* dbuf = static_cast<void*>(ws + p.off_d);   // bf16 output of proj / fc2, added to x by the NEXT norm
  void* const dbuf_shared = dbuf;
  void* const dbuf2 = static_cast<void*>(ws + p.off_d2);
  void* cols = train ? static_cast<void*>(tape + tp->cols) : static_cast<void*>(ws + p.off_cols);
  float* cls_rows = reinterpret_cast<float*>(ws + p.off_cls);
  float* scores = reinterpret_cast<float*>(ws + p.off_scores);
  int32_t* idx_ws = reinterpret_cast<int32_t*>(ws + p.off_idx);
  int32_t* compl_ws = reinterpret_cast<int32_t*>(ws + p.off_compl);
  void* xcls = static_cast<void*>(ws + p.off_xcls);
  float* colsum_part = cfg->family == TR_FAMILY_KMEDOIDS
                           ? reinterpret_cast<float*>(ws + p.off_cluster + align_up(tr_dpcknn_workspace_floats(B, p.N0) * 4))
                           : nullptr;
  float* size_cur = nullptr;                                            // ToMe token sizes: none until the first merge (tome.py:185)
  float* size_a = reinterpret_cast<float*>(ws + p.off_size0);
  float* size_b = reinterpret_cast<float*>(ws + p.off_size1);

  const int D = p.D, H = p.H;
  const int prec = cfg->precision;
  const bool f32 = prec != TR_PREC_BF16;            // fp32 activations (TR_PREC_FP32 and TR_PREC_BF16X3)
  // Lazy norm2 (eval, bf16, families whose blocks all start with a plain norm1): a norm2 that no reduction follows reads x + d_attn but
  // does not store it; the next norm1 (or the final norm) adds d_attn and d_mlp in the reference's order and writes the stream once.
  // Bit-identical to the eager sequence (same fp32 additions), 22 instead of 24 bytes per element and block through the norms.
  static const bool ln_eager = [] { const char* e = getenv("TR_LN_EAGER"); return e && atoi(e) != 0; }();      // lab: A/B switch
  const bool lazy_base = !train && !f32 && !ln_eager && features_out == nullptr;
  // what consumes the pending residuals after block j - 1: a plain norm1 (or the final norm) unless a pre-block reducer fires at block j
  auto starts_plain = [&](int j) {
    if (j >= cfg->depth) return true;
    switch (cfg->family) {
      case TR_FAMILY_DPCKNN: case TR_FAMILY_KMEDOIDS: case TR_FAMILY_PATCHMERGER: case TR_FAMILY_SINKHORN: case TR_FAMILY_DYVIT:
      case TR_FAMILY_SIT: return cfg->keep[j] <= 0;
      default: return true;                                   // in-block families (Top-K, EViT, ToMe, ATS), DeiT, Heuristic (masks only)
    }
  };
  const void* pending_attn = nullptr;      // the attention branch's residual of the previous block, not yet in x (lazy norm2)
  // Fused block tail (tr_mlp_fused_resid_ln_bf16): where the fused eval Mlp runs and the next block starts with a plain norm1, ONE launch does
  // fc1 -> GELU -> fc2, adds the result to the stream in place and writes the next block's norm1 -- into the hidden-activation buffer, which the
  // fused Mlp leaves unused and which nothing touches until that block's own Mlp (its only reader is the next qkv GEMM).  The block's norm2
  // then writes the stream (eager): the kernel's accumulators start at the stream row.  OFF by default (tr_set_mlp_resid_ln): measured in the
  // model it loses 4 % against fused Mlp + LayerNorm launch (the epilogue stalls the workgroup; profiles/r05_mlp_lab.md).
  const bool rl_base = lazy_base && drop_keep == nullptr && drop_scale == nullptr && tr_mlp_resid_ln_enabled();
  const void* xn1_ready = nullptr;         // norm1 of the block about to start, written by the previous block's fused tail
  // a1 + a2: patch embedding, CLS token, position embedding
  static const bool unfused_patch = [] { const char* e = getenv("TR_PATCH_UNFUSED"); return e && atoi(e) != 0; }();   // lab: the three-launch path
  if (!train && !f32 && !unfused_patch && tr_patch_embed_supported(cfg->in_chans, cfg->img_size, cfg->patch, D)) {
    // eval: unfold + GEMM + cls/pos in one launch (tr_patch.hip), at every batch size (the two paths differ in the last bit: an image's
    // tokens must not depend on its batch); training keeps the column matrix (PatchEmbed's weight-gradient operand)
    TR_TRY(tr_patch_embed_bf16(img, static_cast<const uint16_t*>(w->patch_w), w->patch_b, w->cls_token, w->pos_embed, x, B, cfg->in_chans,
                               cfg->img_size, cfg->patch, D, s));
  } else {
    TR_TRY(op_im2col(f32, img, cols, B, cfg->in_chans, cfg->img_size, cfg->img_size, cfg->patch, s));
    TR_TRY(op_gemm(prec, cols, w->patch_w, w->patch_b, x, w->pos_embed, p.P, B * p.P, D, p.kcols, TR_EPI_PATCH_F32, s));
    TR_TRY(tr_cls_pos_rows(w->cls_token, w->pos_embed, x, B, p.N0, D, s));
  }
  // Dropout (timm's drop_rate: pos_drop topk.py:186, proj_drop :53, the Mlp's two nn.Dropout): training only.  The caller draws the keep
  // masks (1 byte per element, in forward order: tr_vit_dropout_mask_bytes); survivors are scaled by 1 / (1 - p) like nn.Dropout.
  const float drop_mul = drop_keep != nullptr ? 1.0f / (1.0f - drop_rate) : 1.0f;
  if (drop_keep != nullptr) {
    TR_TRY(tr_dropout_f32(x, x, drop_keep, drop_mul, (size_t)B * p.N0 * D, s));
    drop_keep += (size_t)B * p.N0 * D;
  }

  // the stream-K hand-over counters of every fused-Mlp launch of this forward (block i: set i) start at zero: ONE memset node here instead
  // of one in front of each launch
  static const bool memset_each = [] { const char* e = getenv("TR_MLP_MEMSET_EACH"); return e && atoi(e) != 0; }();      // lab: A/B switch (a memset node per launch)
  static const bool no_streamk = [] { const char* e = getenv("TR_MLP_NO_STREAMK"); return e && atoi(e) != 0; }();          // lab: whole blocks round-robin, no hand-over
  const int conc = (!train && cfg->concurrent) ? 1 : 0;       // other forwards run beside this one: a launch need not fill the chip on its own
  // the fused Mlp's stream-K scratch is there and wanted.  Beside other forwards it is not: the other forward's launches fill the second
  // round's idle compute units, and whole blocks round-robin move no accumulators (125 MB per launch at the first stage): measured with two
  // forwards in flight +0.5 % (Top-K kr 0.7), +1.5 % (kr 0.5), +2 % (dense DeiT-S); one at a time -1.5 ... -4 % (tools/lab/inflight_ab2.py)
  const bool sk_ok = p.mlp_sk_bytes > 0 && !no_streamk && !conc;
  const bool one_memset = !train && prec == TR_PREC_BF16 && sk_ok && w->blocks[0].mlp_pk != nullptr && !memset_each;
  if (one_memset) TR_TRY(tr_mlp_fused_zero_counters(ws + p.off_mlp_sk, p.mlp_sk_bytes, D, p.Hd, cfg->depth, s));
  int N = p.N0;
  const void* pending = nullptr;   // residual not yet added to x (the previous block's fc2 output)
  const float* policy_cur = nullptr;      // DyViT training: the keep policy every block attends under (all ones before the first stage)
  if (train && cfg->family == TR_FAMILY_DYVIT) {
    float* ones = reinterpret_cast<float*>(tape + tp->ones);
    TR_TRY(tr_fill_f32(ones, 1.0f, (size_t)B * p.N0, s));
    policy_cur = ones;
  }
  for (int i = 0; i < cfg->depth; ++i) {
    const tr_block_weights* bw = &w->blocks[i];
    const bool tome = cfg->family == TR_FAMILY_TOME;
    bool have_xn = false;   // norm1(x) already in xn (written by a pre-block reducer)
    if (cfg->family == TR_FAMILY_DPCKNN && cfg->keep[i] > 0) {
      // a19 + a20: CTM (dpcknn.py:153-172) on x[:, 1:] BEFORE the block; merge fused with the block's norm1
      const tr_stage_weights* sw = &w->stage[i];
      const int Kc = cfg->keep[i], M = B * N;
      TR_REQUIRE(Kc <= N - 1, TR_ERR_CONFIG, "tr_vit_forward: block %d asks for %d clusters of %d patch tokens", i, Kc, N - 1);
      if (train) {          // the stream entering the merge stays on the tape (out of place), norm1's input / output go to their slots
        float* x0 = reinterpret_cast<float*>(tape + tp->blk[i].x0);
        // (the norm output of this pass is not used -- it goes to the shared scratch, NOT to xn, which still names the previous
        // block's norm2 slot on the tape)
        TR_TRY(tr_layernorm_bf16_to(x, D, x0, D, static_cast<const uint16_t*>(pending), D, bw->ln1_g, bw->ln1_b, static_cast<uint16_t*>(xn_shared), M,
                                    D, cfg->ln_eps, s));
        x = x0;
        x_alt = reinterpret_cast<float*>(tape + tp->blk[i].x1);
        xn = tape + tp->blk[i].xn1;
      } else if (pending) TR_TRY(op_ln(f32, x, D, pending, D, bw->ln1_g, bw->ln1_b, xn, M, D, cfg->ln_eps, s));   // x += previous mlp output
      pending = nullptr;
      float* cws = reinterpret_cast<float*>(ws + p.off_cluster);
      float* wtok = train ? reinterpret_cast<float*>(tape + tp->blk[i].scores) : cws + tr_dpcknn_workspace_floats(B, p.N0);
      int32_t* centers = kept_idx ? kept_idx + (size_t)i * B * p.N0 : idx_ws;
      int32_t* assign = compl_idx ? compl_idx + (size_t)i * B * p.N0 : compl_ws;
      if (train) {
        centers = reinterpret_cast<int32_t*>(tape + tp->blk[i].idx);
        assign = reinterpret_cast<int32_t*>(tape + tp->blk[i].idx2);
      }
      TR_TRY(tr_dpcknn_cluster(x, noise_in, cws, centers, assign, scores, B, N, D, Kc, cfg->knn_k > 0 ? cfg->knn_k : 5, f32 ? 0 : 1, s));
      if (noise_in) noise_in += (size_t)B * (N - 1);
      TR_TRY(tr_cluster_merge_layernorm(x, sw->w3, sw->b3, wtok, assign, bw->ln1_g, bw->ln1_b, x_alt, xn, f32 ? 1 : 0, B, N, Kc, D,
                                        cfg->ln_eps, s));
      float* t = x; x = x_alt; x_alt = t;
      N = Kc + 1;
      have_xn = true;
    }
    if (cfg->family == TR_FAMILY_KMEDOIDS && cfg->keep[i] > 0) {
      // a21: KMedoids (kmedoids.py:135-149) on x[:, 1:] BEFORE the block: the medoid tokens replace the patch tokens
      const int Kc = cfg->keep[i], M = B * N;
      TR_REQUIRE(i > 0, TR_ERR_CONFIG, "tr_vit_forward: K-Medoids at block 0 has no previous attention to weigh the tokens "
                                       "(the reference fails there too: `attn` is unbound, kmedoids.py:240)");
      const int kinit = cfg->kmed_init[i];             // > 0: args.equal_weight, first medoid id + 1
      TR_REQUIRE(Kc <= N - 1, TR_ERR_CONFIG, "tr_vit_forward: block %d asks for %d medoids of %d patch tokens", i, Kc, N - 1);
      int32_t* centers = kept_idx ? kept_idx + (size_t)i * B * p.N0 : idx_ws;
      int32_t* assign = compl_idx ? compl_idx + (size_t)i * B * p.N0 : compl_ws;
      if (train) {          // as for DPC-KNN: the stream entering the reduction stays on the tape, the medoid rows become norm1's input slot
        float* x0 = reinterpret_cast<float*>(tape + tp->blk[i].x0);
        TR_TRY(tr_layernorm_bf16_to(x, D, x0, D, static_cast<const uint16_t*>(pending), D, bw->ln1_g, bw->ln1_b, static_cast<uint16_t*>(xn_shared), M,
                                    D, cfg->ln_eps, s));
        x = x0;
        x_alt = reinterpret_cast<float*>(tape + tp->blk[i].x1);
        xn = tape + tp->blk[i].xn1;
        centers = reinterpret_cast<int32_t*>(tape + tp->blk[i].idx);
        assign = reinterpret_cast<int32_t*>(tape + tp->blk[i].idx2);
      } else if (pending) TR_TRY(op_ln(f32, x, D, pending, D, bw->ln1_g, bw->ln1_b, xn, M, D, cfg->ln_eps, s));   // x += previous mlp output
      pending = nullptr;
      if (kinit > 0)
        TR_TRY(tr_kmedoids_equal(x, kinit - 1, reinterpret_cast<float*>(ws + p.off_cluster), centers, assign, B, N, D, Kc, cfg->cluster_iters,
                                 f32 ? 0 : 1, s));
      else
        TR_TRY(tr_kmedoids(x, colsum_part, reinterpret_cast<float*>(ws + p.off_cluster), centers, assign, B, N, D, H, Kc,
                           cfg->cluster_iters, f32 ? 0 : 1, s));
      TR_TRY(op_gather(f32, x, nullptr, centers, nullptr, nullptr, bw->ln1_g, bw->ln1_b, x_alt, xn, B, N, Kc, D, cfg->ln_eps, s));
      float* t = x; x = x_alt; x_alt = t;
      N = Kc + 1;
      have_xn = true;
    }
    if (cfg->family == TR_FAMILY_HEURISTIC && w->stage[i].w3 != nullptr) {
      // f4: a new spatial mask takes effect at this block and stays until the next one (heuristic.py:247-258)
      TR_REQUIRE(w->stage[i].n_pad == N, TR_ERR_CONFIG, "tr_vit_forward: block %d mask has %d entries for %d tokens", i, w->stage[i].n_pad, N);
      float* mask_dst = train ? reinterpret_cast<float*>(tape + tp->blk[i].size) : size_a;      // training keeps every block's mask
      TR_TRY(tr_broadcast_rows(w->stage[i].w3, mask_dst, B, N, s));
      size_cur = mask_dst;
    }
    if (cfg->family == TR_FAMILY_PATCHMERGER && cfg->keep[i] > 0) {
      // f4: PatchMerger.forward patchmerger.py:35-39 on x[:, 1:] BEFORE the block
      const tr_stage_weights* sw = &w->stage[i];
      const int Kc = cfg->keep[i], M = B * N;
      TR_REQUIRE(Kc <= N - 1, TR_ERR_CONFIG, "tr_vit_forward: block %d asks for %d outputs of %d patch tokens", i, Kc, N - 1);
      TR_REQUIRE(sw->ln_g && sw->ln_b && sw->w1 && sw->b1 && sw->n_pad >= Kc && sw->n_pad % 8 == 0 &&
                     sw->n_pad == trplan::soft_ld(Kc),
                 TR_ERR_CONFIG, "tr_vit_forward: block %d PatchMerger weights missing or n_pad=%d invalid for K=%d", i, sw->n_pad, Kc);
      if (train) {
        // every operand of the stage's backward stays on the tape; the merged stream is written to norm1's input slot
        const trplan::BlockTape& bt = tp->blk[i];
        TR_REQUIRE(sw->n_pad == trplan::soft_ld(Kc), TR_ERR_CONFIG, "tr_vit_forward_train: block %d PatchMerger n_pad=%d K=%d", i, sw->n_pad, Kc);
        float* x0 = reinterpret_cast<float*>(tape + bt.x0);
        float* xh = reinterpret_cast<float*>(tape + bt.sxh);
        float* slog = reinterpret_cast<float*>(tape + bt.slog);
        float* swt = reinterpret_cast<float*>(tape + bt.swt);
        uint16_t* pu = reinterpret_cast<uint16_t*>(tape + bt.pu);
        float* x1 = reinterpret_cast<float*>(tape + bt.x1);
        TR_TRY(tr_layernorm_bf16_to(x, D, x0, D, static_cast<const uint16_t*>(pending), D, sw->ln_g, sw->ln_b, pu, M, D, 1e-5f, s));
        pending = nullptr;
        TR_TRY(tr_layernorm_f32(x0, D, nullptr, D, sw->ln_g, sw->ln_b, xh, M, D, 1e-5f, s));
        TR_TRY(tr_gemm_bf16(pu, static_cast<const uint16_t*>(sw->w1), sw->b1, slog, nullptr, 0, M, sw->n_pad, D, TR_EPI_F32, s));
        TR_REQUIRE(hipMemcpyAsync(swt, slog, (size_t)M * sw->n_pad * 4, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(s)) == hipSuccess,
                   TR_ERR_LAUNCH, "tr_vit_forward_train: copy failed");
        TR_TRY(tr_softassign_merge_fast(swt, sw->n_pad, sw->scale, 1, x0, xh, x1, soft_out, B, N, Kc, D, s));
        if (soft_out) soft_out += (size_t)B * Kc * (N - 1);
        x = x1;
        N = Kc + 1;
      } else {
      float* xh = static_cast<float*>(qkv);                       // LayerNorm-ed tokens, fp32 [M, D]
      float* sc = reinterpret_cast<float*>(ws + p.off_soft);      // similarities [M, n_pad]
      TR_TRY(op_ln(f32, x, D, pending, D, sw->ln_g, sw->ln_b, xn, M, D, 1e-5f, s));        // x += previous mlp output; GEMM operand
      pending = nullptr;
      TR_TRY(tr_layernorm_f32(x, D, nullptr, D, sw->ln_g, sw->ln_b, xh, M, D, 1e-5f, s));  // the rows that are summed
      TR_TRY(op_gemm(prec, xn, sw->w1, sw->b1, sc, nullptr, 0, M, sw->n_pad, D, TR_EPI_F32, s));
      if (!f32 && Kc <= 192)
        TR_TRY(tr_softassign_merge_fast(sc, sw->n_pad, sw->scale, 1, x, xh, x_alt, soft_out, B, N, Kc, D, s));
      else
        TR_TRY(tr_softassign_merge(sc, sw->n_pad, sw->scale, x, xh, x_alt, soft_out, B, N, Kc, D, s));
      if (soft_out) soft_out += (size_t)B * Kc * (N - 1);
      float* t = x; x = x_alt; x_alt = t;
      N = Kc + 1;
      }
    }
    if (cfg->family == TR_FAMILY_SINKHORN && cfg->keep[i] > 0) {
      // a22: Sinkhorn.forward sinkhorn.py:66-86 on x[:, 1:] BEFORE the block
      const tr_stage_weights* sw = &w->stage[i];
      const int Kc = cfg->keep[i], M = B * N;
      TR_REQUIRE(Kc <= N - 1, TR_ERR_CONFIG, "tr_vit_forward: block %d asks for %d clusters of %d patch tokens", i, Kc, N - 1);
      TR_REQUIRE(sw->w1 && sw->b1 && sw->n_pad >= Kc && sw->n_pad % 8 == 0 && sw->n_pad == trplan::soft_ld(Kc),
                 TR_ERR_CONFIG, "tr_vit_forward: block %d Sinkhorn centres missing or n_pad=%d invalid for K=%d", i, sw->n_pad, Kc);
      if (train) {
        const trplan::BlockTape& bt = tp->blk[i];
        TR_REQUIRE(sw->n_pad == trplan::soft_ld(Kc), TR_ERR_CONFIG, "tr_vit_forward_train: block %d Sinkhorn n_pad=%d K=%d", i, sw->n_pad, Kc);
        float* x0 = reinterpret_cast<float*>(tape + bt.x0);
        float* xh = reinterpret_cast<float*>(tape + bt.sxh);
        float* slog = reinterpret_cast<float*>(tape + bt.slog);
        float* swt = reinterpret_cast<float*>(tape + bt.swt);
        uint16_t* pu = reinterpret_cast<uint16_t*>(tape + bt.pu);
        float* x1 = reinterpret_cast<float*>(tape + bt.x1);
        // x0 = x + pending (the norm output of this pass is not used: shared scratch)
        TR_TRY(tr_layernorm_bf16_to(x, D, x0, D, static_cast<const uint16_t*>(pending), D, bw->ln1_g, bw->ln1_b, static_cast<uint16_t*>(xn_shared), M, D,
                                    cfg->ln_eps, s));
        pending = nullptr;
        TR_TRY(tr_rownorm(x0, xh, pu, 0, M, D, s));
        TR_TRY(tr_gemm_bf16(pu, static_cast<const uint16_t*>(sw->w1), sw->b1, slog, nullptr, 0, M, sw->n_pad, D, TR_EPI_F32, s));
        TR_TRY(tr_sinkhorn(slog, sw->n_pad, cfg->sinkhorn_eps > 0.f ? cfg->sinkhorn_eps : 1.0f, cfg->cluster_iters, swt, soft_out, B, N, Kc, s));
        if (soft_out) soft_out += (size_t)B * Kc * (N - 1);
        TR_TRY(tr_softassign_merge_fast(swt, sw->n_pad, 1.0f, 0, x0, xh, x1, nullptr, B, N, Kc, D, s));
        x = x1;
        N = Kc + 1;
      } else {
      if (pending) TR_TRY(op_ln(f32, x, D, pending, D, bw->ln1_g, bw->ln1_b, xn, M, D, cfg->ln_eps, s));   // x += previous mlp output
      pending = nullptr;
      float* xh = static_cast<float*>(qkv);                       // unit-norm tokens, fp32 [M, D] (the qkv slab is free here)
      float* sc = reinterpret_cast<float*>(ws + p.off_soft);      // scores, then the transport plan in place [M, n_pad]
      TR_TRY(tr_rownorm(x, xh, xn, f32 ? 1 : 0, M, D, s));
      TR_TRY(op_gemm(prec, xn, sw->w1, sw->b1, sc, nullptr, 0, M, sw->n_pad, D, TR_EPI_F32, s));
      TR_TRY(tr_sinkhorn(sc, sw->n_pad, cfg->sinkhorn_eps > 0.f ? cfg->sinkhorn_eps : 1.0f, cfg->cluster_iters, sc, soft_out, B, N,
                         Kc, s));
      if (soft_out) soft_out += (size_t)B * Kc * (N - 1);
      if (!f32 && Kc <= 192)
        TR_TRY(tr_softassign_merge_fast(sc, sw->n_pad, 1.0f, 0, x, xh, x_alt, nullptr, B, N, Kc, D, s));
      else
        TR_TRY(tr_weighted_merge(sc, sw->n_pad, x, xh, x_alt, B, N, Kc, D, s));
      float* t = x; x = x_alt; x_alt = t;
      N = Kc + 1;
      }
    }
    if (train && cfg->family == TR_FAMILY_DYVIT && cfg->keep[i] > 0) {
      // a11 / f4: DyViT TRAINING (dyvit.py:221-229): PredictorLG on the patch tokens under the previous decision, a straight-through
      // Gumbel-softmax sample becomes this stage's policy; no token is removed.  Every activation goes to the stage's tape slots.
      const tr_stage_weights* sw = &w->stage[i];
      const trplan::BlockTape& bt = tp->blk[i];
      const int M = B * N, Hh = sw->h_pad > 0 ? sw->h_pad : D / 2, Q = (D / 4 + 63) / 64 * 64;
      TR_REQUIRE(sw->ln_g && sw->ln_b && sw->w0 && sw->b0 && sw->w1 && sw->b1 && sw->w2 && sw->b2 && sw->w3 && sw->b3, TR_ERR_NULL,
                 "tr_vit_forward_train: block %d predictor weights missing", i);
      // Hh: the D/2 hidden layer as packed -- zero-padded to a multiple of 64 (DeiT-T: 96 -> 128; zero weight rows / columns and zero
      // bias: the padded activations are gelu(0) = 0 and carry no gradient), like the D/4 layer is padded to Q rows
      TR_REQUIRE(Hh >= D / 2 && Hh % 64 == 0 && Hh == (D / 2 + 63) / 64 * 64 && sw->reserved_ == Q, TR_ERR_CONFIG,
                 "tr_vit_forward_train: the DyViT predictor must be packed with its hidden layers padded to %d / %d columns (got h_pad=%d, %d)",
                 (D / 2 + 63) / 64 * 64, Q, sw->h_pad, sw->reserved_);
      TR_REQUIRE(noise_in != nullptr, TR_ERR_NULL, "tr_vit_forward_train: DyViT needs the Gumbel noise of every stage (noise_in)");
      float* x0 = reinterpret_cast<float*>(tape + bt.x0);
      uint16_t* pu = reinterpret_cast<uint16_t*>(tape + bt.pu);
      uint16_t* ppre0 = reinterpret_cast<uint16_t*>(tape + bt.ppre0);
      uint16_t* pcat = reinterpret_cast<uint16_t*>(tape + bt.pcat);
      uint16_t* ppre1 = reinterpret_cast<uint16_t*>(tape + bt.ppre1);
      uint16_t* ph1 = reinterpret_cast<uint16_t*>(tape + bt.ph1);
      uint16_t* ppre2 = reinterpret_cast<uint16_t*>(tape + bt.ppre2);
      uint16_t* ph2 = reinterpret_cast<uint16_t*>(tape + bt.ph2);
      float* pol = reinterpret_cast<float*>(tape + bt.pol);
      TR_TRY(tr_layernorm_bf16_to(x, D, x0, D, static_cast<const uint16_t*>(pending), D, sw->ln_g, sw->ln_b, pu, M, D, 1e-5f, s));
      pending = nullptr;
      x = x0;
      TR_TRY(tr_gemm_gelu_keep_bf16(pu, static_cast<const uint16_t*>(sw->w0), sw->b0, ppre0, pcat, M, D, D, s));
      TR_TRY(tr_pool_policy(pcat, policy_cur, B, N, D, 1e-6f, s));
      TR_TRY(tr_gemm_gelu_keep_bf16(pcat, static_cast<const uint16_t*>(sw->w1), sw->b1, ppre1, ph1, M, Hh, D, s));
      TR_TRY(tr_gemm_gelu_keep_bf16(ph1, static_cast<const uint16_t*>(sw->w2), sw->b2, ppre2, ph2, M, Q, Hh, s));
      TR_TRY(tr_dyvit_decide(ph2, Q, sw->w3, sw->b3, noise_in, policy_cur, pol, reinterpret_cast<float*>(tape + bt.ysoft),
                             reinterpret_cast<float*>(tape + bt.sm), reinterpret_cast<float*>(tape + bt.hard), B, N, D / 4, s));
      noise_in += (size_t)B * (N - 1) * 2;
      policy_cur = pol;
    } else if (train && cfg->family == TR_FAMILY_SIT && cfg->keep[i] > 0) {
      // a23 TRAINING: TokenSlimmingModule (sit.py:36-40) with every activation on the tape
      const tr_stage_weights* sw = &w->stage[i];
      const trplan::BlockTape& bt = tp->blk[i];
      const int Kc = cfg->keep[i], M = B * N, Hh = (D / 2 + 63) / 64 * 64;     // hidden width as packed (DeiT-T: 96 -> 128, zero padded)
      TR_REQUIRE(Kc <= N - 1, TR_ERR_CONFIG, "tr_vit_forward_train: block %d asks for %d of %d patch tokens", i, Kc, N - 1);
      TR_REQUIRE(sw->ln_g && sw->ln_b && sw->w0 && sw->b0 && sw->w1 && sw->b1, TR_ERR_NULL, "tr_vit_forward_train: block %d SiT weights missing", i);
      TR_REQUIRE(sw->n_pad == trplan::soft_ld(Kc) && (sw->h_pad == Hh || (sw->h_pad == 0 && Hh == D / 2)), TR_ERR_CONFIG,
                 "tr_vit_forward_train: the SiT module must be packed with its hidden layer padded to %d columns and n_pad == %d (got h_pad=%d n_pad=%d)",
                 Hh, trplan::soft_ld(Kc), sw->h_pad, sw->n_pad);
      float* x0 = reinterpret_cast<float*>(tape + bt.x0);
      float* slog = reinterpret_cast<float*>(tape + bt.slog);
      float* swt = reinterpret_cast<float*>(tape + bt.swt);
      uint16_t* pu = reinterpret_cast<uint16_t*>(tape + bt.pu);
      uint16_t* ppre0 = reinterpret_cast<uint16_t*>(tape + bt.ppre0);
      uint16_t* ph0 = reinterpret_cast<uint16_t*>(tape + bt.pcat);
      float* x1 = reinterpret_cast<float*>(tape + bt.x1);
      TR_TRY(tr_layernorm_bf16_to(x, D, x0, D, static_cast<const uint16_t*>(pending), D, sw->ln_g, sw->ln_b, pu, M, D, 1e-5f, s));
      pending = nullptr;
      TR_TRY(tr_gemm_gelu_keep_bf16(pu, static_cast<const uint16_t*>(sw->w0), sw->b0, ppre0, ph0, M, Hh, D, s));
      TR_TRY(tr_gemm_bf16(ph0, static_cast<const uint16_t*>(sw->w1), sw->b1, slog, nullptr, 0, M, sw->n_pad, Hh, TR_EPI_F32, s));
      TR_REQUIRE(hipMemcpyAsync(swt, slog, (size_t)M * sw->n_pad * 4, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(s)) == hipSuccess,
                 TR_ERR_LAUNCH, "tr_vit_forward_train: copy failed");
      TR_TRY(tr_softassign_merge_fast(swt, sw->n_pad, sw->scale, 1, x0, x0, x1, soft_out, B, N, Kc, D, s));
      if (soft_out) soft_out += (size_t)B * Kc * (N - 1);
      x = x1;
      N = Kc + 1;
    } else if ((cfg->family == TR_FAMILY_DYVIT || cfg->family == TR_FAMILY_SIT) && cfg->keep[i] > 0) {
      const tr_stage_weights* sw = &w->stage[i];
      const int Kc = cfg->keep[i], M = B * N;
      TR_REQUIRE(Kc <= N - 1, TR_ERR_CONFIG, "tr_vit_forward: block %d asks for %d of %d patch tokens", i, Kc, N - 1);
      TR_REQUIRE(sw->ln_g && sw->ln_b && sw->w0 && sw->b0 && sw->w1 && sw->b1, TR_ERR_NULL,
                 "tr_vit_forward: block %d has no reduction-module weights (tr_vit_weights.stage)", i);
      // x (+= previous mlp output), module's own LayerNorm (nn.LayerNorm default eps 1e-5: dyvit.py:97, sit.py:30)
      TR_TRY(op_ln(f32, x, D, pending, D, sw->ln_g, sw->ln_b, xn, M, D, 1e-5f, s));
      pending = nullptr;
      if (cfg->family == TR_FAMILY_DYVIT) {
        // a10: PredictorLG (dyvit.py:113-119, policy == 1 in eval) -> score -> argsort(desc)[:K] -> batch_index_select
        TR_REQUIRE(sw->w2 && sw->b2 && sw->w3 && sw->b3, TR_ERR_NULL, "tr_vit_forward: block %d predictor weights missing", i);
        const int Hh = sw->h_pad > 0 ? sw->h_pad : D / 2;            // hidden width as packed (zero-padded to 64 for DeiT-T)
        TR_REQUIRE(Hh >= D / 2 && (Hh % 64 == 0 || f32), TR_ERR_CONFIG, "tr_vit_forward: DyViT predictor hidden width %d invalid (D=%d)", Hh, D);
        TR_TRY(op_gemm(prec, xn, sw->w0, sw->b0, ao, nullptr, 0, M, D, D, TR_EPI_GELU_BF16, s));
        TR_TRY(tr_pool_broadcast(ao, f32 ? 1 : 0, B, N, D, 1e-6f, s));
        TR_TRY(op_gemm(prec, ao, sw->w1, sw->b1, qkv, nullptr, 0, M, Hh, D, TR_EPI_GELU_BF16, s));
        TR_TRY(op_gemm(prec, qkv, sw->w2, sw->b2, hbuf, nullptr, 0, M, D / 4, Hh, TR_EPI_GELU_BF16, s));
        TR_TRY(tr_dyvit_score(hbuf, f32 ? 1 : 0, sw->w3, sw->b3, cls_rows, M, D / 4, s));
        int32_t* idx_dst = kept_idx ? kept_idx + (size_t)i * B * p.N0 : idx_ws;
        TR_TRY(tr_cls_topk(cls_rows, idx_dst, nullptr, scores, B, 1, N, Kc, s));
        TR_TRY(op_gather(f32, x, nullptr, idx_dst, nullptr, nullptr, bw->ln1_g, bw->ln1_b, x_alt, xn, B, N, Kc, D, cfg->ln_eps, s));
        have_xn = true;
      } else {
        // a23: TokenSlimmingModule (sit.py:36-40)
        TR_REQUIRE(sw->n_pad >= Kc && sw->n_pad % 8 == 0 && sw->n_pad == trplan::soft_ld(Kc), TR_ERR_CONFIG,
                   "tr_vit_forward: block %d SiT n_pad=%d invalid for K=%d", i, sw->n_pad, Kc);
        const int Hh = sw->h_pad > 0 ? sw->h_pad : D / 2;
        TR_REQUIRE(Hh >= D / 2 && (Hh % 64 == 0 || f32), TR_ERR_CONFIG, "tr_vit_forward: SiT hidden width %d invalid (D=%d)", Hh, D);
        TR_TRY(op_gemm(prec, xn, sw->w0, sw->b0, ao, nullptr, 0, M, Hh, D, TR_EPI_GELU_BF16, s));
        float* sc = reinterpret_cast<float*>(ws + p.off_soft);    // logits [M, n_pad]
        TR_TRY(op_gemm(prec, ao, sw->w1, sw->b1, sc, nullptr, 0, M, sw->n_pad, Hh, TR_EPI_F32, s));
        if (!f32 && Kc <= 192)
          TR_TRY(tr_softassign_merge_fast(sc, sw->n_pad, sw->scale, 1, x, x, x_alt, soft_out, B, N, Kc, D, s));
        else
          TR_TRY(tr_sit_merge(sc, sw->n_pad, sw->scale, x, x_alt, soft_out, B, N, Kc, D, s));
        if (soft_out) soft_out += (size_t)B * Kc * (N - 1);
      }
      float* t = x; x = x_alt; x_alt = t;
      N = Kc + 1;
    }
    const bool ats = cfg->family == TR_FAMILY_ATS;
    const int Ks = ats ? cfg->keep[i] : 0;      // ATS sample_count of this block (0 = plain block)
    const bool in_block = cfg->family == TR_FAMILY_TOPK || cfg->family == TR_FAMILY_EVIT;
    int K = in_block ? cfg->keep[i] : 0;
    TR_REQUIRE(K >= 0 && K <= N - 1, TR_ERR_CONFIG, "tr_vit_forward: block %d keeps %d of %d patch tokens", i, K, N - 1);
    if (K == N - 1) K = 0;  // topk.py:57 / evit.py:79: left_tokens == N-1 -> plain block
    int r = 0;              // ToMe: tokens merged away by this block, r = min(r, (N - protected) // 2)  (tome.py:253)
    if (tome) {
      TR_REQUIRE(cfg->keep[i] >= 0, TR_ERR_CONFIG, "tr_vit_forward: block %d has negative ToMe r", i);
      r = cfg->keep[i] < (N - 1) / 2 ? cfg->keep[i] : (N - 1) / 2;
    }
    const int M = B * N;
    if (train) {          // this block's tape slots replace the shared scratch
      const trplan::BlockTape& bt = tp->blk[i];
      xn = tape + bt.xn1; qkv = tape + bt.qkv; hbuf = tape + bt.h;
      ao = (ats && Ks > 0) ? ao_shared : static_cast<void*>(tape + bt.ao);      // ATS keeps only the sampled rows of attn @ v
      dbuf = (cfg->family == TR_FAMILY_EVIT && K > 0) ? static_cast<void*>(tape + bt.dattn) : dbuf_shared;
      x_alt = reinterpret_cast<float*>(tape + bt.x2);
    }
    // x (+= previous mlp output); attn(norm1(x)) -> dbuf   [x + dbuf is the reference's post-attention x, topk.py:87]
    if (train && !have_xn) {
      float* x1 = reinterpret_cast<float*>(tape + tp->blk[i].x1);
      TR_TRY(tr_layernorm_bf16_to(x, D, x1, D, static_cast<const uint16_t*>(pending), D, bw->ln1_g, bw->ln1_b, static_cast<uint16_t*>(xn), M, D,
                                  cfg->ln_eps, s));
      x = x1;
    } else if (!have_xn && xn1_ready == nullptr) {
      TR_TRY(op_ln_pending(f32, x, D, pending, pending_attn, D, bw->ln1_g, bw->ln1_b, xn, M, D, cfg->ln_eps, s));
      pending_attn = nullptr;
    }
    TR_REQUIRE(pending_attn == nullptr, TR_ERR_CONFIG, "tr_vit_forward: internal: block %d did not absorb the lazy residual", i);
    TR_TRY(op_gemm(prec, xn1_ready ? xn1_ready : xn, bw->qkv_w, bw->qkv_b, qkv, nullptr, 0, M, 3 * D, D, TR_EPI_BF16, s));
    xn1_ready = nullptr;
    // ToMe: log(size) bias on the keys; ATS: key mask as a 1/0 "size" (log 0 = -inf -> exactly zero weight, like
    // masked_fill(-finfo.max) underflowing in the reference's softmax, ats.py:117-120)
    // K-Medoids: the NEXT block's clustering is seeded by the column sums of THIS block's attention (kmedoids.py:240)
    const bool want_colsum = cfg->family == TR_FAMILY_KMEDOIDS && i + 1 < cfg->depth && cfg->keep[i + 1] > 0;
    const bool masked = ats || cfg->family == TR_FAMILY_HEURISTIC;
    if (policy_cur != nullptr)       // DyViT training: softmax_with_policy in every block (dyvit.py:245-246)
      TR_TRY(tr_attention_policy_bf16(static_cast<const uint16_t*>(qkv), static_cast<uint16_t*>(ao), policy_cur, B, N, H, s));
    else
      TR_TRY(op_attn(prec, qkv, ao, (K > 0 || Ks > 0) ? cls_rows : nullptr, (tome || masked) ? size_cur : nullptr,
                     want_colsum ? colsum_part : nullptr, B, N, H, s));
    int Nn = N;
    bool norm2_in_mlp = false;      // this block's norm2 runs inside its fused Mlp launch
    if (Ks > 0) {
      // a16-a18: sample token ids on the CLS attention x |v|, keep those rows of x and of attn @ v
      const tr_stage_weights* sw = &w->stage[i];
      const bool dyn = cfg->ats_dynamic != 0 && !train;
      // (dynamic width: N is the batch maximum of the previous stage and may be below the static Ks; the buffers -- ids, the mask in the
      // score buffer -- are sized for the first stage's N0 rows, which bounds Ks in either mode)
      TR_REQUIRE(Ks >= 2 && (dyn ? Ks <= p.N0 : Ks <= N), TR_ERR_CONFIG, "tr_vit_forward: block %d ATS sample_count %d out of range for %d tokens", i, Ks,
                 dyn ? p.N0 : N);
      TR_REQUIRE(sw->w3 && sw->n_pad >= 1, TR_ERR_NULL, "tr_vit_forward: block %d has no ATS sample grid (tr_vit_weights.stage)", i);
      int32_t* ids = kept_idx ? kept_idx + (size_t)i * B * p.N0 : idx_ws;
      float* mask_next = (size_cur == size_a) ? size_b : size_a;
      if (train) {
        ids = reinterpret_cast<int32_t*>(tape + tp->blk[i].idx);
        mask_next = reinterpret_cast<float*>(tape + tp->blk[i].size);
      }
      int Kg = Ks;             // rows the block keeps
      if (dyn) {
        // ats.py:77-78: the reference pads the unique ids to the batch maximum -- sample at the static bound (ids [B,Ks] stay the
        // Kept_Tokens record), read that maximum back (one int: the stream is synchronised HERE), continue on the first Kg columns
        float* mask_full = scores;                                   // [B,Ks]; the score buffer is not used by this family
        int32_t* width_dev = reinterpret_cast<int32_t*>(ws + p.off_misc);
        TR_TRY(tr_ats_sample(cls_rows, qkv, f32 ? 1 : 0, size_cur, sw->w3, sw->n_pad, ids, mask_full, nullptr, B, N, H, Ks, s));
        TR_TRY(tr_ats_width(mask_full, width_dev, B, Ks, s));
        int32_t width = 0;
        hipError_t e = hipMemcpyAsync(&width, width_dev, sizeof(width), hipMemcpyDeviceToHost, static_cast<hipStream_t>(s));
        if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(s));
        TR_REQUIRE(e == hipSuccess, TR_ERR_LAUNCH, "tr_vit_forward: ATS dynamic width read-back at block %d: %s (not capturable in a hipGraph)", i,
                   hipGetErrorString(e));
        Kg = width < 2 ? 2 : (width > Ks ? Ks : width);              // CLS + at least one column (an all-masked batch cannot occur: >= 1 sample)
        TR_TRY(tr_ats_narrow(ids, mask_full, compl_ws, mask_next, B, Ks, Kg, s));
        ids = compl_ws;
      } else {
        TR_TRY(tr_ats_sample(cls_rows, qkv, f32 ? 1 : 0, size_cur, sw->w3, sw->n_pad, ids, mask_next, nullptr, B, N, H, Ks, s));
      }
      if (train) {          // sampled rows of the stream -> x0 slot, of attn @ v -> ao slot (proj's operand); norm1's input stays in x1
        float* xg = reinterpret_cast<float*>(tape + tp->blk[i].x0);
        xn = tape + tp->blk[i].ao;
        TR_TRY(tr_ats_gather(x, ao, 0, ids, xg, xn, B, N, Ks, D, s));
        x = xg;
      } else {
        TR_TRY(tr_ats_gather(x, ao, f32 ? 1 : 0, ids, x_alt, xn, B, N, Kg, D, s));
        float* t = x; x = x_alt; x_alt = t;
      }
      size_cur = mask_next;
      Nn = Kg;
      TR_TRY(op_gemm(prec, xn, bw->proj_w, bw->proj_b, dbuf, nullptr, 0, B * Nn, D, D, TR_EPI_BF16, s));
    } else {
      TR_TRY(op_gemm(prec, ao, bw->proj_w, bw->proj_b, dbuf, nullptr, 0, M, D, D, TR_EPI_BF16, s));
    }
    if (drop_keep != nullptr) {      // proj_drop (topk.py:53), inside the attention module: before the branch's DropPath
      TR_TRY(tr_dropout_bf16(static_cast<const uint16_t*>(dbuf), static_cast<uint16_t*>(dbuf), drop_keep, drop_mul, (size_t)B * Nn * D, s));
      drop_keep += (size_t)B * Nn * D;
    }
    if (drop_scale != nullptr)      // DropPath on the attention branch (topk.py:87): this block's per-image scale, first of its two draws
      TR_TRY(tr_rowscale_bf16(static_cast<const uint16_t*>(dbuf), static_cast<uint16_t*>(dbuf), drop_scale + (size_t)(2 * i) * B, B, Nn, D, s));
    if (K > 0) {
      // Top-K on the CLS attention, then residual add + gather/compact (+ EViT fused token) + norm2 in one pass
      const bool fuse = cfg->family == TR_FAMILY_EVIT;
      int32_t* idx_dst = kept_idx ? kept_idx + (size_t)i * B * p.N0 : idx_ws;
      int32_t* compl_dst = fuse ? (compl_idx ? compl_idx + (size_t)i * B * p.N0 : compl_ws) : nullptr;
      float* sc_dst = scores;
      if (train) {
        idx_dst = reinterpret_cast<int32_t*>(tape + tp->blk[i].idx);
        if (fuse) compl_dst = reinterpret_cast<int32_t*>(tape + tp->blk[i].idx2);
        sc_dst = reinterpret_cast<float*>(tape + tp->blk[i].scores);
        xn = tape + tp->blk[i].xn2;
      }
      TR_TRY(tr_cls_topk(cls_rows, idx_dst, compl_dst, sc_dst, B, H, N, K, s));
      TR_TRY(op_gather(f32, x, dbuf, idx_dst, compl_dst, sc_dst, bw->ln2_g, bw->ln2_b, x_alt, xn, B, N, K, D, cfg->ln_eps, s));
      float* t = x; x = x_alt; x_alt = t;
      Nn = K + 1 + (fuse ? 1 : 0);
    } else if (r > 0) {
      // ToMe: bipartite matching on mean-over-heads K, then residual add + size-weighted merge + norm2 in one pass
      const int na = (N + 1) / 2;
      int32_t* slab = kept_idx ? kept_idx + (size_t)i * B * p.N0 : idx_ws;
      if (train) {
        slab = reinterpret_cast<int32_t*>(tape + tp->blk[i].idx);
        xn = tape + tp->blk[i].xn2;
      }
      int32_t* unm = slab;
      int32_t* src = slab + (size_t)B * (na - r);
      int32_t* dst = src + (size_t)B * r;
      float* size_next = train ? reinterpret_cast<float*>(tape + tp->blk[i].size) : ((size_cur == size_a) ? size_b : size_a);
      TR_TRY(tr_tome_match(qkv, f32 ? 1 : 0, unm, src, dst, B, N, H, r, s));
      TR_TRY(tr_tome_merge_layernorm(x, dbuf, f32 ? 1 : 0, size_cur, unm, src, dst, bw->ln2_g, bw->ln2_b, x_alt, size_next, xn, B, N, r,
                                     D, cfg->ln_eps, s));
      float* t = x; x = x_alt; x_alt = t;
      size_cur = size_next;
      Nn = N - r;
    } else if (train) {
      xn = tape + tp->blk[i].xn2;
      TR_TRY(tr_layernorm_bf16_to(x, D, x_alt, D, static_cast<const uint16_t*>(dbuf), D, bw->ln2_g, bw->ln2_b, static_cast<uint16_t*>(xn), B * Nn, D,
                                  cfg->ln_eps, s));
      x = x_alt;
    } else if (lazy_base && starts_plain(i + 1) && !(rl_base && i + 1 < cfg->depth && bw->mlp_pk != nullptr &&
                                                     tr_mlp_fused_wanted(B * Nn, D, p.Hd, sk_ok, conc))) {
      // lazy norm2.  Where the fused Mlp follows as ONE round of blocks, the norm moves INTO that launch (tr_mlp_fused_ln_bf16: its fc1 waves
      // normalise x + dbuf in registers; bit-identical to the launch below followed by the plain fused Mlp) -- no LayerNorm launch, no bf16
      // rows in between (tr_set_mlp_ln; under the stream-K schedule the separate launch is faster: tr_mlp_fused.hip)
      norm2_in_mlp = prec == TR_PREC_BF16 && drop_keep == nullptr && bw->mlp_pk != nullptr &&
                     tr_mlp_ln_wanted(B * Nn, D, p.Hd, sk_ok, conc);
      if (!norm2_in_mlp)
        TR_TRY(tr_layernorm2_bf16(x, D, nullptr, 0, static_cast<const uint16_t*>(dbuf), D, nullptr, 0, bw->ln2_g, bw->ln2_b,
                                  static_cast<uint16_t*>(xn), B * Nn, D, cfg->ln_eps, s));
      pending_attn = dbuf;
    } else {
      TR_TRY(op_ln(f32, x, D, dbuf, D, bw->ln2_g, bw->ln2_b, xn, B * Nn, D, cfg->ln_eps, s));
    }
    N = Nn;
    const int M2 = B * N;
    // mlp(norm2(x)) -> dbuf, added to x by the next block's norm1 (or the final norm)
    bool fused_mlp = false;
    if (train) {
      void* pre = tape + tp->blk[i].pre;
      TR_TRY(tr_gemm_gelu_keep_bf16(static_cast<const uint16_t*>(xn), static_cast<const uint16_t*>(bw->fc1_w), bw->fc1_b, static_cast<uint16_t*>(pre),
                                    static_cast<uint16_t*>(hbuf), M2, p.Hd, D, s));
    } else if (prec == TR_PREC_BF16 && bw->mlp_pk != nullptr && tr_mlp_fused_wanted(M2, D, p.Hd, sk_ok, conc)) {
      // eval: fc1 -> GELU -> fc2 in one launch, the hidden activation never leaves the CU (tr_mlp_fused.hip; bit-identical to the pair below,
      // taken where its block schedule fills the chip)
      fused_mlp = true;
    } else {
      TR_REQUIRE(!norm2_in_mlp, TR_ERR_CONFIG, "tr_vit_forward: internal: block %d skipped its norm2 launch but does not run the fused Mlp", i);
      TR_TRY(op_gemm(prec, xn, bw->fc1_w, bw->fc1_b, hbuf, nullptr, 0, M2, p.Hd, D, TR_EPI_GELU_BF16, s));
    }
    if (drop_keep != nullptr) {      // timm Mlp: drop after the activation ...
      TR_TRY(tr_dropout_bf16(static_cast<const uint16_t*>(hbuf), static_cast<uint16_t*>(hbuf), drop_keep, drop_mul, (size_t)M2 * p.Hd, s));
      drop_keep += (size_t)M2 * p.Hd;
    }
    dbuf = (pending_attn == dbuf_shared) ? dbuf2 : dbuf_shared;      // the attention residual is still pending: fc2 writes beside it
    const bool fused_tail = fused_mlp && rl_base && pending_attn == nullptr && i + 1 < cfg->depth && starts_plain(i + 1);
    if (fused_tail) {
      const tr_block_weights* nb = &w->blocks[i + 1];
      TR_TRY(tr_mlp_fused_resid_ln_bf16_set(static_cast<const uint16_t*>(xn), bw->mlp_pk, bw->fc1_b, bw->fc2_b, x, nb->ln1_g, nb->ln1_b, cfg->ln_eps,
                                            static_cast<uint16_t*>(hbuf), sk_ok ? ws + p.off_mlp_sk : nullptr, p.mlp_sk_bytes, M2, D, p.Hd,
                                            one_memset ? i : -1, s));
      xn1_ready = hbuf;
    } else if (fused_mlp && norm2_in_mlp)
      TR_TRY(tr_mlp_fused_ln_bf16_set(x, static_cast<const uint16_t*>(pending_attn), bw->ln2_g, bw->ln2_b, cfg->ln_eps, bw->mlp_pk, bw->fc1_b,
                                      static_cast<uint16_t*>(dbuf), sk_ok ? ws + p.off_mlp_sk : nullptr, p.mlp_sk_bytes, M2, D, p.Hd, one_memset ? i : -1, s));
    else if (fused_mlp)
      TR_TRY(tr_mlp_fused_bf16_set(static_cast<const uint16_t*>(xn), bw->mlp_pk, bw->fc1_b, static_cast<uint16_t*>(dbuf),
                                   sk_ok ? ws + p.off_mlp_sk : nullptr, p.mlp_sk_bytes, M2, D, p.Hd, one_memset ? i : -1, s));
    else
      TR_TRY(op_gemm(prec, hbuf, bw->fc2_w, bw->fc2_b, dbuf, nullptr, 0, M2, D, p.Hd, TR_EPI_BF16, s));
    if (drop_keep != nullptr) {      // ... and after fc2
      TR_TRY(tr_dropout_bf16(static_cast<const uint16_t*>(dbuf), static_cast<uint16_t*>(dbuf), drop_keep, drop_mul, (size_t)M2 * D, s));
      drop_keep += (size_t)M2 * D;
    }
    if (drop_scale != nullptr)      // DropPath on the MLP branch (topk.py:95)
      TR_TRY(tr_rowscale_bf16(static_cast<const uint16_t*>(dbuf), static_cast<uint16_t*>(dbuf), drop_scale + (size_t)(2 * i + 1) * B, B, N, D, s));
    pending = fused_tail ? nullptr : dbuf;
    if (features_out && !train) {      // viz_data["Features"][i] (topk.py:197): x + mlp output, which x itself only absorbs in the next norm
      TR_TRY(tr_residual_snapshot(x, pending, f32 ? 1 : 0, features_out, (size_t)M2 * D, s));
      features_out += (size_t)M2 * D;
    }
    if (tokens_out) tokens_out[i] = N;
  }
  // a5: (x += last mlp output and) norm on the CLS rows only (LayerNorm is per-row), then the classifier
  if (train && features_out != nullptr) {
    // DyViT distillation (dyvit.py:252-258): the final norm of EVERY row is an output; the whole stream stays for its backward
    float* xfa = reinterpret_cast<float*>(tape + tp->xfin_all);
    TR_TRY(tr_layernorm_bf16_to(x, D, xfa, D, static_cast<const uint16_t*>(pending), D, w->norm_g, w->norm_b, static_cast<uint16_t*>(xn_shared),
                                B * N, D, cfg->ln_eps, s));
    TR_TRY(tr_layernorm_f32(xfa, D, nullptr, D, w->norm_g, w->norm_b, features_out, B * N, D, cfg->ln_eps, s));
    x = xfa;
    pending = nullptr;
  }
  if (train) {
    xcls = tape + tp->xcls;
    TR_TRY(tr_layernorm_bf16_to(x, (long)N * D, reinterpret_cast<float*>(tape + tp->xfinal), D, static_cast<const uint16_t*>(pending), (long)N * D,
                                w->norm_g, w->norm_b, static_cast<uint16_t*>(xcls), B, D, cfg->ln_eps, s));
  } else {
    TR_TRY(op_ln_pending(f32, x, (long)N * D, pending, pending_attn, (long)N * D, w->norm_g, w->norm_b, xcls, B, D, cfg->ln_eps, s));
  }
  TR_TRY(op_gemm(prec, xcls, w->head_w, w->head_b, logits, nullptr, 0, B, p.C, D, TR_EPI_F32, s));
  return TR_OK;
}

extern "C" int tr_vit_forward_status(const tr_vit_config* cfg, void* workspace, size_t workspace_bytes, int B, tr_stream_t s) {
  Plan p;
  TR_REQUIRE(make_plan(cfg, B, &p), TR_ERR_CONFIG, "tr_vit_forward_status: invalid config");
  TR_REQUIRE(workspace != nullptr, TR_ERR_NULL, "tr_vit_forward_status: null workspace");
  TR_REQUIRE(workspace_bytes >= p.total, TR_ERR_SHAPE, "tr_vit_forward_status: workspace of %zu bytes, tr_vit_workspace_bytes says %zu",
             workspace_bytes, p.total);
  if (p.mlp_sk_bytes == 0) {      // nothing on this executor keeps a device-side record: the stream's own state is the status
    hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(s));
    TR_REQUIRE(e == hipSuccess, TR_ERR_LAUNCH, "tr_vit_forward_status: %s", hipGetErrorString(e));
    return TR_OK;
  }
  return tr_mlp_fused_status(static_cast<unsigned char*>(workspace) + p.off_mlp_sk, p.mlp_sk_bytes, p.D, p.Hd, s);
}

extern "C" int tr_vit_forward(const tr_vit_config* cfg, const tr_vit_weights* w, const float* img, float* logits, void* workspace,
                              size_t workspace_bytes, int32_t* kept_idx, int32_t* compl_idx, float* soft_out,
                              const float* noise_in, float* features_out, int* tokens_out, int B, tr_stream_t s) {
  return vit_forward_impl(cfg, w, img, logits, workspace, workspace_bytes, kept_idx, compl_idx, soft_out, noise_in, features_out, tokens_out,
                          B, s, nullptr, nullptr);
}

extern "C" size_t tr_vit_tape_bytes(const tr_vit_config* cfg, int B) {
  Plan p;
  trplan::TokenPlan t;
  trplan::TapePlan tp;
  if (!make_plan(cfg, B, &p) || cfg->precision != TR_PREC_BF16 || !trplan::trainable_family(cfg->family)) return 0;
  // beyond 224 tokens (384 x 384 inputs) the attention backward runs key-blocked (tr_attention_bwd_long.hip)
  if (p.N0 > 640) return 0;
  if (!trplan::make_token_plan(cfg, &t) || !trplan::make_tape_plan(cfg, B, t, &tp)) return 0;
  return tp.total;
}

// offsets (bytes) of block blk's tape slots, in BlockTape order: x0,x1,xn1,qkv,ao,dattn,x2,xn2,pre,h,idx,idx2,scores,size + token
// counts n_pre,n_att,n_mlp,kk: lets a host read the decisions (kept ids, sizes) a training forward left on the tape
extern "C" int tr_vit_tape_layout(const tr_vit_config* cfg, int B, int blk, size_t* out18) {
  Plan p;
  trplan::TokenPlan t;
  trplan::TapePlan tp;
  TR_REQUIRE(cfg && out18, TR_ERR_NULL, "tr_vit_tape_layout: null pointer");
  TR_REQUIRE(make_plan(cfg, B, &p) && trplan::make_token_plan(cfg, &t) && trplan::make_tape_plan(cfg, B, t, &tp) && blk >= 0 && blk < cfg->depth,
             TR_ERR_CONFIG, "tr_vit_tape_layout: invalid config or block");
  const trplan::BlockTape& b = tp.blk[blk];
  const bool dy = cfg->family == TR_FAMILY_DYVIT;       // DyViT: the stage's one-hot decisions / policy [B,N] sit in the "scores" / "size" places
  const size_t v[18] = {b.x0, b.x1, b.xn1, b.qkv, b.ao, b.dattn, b.x2, b.xn2, b.pre, b.h, b.idx, b.idx2, dy ? b.hard : b.scores,
                        dy ? b.pol : b.size,
                        (size_t)t.n_pre[blk], (size_t)t.n_att[blk], (size_t)t.n_mlp[blk], (size_t)t.kk[blk]};
  for (int i = 0; i < 18; ++i) out18[i] = v[i];
  return TR_OK;
}

extern "C" int tr_vit_forward_train(const tr_vit_config* cfg, const tr_vit_weights* w, const float* img, float* logits, void* workspace,
                                    size_t workspace_bytes, void* tape, size_t tape_bytes, const float* noise_in, float* features_out,
                                    const float* drop_scale, int* tokens_out, int B, tr_stream_t s, const uint8_t* dropout_keep, float drop_rate) {
  TR_REQUIRE(cfg && tape, TR_ERR_NULL, "tr_vit_forward_train: null pointer");
  TR_REQUIRE((dropout_keep == nullptr) == (drop_rate == 0.f) && drop_rate >= 0.f && drop_rate < 1.f, TR_ERR_CONFIG,
             "tr_vit_forward_train: dropout needs a keep mask AND 0 < drop_rate < 1 (got mask %p, rate %g)", (const void*)dropout_keep, (double)drop_rate);
  TR_REQUIRE(cfg->precision == TR_PREC_BF16, TR_ERR_CONFIG, "tr_vit_forward_train: the training path is bf16 only");
  TR_REQUIRE(trplan::trainable_family(cfg->family), TR_ERR_CONFIG, "tr_vit_forward_train: family %d has no training path yet", cfg->family);
  trplan::TokenPlan t;
  trplan::TapePlan tp;
  TR_REQUIRE(trplan::make_token_plan(cfg, &t) && trplan::make_tape_plan(cfg, B, t, &tp), TR_ERR_CONFIG, "tr_vit_forward_train: invalid config");
  TR_REQUIRE(tape_bytes >= tp.total, TR_ERR_SHAPE, "tr_vit_forward_train: tape too small (%zu < %zu)", tape_bytes, tp.total);
  TR_REQUIRE(tr_aligned16(tape), TR_ERR_ALIGN, "tr_vit_forward_train: tape must be 16-byte aligned");
  for (int i = 0; i < cfg->depth; ++i)
    TR_REQUIRE(t.n_att[i] <= 640, TR_ERR_SHAPE, "tr_vit_forward_train: %d tokens in block %d (the training path holds 640)", t.n_att[i], i);
  TR_REQUIRE(features_out == nullptr || cfg->family == TR_FAMILY_DYVIT, TR_ERR_CONFIG, "tr_vit_forward_train: features_out is DyViT's distillation output");
  return vit_forward_impl(cfg, w, img, logits, workspace, workspace_bytes, nullptr, nullptr, nullptr, noise_in, features_out, tokens_out, B, s,
                          static_cast<char*>(tape), &tp, drop_scale, dropout_keep, drop_rate);
}

// Bytes of the dropout keep mask of one training forward (1 byte per element, 1 = keep), in the order the forward consumes it:
// the embedded tokens [B,N0,D] (pos_drop), then per block proj's output rows [B*n_proj,D], the Mlp's hidden layer [B*n_mlp,Hd] and its
// output [B*n_mlp,D] -- the order in which the reference module draws them (topk.py:186, :53, timm Mlp).  0 = no training path.
extern "C" size_t tr_vit_dropout_mask_bytes(const tr_vit_config* cfg, int B) {
  trplan::TokenPlan t;
  if (cfg == nullptr || B <= 0 || !trplan::make_token_plan(cfg, &t)) return 0;
  const size_t D = (size_t)cfg->embed_dim, Hd = (size_t)cfg->mlp_hidden;
  size_t n = (size_t)B * t.N0 * D;
  for (int i = 0; i < cfg->depth; ++i) n += (size_t)B * (trplan::proj_rows(cfg, t, i) * D + (size_t)t.n_mlp[i] * (Hd + D));
  return n;
}
