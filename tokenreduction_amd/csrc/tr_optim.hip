// Fused AdamW step of the fine-tune path (engine.py:76-91: loss.backward(); optimizer.step(); the reference's optimizer is
// torch.optim.AdamW built by optim.py / timm's factory).  ONE launch updates every parameter and leaves the executor ready for the next
// forward: per 64 x 64 tile of every parameter tensor it reads p, g, exp_avg, exp_avg_sq (fp32), applies the AdamW update, writes the
// three states back, ZEROES the gradient when asked to, and -- for the matrices the bf16 executor reads -- writes the bf16 operand copy and its
// transposed copy (what tr_cast_pack_bf16 did in a second pass over the parameters).  Replaces, per training step of DeiT-B:
// torch's multi_tensor_apply AdamW kernels (4.9 % of the step), ~69 fill launches and the cast_pack pass (1.1 %).
//
// Arithmetic = torch's fused AdamW (aten/src/ATen/native/cuda/fused_adam_utils.cuh adam_math, ADAMW mode, opmath float, scalars
// double), expression by expression, so that the parameters stay bit-identical to torch.optim.AdamW(fused=True):
//   p  -= lr * wd * p                                  (double, rounded to float on the store)
//   m   = beta1 * m + (1 - beta1) * g                  (double)
//   v   = beta2 * v + (1 - beta2) * g * g              (double)
//   p  -= (lr / bc1) * m / (sqrt(v) / bc2_sqrt + eps)  (step size and eps sum in double -> float, the rest float)
// with bc1 = float(1 - beta1^step), bc2_sqrt = float(sqrt(1 - beta2^step)) computed by the caller in double.
// The double expressions are FMA-contracted in torch's build, and which product is folded decides the float the result rounds to when
// the double lands next to a tie (0.6 % of exp_avg after two steps).  The contractions are therefore spelled out here, in the form
// matched offline against torch's outputs (tools/adamw_dbg.py): m = fma(beta1, m, (1 - beta1) * g), v = fma(beta2, v, ((1 - beta2) * g) * g),
// p = fma(-(lr * wd), p, p).
#include "tr_common.h"

namespace {

struct adamw_groups { double lr[8]; double wd[8]; };

template <bool ZERO>
__global__ __launch_bounds__(256) void adamw_pack_kernel(const tr_adamw_item* __restrict__ items, const int* __restrict__ first, int n_items,
                                                         double beta1, double beta2, double eps, float bc1, float bc2_sqrt, adamw_groups G) {
  __shared__ unsigned short tile[64][66];
  int it = 0;
  {     // the item this tile belongs to: binary search over the prefix sums
    int lo = 0, hi = n_items;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
    it = lo;
  }
  const tr_adamw_item I = items[it];
  const int tiles_c = (I.cols + 63) >> 6;
  const int t = blockIdx.x - first[it], r0 = (t / tiles_c) << 6, c0 = (t % tiles_c) << 6;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 x 16 threads, 4 columns each, 4 row passes
  float* P = static_cast<float*>(I.p);
  float* Gd = static_cast<float*>(I.g);
  float* Mo = static_cast<float*>(I.m);
  float* Vo = static_cast<float*>(I.v);
  uint16_t* dst = static_cast<uint16_t*>(I.dst);
  uint16_t* dst_t = static_cast<uint16_t*>(I.dst_t);
  const double lr = G.lr[I.group & 7], wd = G.wd[I.group & 7];
  const float step_size = (float)(lr / bc1);
  const double w1 = 1.0 - beta1, w2 = 1.0 - beta2, decay = lr * wd;
  const bool vec = (I.cols & 3) == 0 && ((reinterpret_cast<uintptr_t>(I.p) | reinterpret_cast<uintptr_t>(I.g) | reinterpret_cast<uintptr_t>(I.m) |
                                          reinterpret_cast<uintptr_t>(I.v)) & 15u) == 0 && (reinterpret_cast<uintptr_t>(I.dst) & 7u) == 0;
  auto upd = [&](float& p, float& g, float& m, float& v) __attribute__((always_inline)) {
    if (wd != 0.0) p = (float)__fma_rn(-decay, (double)p, (double)p);
    m = (float)__fma_rn(beta1, (double)m, __dmul_rn(w1, (double)g));
    v = (float)__fma_rn(beta2, (double)v, __dmul_rn(__dmul_rn(w2, (double)g), (double)g));
    const float denom = (float)((double)(sqrtf(v) / bc2_sqrt) + eps);
    p -= step_size * m / denom;
  };
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int r = r0 + ty + 16 * ps, c = c0 + 4 * tx;
    float pv[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < I.rows) {
      const size_t o = (size_t)r * I.cols + c;
      if (vec && c + 3 < I.cols) {
        float4 p4 = *reinterpret_cast<const float4*>(P + o), g4 = *reinterpret_cast<const float4*>(Gd + o);
        float4 m4 = *reinterpret_cast<const float4*>(Mo + o), v4 = *reinterpret_cast<const float4*>(Vo + o);
        upd(p4.x, g4.x, m4.x, v4.x); upd(p4.y, g4.y, m4.y, v4.y); upd(p4.z, g4.z, m4.z, v4.z); upd(p4.w, g4.w, m4.w, v4.w);
        *reinterpret_cast<float4*>(P + o) = p4;
        if (ZERO) *reinterpret_cast<float4*>(Gd + o) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(Mo + o) = m4; *reinterpret_cast<float4*>(Vo + o) = v4;
        pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < I.cols) {
            float p = P[o + e], g = Gd[o + e], m = Mo[o + e], v = Vo[o + e];
            upd(p, g, m, v);
            P[o + e] = p; Mo[o + e] = m; Vo[o + e] = v;
            if (ZERO) Gd[o + e] = 0.f;
            pv[e] = p;
          }
      }
    }
    if (dst == nullptr && dst_t == nullptr) continue;
    const unsigned lo = pack_bf16x2(pv[0], pv[1]), hi = pack_bf16x2(pv[2], pv[3]);
    if (dst != nullptr && r < I.rows) {
      if (vec && c + 3 < I.cols) *reinterpret_cast<uint2*>(dst + (size_t)r * I.cols + c) = make_uint2(lo, hi);
      else {
        if (c < I.cols) dst[(size_t)r * I.cols + c] = (uint16_t)(lo & 0xffffu);
        if (c + 1 < I.cols) dst[(size_t)r * I.cols + c + 1] = (uint16_t)(lo >> 16);
        if (c + 2 < I.cols) dst[(size_t)r * I.cols + c + 2] = (uint16_t)(hi & 0xffffu);
        if (c + 3 < I.cols) dst[(size_t)r * I.cols + c + 3] = (uint16_t)(hi >> 16);
      }
    }
    if (dst_t != nullptr) {
      tile[ty + 16 * ps][4 * tx] = (unsigned short)(lo & 0xffffu);
      tile[ty + 16 * ps][4 * tx + 1] = (unsigned short)(lo >> 16);
      tile[ty + 16 * ps][4 * tx + 2] = (unsigned short)(hi & 0xffffu);
      tile[ty + 16 * ps][4 * tx + 3] = (unsigned short)(hi >> 16);
    }
  }
  if (dst_t == nullptr) return;
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int c = c0 + ty + 16 * ps, r = r0 + 4 * tx;               // transposed: row c of dst_t, columns r .. r + 3
    if (c >= I.cols) continue;
    const unsigned short a = tile[4 * tx][ty + 16 * ps], b = tile[4 * tx + 1][ty + 16 * ps], cc = tile[4 * tx + 2][ty + 16 * ps],
                         d = tile[4 * tx + 3][ty + 16 * ps];
    uint16_t* o = dst_t + (size_t)c * I.rows + r;
    if (r + 3 < I.rows && (I.rows & 3) == 0 && (reinterpret_cast<uintptr_t>(dst_t) & 7u) == 0)
      *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)a | ((unsigned)b << 16), (unsigned)cc | ((unsigned)d << 16));
    else {
      if (r < I.rows) o[0] = a;
      if (r + 1 < I.rows) o[1] = b;
      if (r + 2 < I.rows) o[2] = cc;
      if (r + 3 < I.rows) o[3] = d;
    }
  }
}

}  // namespace

extern "C" int tr_adamw_step(const tr_adamw_item* items, const int* first, int n_items, int total_tiles, double beta1, double beta2, double eps,
                             float bias_correction1, float bias_correction2_sqrt, const double* lr8, const double* wd8, int zero_grads,
                             tr_stream_t s) {
  TR_REQUIRE(items && first && lr8 && wd8, TR_ERR_NULL, "tr_adamw_step: null pointer");
  TR_REQUIRE(n_items > 0 && total_tiles > 0, TR_ERR_SHAPE, "tr_adamw_step: nothing to update (n_items=%d tiles=%d)", n_items, total_tiles);
  TR_REQUIRE(bias_correction1 > 0.f && bias_correction2_sqrt > 0.f, TR_ERR_SHAPE, "tr_adamw_step: bias corrections must be positive");
  adamw_groups G;
  for (int i = 0; i < 8; ++i) { G.lr[i] = lr8[i]; G.wd[i] = wd8[i]; }
  if (zero_grads)
    hipLaunchKernelGGL(adamw_pack_kernel<true>, dim3((unsigned)total_tiles), dim3(256), 0, static_cast<hipStream_t>(s), items, first, n_items, beta1,
                       beta2, eps, bias_correction1, bias_correction2_sqrt, G);
  else
    hipLaunchKernelGGL(adamw_pack_kernel<false>, dim3((unsigned)total_tiles), dim3(256), 0, static_cast<hipStream_t>(s), items, first, n_items, beta1,
                       beta2, eps, bias_correction1, bias_correction2_sqrt, G);
  TR_CHECK_LAUNCH("tr_adamw_step");
  return TR_OK;
}
