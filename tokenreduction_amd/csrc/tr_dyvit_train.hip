// DynamicViT TRAINING path (models/dyvit.py:221-229, 245-246): nothing is pruned while training -- every block attends under a keep
// policy (Policy_Attention.softmax_with_policy dyvit.py:39-51, forward kernel in tr_attention.hip) and the policy of each pruning
// stage is a straight-through Gumbel-softmax sample of the PredictorLG scores (dyvit.py:223-224).  This file has the pieces of the
// predictor that differ from the eval path, and their gradients:
//   tr_pool_policy        PredictorLG.forward dyvit.py:115-118 with a real policy: channels C/2.. of every row become the
//                         policy-weighted mean over the image's patch rows (+ eps)
//   tr_pool_policy_bwd    its gradient wrt the rows and wrt the policy
//   tr_dyvit_decide       out_conv.4 + LogSoftmax + F.gumbel_softmax(hard=True)[:, :, 0:1] * prev_decision (dyvit.py:108-109,223-224):
//                         the Gumbel noise is an INPUT (torch draws it: -log(Exp(1))), so a run can be replayed
//   tr_dyvit_decide_bwd   straight-through gradient back to the D/4-wide hidden layer, the 2 x D/4 weight and prev_decision
// The Linear layers of the predictor run through tr_gemm_bf16 / tr_wgrad_bf16 like every other Linear.
#include "tr_common.h"

extern "C" int tr_reduce_partials_f32(const float* part, int S, size_t count, float* dst, int accumulate, tr_stream_t s);

namespace {

// grid (ceil(C/2/64), B), 4 waves: wave w sums patch rows 1+w, 5+w, ... of its 64 channels (fixed order), combined in wave order
__global__ __launch_bounds__(256) void pool_policy_kernel(uint16_t* __restrict__ h, const float* __restrict__ policy, int N, int C,
                                                          float eps) {
  __shared__ float part[4][64];
  __shared__ float psum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int c = C / 2 + blockIdx.x * 64 + lane;
  const bool ok = c < C;
  const size_t base = (size_t)b * N * C;
  const float* pol = policy + (size_t)b * N;            // [B,N]: entry 0 = CLS (unused), 1.. = prev_decision
  float acc = 0.f, ps = 0.f;
  for (int n = 1 + wave; n < N; n += 4) {
    const float p = pol[n];
    ps += p;
    if (ok) acc += bf16_bits_to_f32(h[base + (size_t)n * C + c]) * p;
  }
  part[wave][lane] = acc;
  if (lane == 0) psum[wave] = ps;
  __syncthreads();
  const float tot = (psum[0] + psum[1]) + (psum[2] + psum[3]);
  const float g = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / tot + eps;
  const uint16_t gb = (uint16_t)(pack_bf16x2(g, 0.f) & 0xffffu);
  if (ok)
    for (int n = wave; n < N; n += 4) h[base + (size_t)n * C + c] = gb;
}

// dcat bf16 [B,N,C] (gradient wrt [local | global]) -> dh bf16 [B,N,C] (wrt the GELU output h0) and dpolicy [B,N] (+=).
//   glob_c = sum_p h0[p,c] pol_p / S + eps, S = sum_p pol_p;   G_c = sum_{rows} dcat[row, c]  (the global half is broadcast to all rows,
//   the CLS row included in the forward's layout but its gradient is zero by construction)
//   d h0[p,c] = G_c pol_p / S;   d pol_p = sum_c G_c (h0[p,c] - glob_c + eps) / S
// h0's global half was overwritten by the broadcast, so it is recomputed from the saved pre-activation (same GELU fit as the forward).
// One workgroup per image; phase 1: G_c (thread per channel pair), phase 2: one wave per row.
__global__ __launch_bounds__(256) void pool_policy_bwd_kernel(const uint16_t* __restrict__ dcat, const uint16_t* __restrict__ pre0,
                                                              const uint16_t* __restrict__ cat, const float* __restrict__ policy,
                                                              uint16_t* __restrict__ dh, float* __restrict__ dpolicy, int N, int C) {
  extern __shared__ float sG[];          // [C/2] column sums of the global half, then [C/2] glob - eps
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Ch = C / 2;
  const size_t base = (size_t)b * N * C;
  const float* pol = policy + (size_t)b * N;
  float* sGl = sG + Ch;
  for (int c = tid; c < Ch; c += 256) {
    float a = 0.f;
    for (int n = 1; n < N; ++n) a += bf16_bits_to_f32(dcat[base + (size_t)n * C + Ch + c]);
    sG[c] = a;
    sGl[c] = bf16_bits_to_f32(cat[base + (size_t)C + Ch + c]);      // the broadcast value (row 1) = glob + eps
  }
  float S = 0.f;
  for (int n = 1 + lane; n < N; n += 64) S += pol[n];
  S = wave_sum(S);
  __syncthreads();
  const float invS = 1.0f / S;
  for (int n = wave; n < N; n += 4) {
    const float p = n == 0 ? 0.f : pol[n];
    float dp = 0.f;
    for (int c = lane; c < Ch; c += 64) {
      const size_t e = base + (size_t)n * C;
      dh[e + c] = n == 0 ? (uint16_t)0 : dcat[e + c];                                       // local half passes through
      const float g = sG[c];
      dh[e + Ch + c] = (uint16_t)(pack_bf16x2(n == 0 ? 0.f : g * p * invS, 0.f) & 0xffffu);
      if (n > 0) {
        const f32x2 hv = gelu2(f32x2{bf16_bits_to_f32(pre0[e + Ch + c]), 0.f});
        dp += g * (hv[0] - sGl[c]);                      // sGl = glob + eps (1e-6: below the bf16 resolution of the broadcast value)
      }
    }
    dp = wave_sum(dp);
    if (lane == 0 && n > 0) dpolicy[(size_t)b * N + n] += dp * invS;
  }
}

// The same with 16-byte accesses (C / 2 a multiple of 8: every registered width).  Phase 1: thread = (8-channel chunk, row slice), the
// slices' partial column sums combined through LDS in slice order; phase 2: one wave per row, lane = chunk, both halves of the row.
// (The element-wise version above moves 2 bytes per lane and instruction: 192 us at B = 256, N = 197, C = 384.)
__global__ __launch_bounds__(256) void pool_policy_bwd_vec_kernel(const uint16_t* __restrict__ dcat, const uint16_t* __restrict__ pre0,
                                                                  const uint16_t* __restrict__ cat, const float* __restrict__ policy,
                                                                  uint16_t* __restrict__ dh, float* __restrict__ dpolicy, int N, int C) {
  extern __shared__ float sG[];          // [C/2] column sums, [C/2] glob + eps, [nsl][C/2] slice partials
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Ch = C / 2, nch = Ch / 8;    // 8-channel chunks of a half row
  const size_t base = (size_t)b * N * C;
  const float* pol = policy + (size_t)b * N;
  float* sGl = sG + Ch;
  float* sPart = sG + 2 * Ch;
  const int nsl = 256 / nch;             // row slices (nch <= 256)
  const int ck = tid % nch, sl = tid / nch;
  if (sl < nsl) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int n = 1 + sl; n < N; n += nsl) {
      const uint4 u = *reinterpret_cast<const uint4*>(dcat + base + (size_t)n * C + Ch + 8 * ck);
      const unsigned int w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[2 * e] += __uint_as_float(w[e] << 16);
        a[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sPart[sl * Ch + 8 * ck + e] = a[e];
  }
  float S = 0.f;
  for (int n = 1 + lane; n < N; n += 64) S += pol[n];
  S = wave_sum(S);
  __syncthreads();
  for (int c = tid; c < Ch; c += 256) {
    float a = 0.f;
    for (int q = 0; q < nsl; ++q) a += sPart[q * Ch + c];      // slice order: fixed
    sG[c] = a;
    sGl[c] = bf16_bits_to_f32(cat[base + (size_t)C + Ch + c]);      // the broadcast value (row 1) = glob + eps
  }
  __syncthreads();
  const float invS = 1.0f / S;
  for (int n = wave; n < N; n += 4) {
    const float p = n == 0 ? 0.f : pol[n];
    float dp = 0.f;
    for (int c8 = lane; c8 < nch; c8 += 64) {
      const size_t e = base + (size_t)n * C + 8 * c8;
      const uint4 loc = n == 0 ? make_uint4(0u, 0u, 0u, 0u) : *reinterpret_cast<const uint4*>(dcat + e);
      *reinterpret_cast<uint4*>(dh + e) = loc;                                                   // local half passes through
      const uint4 pu = *reinterpret_cast<const uint4*>(pre0 + e + Ch);
      const unsigned int pw[4] = {pu.x, pu.y, pu.z, pu.w};
      unsigned int o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float g0 = sG[8 * c8 + 2 * q], g1 = sG[8 * c8 + 2 * q + 1];
        o[q] = n == 0 ? 0u : pack_bf16x2(g0 * p * invS, g1 * p * invS);
        if (n > 0) {
          const f32x2 hv = gelu2(f32x2{__uint_as_float(pw[q] << 16), __uint_as_float(pw[q] & 0xffff0000u)});
          dp += g0 * (hv[0] - sGl[8 * c8 + 2 * q]) + g1 * (hv[1] - sGl[8 * c8 + 2 * q + 1]);
        }
      }
      *reinterpret_cast<uint4*>(dh + e + Ch) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    dp = wave_sum(dp);
    if (lane == 0 && n > 0) dpolicy[(size_t)b * N + n] += dp * invS;
  }
}

// 16 lanes per row: z = h2 W3^T + b3, score = log_softmax(z), y = softmax(score + gumbel), hard = (score0 + g0 >= score1 + g1),
// keep = hard * prev.  Row 0 of every image is the CLS token: policy 1, nothing else written.
__global__ __launch_bounds__(256) void dyvit_decide_kernel(const uint16_t* __restrict__ h2, int ldh, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ gumbel,
                                                           const float* __restrict__ prev, float* __restrict__ policy_out,
                                                           float* __restrict__ ysoft0, float* __restrict__ sm0, float* __restrict__ hard0,
                                                           int N, int M, int C) {
  const int sub = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  float a0 = 0.f, a1 = 0.f;
  if (row < M)
    for (int c = sub; c < C; c += 16) {
      const float v = bf16_bits_to_f32(h2[(size_t)row * ldh + c]);
      a0 = fmaf(v, w[c], a0);
      a1 = fmaf(v, w[C + c], a1);
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    a0 += __shfl_xor(a0, o, 64);
    a1 += __shfl_xor(a1, o, 64);
  }
  if (row >= M || sub != 0) return;
  const int b = row / N, n = row - b * N;
  if (n == 0) {
    policy_out[row] = 1.0f;
    return;
  }
  const float l0 = a0 + bias[0], l1 = a1 + bias[1];
  const float m = fmaxf(l0, l1);
  const float lse = logf(expf(l0 - m) + expf(l1 - m));
  const float s0 = (l0 - m) - lse, s1 = (l1 - m) - lse;         // torch log_softmax
  const size_t pe = (size_t)b * (N - 1) + (n - 1);
  const float t0 = s0 + gumbel[2 * pe], t1 = s1 + gumbel[2 * pe + 1];
  const float tm = fmaxf(t0, t1);
  const float e0 = expf(t0 - tm), e1 = expf(t1 - tm);
  const float y0 = e0 / (e0 + e1);
  const float hard = t0 >= t1 ? 1.0f : 0.0f;                    // argmax, first index on a tie (torch.max)
  policy_out[row] = hard * prev[row];
  ysoft0[row] = y0;
  sm0[row] = expf(s0);
  hard0[row] = hard;
}

// d keep [B,N] (entry 0 unused) -> dz, then dh2 bf16 [M, ldh] = dz W3, d prev [B,N] (+)= d keep * hard, and per-workgroup partials of
// dW3 [2,C] / db3 [2] (part[wg][2C+2], reduced by the caller's partial reduce).  16 lanes per row.
__global__ __launch_bounds__(256) void dyvit_decide_bwd_kernel(const float* __restrict__ dkeep, const float* __restrict__ prev,
                                                               const float* __restrict__ hard0, const float* __restrict__ ysoft0,
                                                               const float* __restrict__ sm0, const uint16_t* __restrict__ h2, int ldh,
                                                               const float* __restrict__ w, uint16_t* __restrict__ dh2,
                                                               float* __restrict__ dprev, float* __restrict__ part, int N, int M, int C) {
  __shared__ float sdz[16][2];           // the 16 rows' dz: the weight-gradient partials below are summed over the rows in a FIXED order
  const int sub = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int row = blockIdx.x * 16 + rl;
  float dz0 = 0.f, dz1 = 0.f;
  if (row < M) {
    const int n = row % N;
    if (n > 0) {
      const float pv = prev[row];
      const float dk = dkeep[row];
      const float hard = hard0[row];                                    // forward value of the one-hot's first entry
      const float dy0 = dk * pv;                                        // straight-through: gradient goes to y_soft[..., 0]
      const float y0 = ysoft0[row], y1 = 1.0f - y0;
      const float dl0 = y0 * y1 * dy0, dl1 = -y0 * y1 * dy0;           // softmax over the 2 logits (tau = 1)
      const float q0 = sm0[row];                                        // softmax(z)_0; log_softmax backward: dz = dl - softmax * sum(dl)
      dz0 = dl0 - q0 * (dl0 + dl1);
      dz1 = dl1 - (1.0f - q0) * (dl0 + dl1);
      if (sub == 0) dprev[row] += dk * hard;
    }
    for (int c = sub; c < C; c += 16) dh2[(size_t)row * ldh + c] = (uint16_t)(pack_bf16x2(dz0 * w[c] + dz1 * w[C + c], 0.f) & 0xffffu);
    for (int c = C + sub; c < ldh; c += 16) dh2[(size_t)row * ldh + c] = 0;      // zero-padded columns of the D/4 layer
  }
  if (sub == 0) { sdz[rl][0] = dz0; sdz[rl][1] = dz1; }                 // rows beyond M and the class token: zeros
  __syncthreads();
  // dW3[k][c] partial = sum_r dz[r][k] * h2[r][c], db3[k] partial = sum_r dz[r][k]: one thread per output, rows 0..15 in order (round 6: these
  // were LDS float atomics -- the only run-to-run variation of the training path, at the 1e-6 level; tools/lab/train_soak.py)
  const int nrow = min(16, M - blockIdx.x * 16);
  for (int i = threadIdx.x; i < 2 * C + 2; i += 256) {
    float a = 0.f;
    if (i < 2 * C) {
      const int k = i >= C ? 1 : 0, c = i - k * C;
      const uint16_t* hp = h2 + (size_t)blockIdx.x * 16 * ldh + c;
      for (int r = 0; r < nrow; ++r) a += sdz[r][k] * bf16_bits_to_f32(hp[(size_t)r * ldh]);
    } else {
      const int k = i - 2 * C;
      for (int r = 0; r < 16; ++r) a += sdz[r][k];
    }
    part[(size_t)blockIdx.x * (2 * C + 4) + i] = a;
  }
}

// out[i] (+)= sum over heads of part[b][h][n]
__global__ __launch_bounds__(256) void head_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int H, int N) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * N) return;
  const int b = e / N, n = e - b * N;
  float a = 0.f;
  for (int h = 0; h < H; ++h) a += part[((size_t)b * H + h) * N + n];
  out[e] += a;
}

__global__ __launch_bounds__(256) void fill_f32_kernel(float* __restrict__ p, float v, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ dst, const float* __restrict__ src, int B, int N) {
  // dst[b][n] += src[b][n-1] for n >= 1 (src [B,N-1]: a gradient given per patch token)
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * N) return;
  const int b = e / N, n = e - b * N;
  if (n > 0) dst[e] += src[(size_t)b * (N - 1) + n - 1];
}

}  // namespace

extern "C" int tr_pool_policy(uint16_t* h, const float* policy, int B, int N, int C, float eps, tr_stream_t s) {
  TR_REQUIRE(h && policy, TR_ERR_NULL, "tr_pool_policy: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && C >= 2 && C % 2 == 0, TR_ERR_SHAPE, "tr_pool_policy: bad shape B=%d N=%d C=%d", B, N, C);
  hipLaunchKernelGGL(pool_policy_kernel, dim3((C / 2 + 63) / 64, B), dim3(256), 0, static_cast<hipStream_t>(s), h, policy, N, C, eps);
  TR_CHECK_LAUNCH("tr_pool_policy");
  return TR_OK;
}

extern "C" int tr_pool_policy_bwd(const uint16_t* dcat, const uint16_t* pre0, const uint16_t* cat, const float* policy, uint16_t* dh,
                                  float* dpolicy, int B, int N, int C, tr_stream_t s) {
  TR_REQUIRE(dcat && pre0 && cat && policy && dh && dpolicy, TR_ERR_NULL, "tr_pool_policy_bwd: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && C >= 2 && C % 2 == 0 && C <= 4096, TR_ERR_SHAPE, "tr_pool_policy_bwd: bad shape B=%d N=%d C=%d", B, N, C);
  const int Ch = C / 2;
  if (Ch % 8 == 0 && Ch / 8 <= 256 && tr_aligned16(dcat) && tr_aligned16(pre0) && tr_aligned16(dh)) {
    const size_t lds = ((size_t)2 * Ch + (size_t)(256 / (Ch / 8)) * Ch) * sizeof(float);
    hipLaunchKernelGGL(pool_policy_bwd_vec_kernel, dim3(B), dim3(256), lds, static_cast<hipStream_t>(s), dcat, pre0, cat, policy, dh, dpolicy, N, C);
  } else {
    hipLaunchKernelGGL(pool_policy_bwd_kernel, dim3(B), dim3(256), (size_t)C * sizeof(float), static_cast<hipStream_t>(s), dcat, pre0, cat, policy, dh,
                       dpolicy, N, C);
  }
  TR_CHECK_LAUNCH("tr_pool_policy_bwd");
  return TR_OK;
}

extern "C" int tr_dyvit_decide(const uint16_t* h2, int ldh, const float* w, const float* bias, const float* gumbel, const float* prev,
                               float* policy_out, float* ysoft0, float* sm0, float* hard0, int B, int N, int C, tr_stream_t s) {
  TR_REQUIRE(h2 && w && bias && gumbel && prev && policy_out && ysoft0 && sm0 && hard0, TR_ERR_NULL, "tr_dyvit_decide: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && C >= 1 && ldh >= C, TR_ERR_SHAPE, "tr_dyvit_decide: bad shape B=%d N=%d C=%d ldh=%d", B, N, C, ldh);
  const int M = B * N;
  hipLaunchKernelGGL(dyvit_decide_kernel, dim3((M + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(s), h2, ldh, w, bias, gumbel, prev, policy_out,
                     ysoft0, sm0, hard0, N, M, C);
  TR_CHECK_LAUNCH("tr_dyvit_decide");
  return TR_OK;
}

extern "C" size_t tr_dyvit_decide_bwd_workspace_floats(int B, int N, int C) { return (size_t)((B * N + 15) / 16 + 1) * (2 * C + 4); }

extern "C" int tr_dyvit_decide_bwd(const float* dkeep, const float* prev, const float* hard0, const float* ysoft0, const float* sm0,
                                   const uint16_t* h2, int ldh, const float* w, uint16_t* dh2, float* dprev, float* dw, float* db,
                                   int accumulate, float* ws, size_t ws_floats, int B, int N, int C, tr_stream_t s) {
  TR_REQUIRE(dkeep && prev && hard0 && ysoft0 && sm0 && h2 && w && dh2 && dprev && dw && db && ws, TR_ERR_NULL, "tr_dyvit_decide_bwd: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && C >= 1 && ldh >= C && C <= 4096, TR_ERR_SHAPE, "tr_dyvit_decide_bwd: bad shape");
  const int M = B * N, nwg = (M + 15) / 16;
  TR_REQUIRE(ws_floats >= tr_dyvit_decide_bwd_workspace_floats(B, N, C), TR_ERR_SHAPE, "tr_dyvit_decide_bwd: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(s);
  hipLaunchKernelGGL(dyvit_decide_bwd_kernel, dim3(nwg), dim3(256), 0, st, dkeep, prev, hard0, ysoft0, sm0, h2,
                     ldh, w, dh2, dprev, ws, N, M, C);
  TR_CHECK_LAUNCH("tr_dyvit_decide_bwd");
  // reduce the per-workgroup partials [nwg][2C+4] (entries 2C+2, 2C+3 are padding) into one row, then split into dW3 and db3
  float* tail = ws + (size_t)nwg * (2 * C + 4);
  int rc = tr_reduce_partials_f32(ws, nwg, (size_t)2 * C + 4, tail, 0, s);
  if (rc != TR_OK) return rc;
  rc = tr_reduce_partials_f32(tail, 1, (size_t)2 * C, dw, accumulate, s);
  if (rc != TR_OK) return rc;
  return tr_reduce_partials_f32(tail + 2 * C, 1, (size_t)2, db, accumulate, s);
}

// dst[b][n] += sum_h part[b][h][n]  (the per-head policy gradients of tr_attention_policy_bwd_bf16)
extern "C" int tr_head_sum(const float* part, float* dst, int B, int H, int N, tr_stream_t s) {
  TR_REQUIRE(part && dst, TR_ERR_NULL, "tr_head_sum: null pointer");
  hipLaunchKernelGGL(head_sum_kernel, dim3((B * N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(s), part, dst, B, H, N);
  TR_CHECK_LAUNCH("tr_head_sum");
  return TR_OK;
}

extern "C" int tr_fill_f32(float* p, float v, size_t n, tr_stream_t s) {
  TR_REQUIRE(p, TR_ERR_NULL, "tr_fill_f32: null pointer");
  hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(s), p, v, n);
  TR_CHECK_LAUNCH("tr_fill_f32");
  return TR_OK;
}

extern "C" int tr_add_patch_rows(float* dst, const float* src, int B, int N, tr_stream_t s) {
  TR_REQUIRE(dst && src, TR_ERR_NULL, "tr_add_patch_rows: null pointer");
  hipLaunchKernelGGL(add_rows_kernel, dim3((B * N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(s), dst, src, B, N);
  TR_CHECK_LAUNCH("tr_add_patch_rows");
  return TR_OK;
}
