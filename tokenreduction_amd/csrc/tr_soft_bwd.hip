// Backward of the soft-assignment reducers that run BEFORE a block: SiT's TokenSlimmingModule (sit.py:36-40), PatchMerger
// (patchmerger.py:35-39) and Sinkhorn (sinkhorn.py:41-86).  All three end in
//     out[b,k,:] = sum_p W[b,p,k] * src[b,p,:]            (src = the tokens, or their LayerNorm / unit-norm image)
// with W a per-output softmax over the TOKENS (SiT, PatchMerger) or a Sinkhorn transport plan.  Given g = d out:
//     dW[p,k]   = <g[k,:], src[p,:]>                       tr_soft_dweights   (batched fp32 product, contraction over D)
//     dsrc[p,:] = sum_k W[p,k] g[k,:]                      tr_soft_dsrc       (batched fp32 product, contraction over K)
//     softmax:  ds[p,k] = scale * W[p,k] (dW[p,k] - sum_p' W[p',k] dW[p',k]),   d scale = sum W (dW - c) * logits
//     Sinkhorn: the unrolled log-domain iterations backwards                    tr_sinkhorn_bwd
// Per image the matrices are ~196 x 137 x D: a few GFLOP per stage and batch, on the vector ALUs in fp32 (64 x 64 tiles).
// Layouts as the forward's: token-major weights [B, N, ldl] with row 0 of an image (CLS) unused.
#include "tr_common.h"

namespace {

constexpr int GT = 64, GK = 16;

// C[b][i][j] = sum_l A[b][i][l] * (NN ? Bm[b][l][j] : Bm[b][j][l]);  I x J output, contraction L; all fp32, row strides in elements
template <bool NN>
__global__ __launch_bounds__(256) void bgemm_f32_kernel(const float* __restrict__ A, long sA, int lda, const float* __restrict__ Bm, long sB,
                                                        int ldb, float* __restrict__ Cm, long sC, int ldc, int I, int J, int L) {
  __shared__ float sa[GK][GT + 4];
  __shared__ float sb[GK][GT + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int nJt = (J + GT - 1) / GT;
  const int i0 = (blockIdx.x / nJt) * GT, j0 = (blockIdx.x % nJt) * GT;
  const float* a = A + (size_t)blockIdx.y * sA;
  const float* bm = Bm + (size_t)blockIdx.y * sB;
  float* c = Cm + (size_t)blockIdx.y * sC;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int l0 = 0; l0 < L; l0 += GK) {
    __syncthreads();
    {   // A tile: 64 rows x 16 l, thread -> row tid>>2, 4 consecutive l
      const int r = tid >> 2, lq = (tid & 3) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = i0 + r, l = l0 + lq + e;
        sa[lq + e][r] = (i < I && l < L) ? a[(size_t)i * lda + l] : 0.f;
      }
    }
    if (NN) {   // B tile [l][j]: thread -> l = tid>>4, 4 consecutive j
      const int l = tid >> 4, jq = (tid & 15) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j0 + jq + e;
        sb[l][jq + e] = (l0 + l < L && j < J) ? bm[(size_t)(l0 + l) * ldb + j] : 0.f;
      }
    } else {    // B tile [j][l]
      const int r = tid >> 2, lq = (tid & 3) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j0 + r, l = l0 + lq + e;
        sb[lq + e][r] = (j < J && l < L) ? bm[(size_t)j * ldb + l] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GK; ++k) {
      const float4 av = *reinterpret_cast<const float4*>(&sa[k][ty * 4]);
      const float4 bv = *reinterpret_cast<const float4*>(&sb[k][tx * 4]);
      const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ii = i0 + ty * 4 + i;
    if (ii >= I) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int jj = j0 + tx * 4 + j;
      if (jj < J) c[(size_t)ii * ldc + jj] = acc[i][j];
    }
  }
}

// softmax over the token axis, backwards.  grid (ceil(K/32), B) as the forward's token_softmax_kernel: lanes along k (coalesced),
// 8 token groups.  ds (bf16, row stride ldo) gets scale * W (dW - c) on the patch rows and 0 on the CLS row; one partial of
// d scale = sum W (dW - c) logits per workgroup.
__global__ __launch_bounds__(256) void token_softmax_bwd_kernel(const float* __restrict__ wt, const float* __restrict__ dwt,
                                                                const float* __restrict__ logits, int ldl, float scale,
                                                                uint16_t* __restrict__ ds, int ldo, float* __restrict__ dscale_part,
                                                                int N, int K) {
  __shared__ float s_red[8][32];
  __shared__ float s_c[32];
  const int tid = threadIdx.x, kk = tid & 31, pg = tid >> 5;
  const int b = blockIdx.y, k = blockIdx.x * 32 + kk;
  const bool kval = k < K;
  const int P = N - 1;
  const size_t base = ((size_t)b * N + 1) * ldl + k;
  float c = 0.f;
  if (kval)
    for (int p = pg; p < P; p += 8) c += wt[base + (size_t)p * ldl] * dwt[base + (size_t)p * ldl];
  s_red[pg][kk] = c;
  __syncthreads();
  if (tid < 32) {
    float t = s_red[0][tid];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += s_red[g][tid];
    s_c[tid] = t;
  }
  __syncthreads();
  c = s_c[kk];
  float dsc = 0.f;
  if (kval) {
    uint16_t* o = ds + ((size_t)b * N + 1) * ldo + k;
    for (int p = pg; p < P; p += 8) {
      const float v = wt[base + (size_t)p * ldl] * (dwt[base + (size_t)p * ldl] - c);
      if (logits != nullptr) dsc += v * logits[base + (size_t)p * ldl];
      o[(size_t)p * ldo] = (uint16_t)(pack_bf16x2(v * scale, 0.f) & 0xffffu);
    }
    if (pg == 0) ds[(size_t)b * N * ldo + k] = 0;          // CLS row
  }
  if (dscale_part != nullptr) {
    __syncthreads();
    s_red[pg][kk] = dsc;
    __syncthreads();
    if (tid < 64) {
      float t = tid < 32 ? s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid]
                         : s_red[4][tid - 32] + s_red[5][tid - 32] + s_red[6][tid - 32] + s_red[7][tid - 32];
      t = wave_sum(t);
      if (tid == 0) dscale_part[(size_t)b * gridDim.x + blockIdx.x] = t;
    }
  }
}

// y_bf16[i] = bf16(a[i] + y_bf16[i])
__global__ __launch_bounds__(256) void add_into_bf16_kernel(const float* __restrict__ a, uint16_t* __restrict__ y, size_t n) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(a + i);
    const uint2 p = *reinterpret_cast<const uint2*>(y + i);
    uint2 o;
    o.x = pack_bf16x2(v[0] + __uint_as_float(p.x << 16), v[1] + __uint_as_float(p.x & 0xffff0000u));
    o.y = pack_bf16x2(v[2] + __uint_as_float(p.y << 16), v[3] + __uint_as_float(p.y & 0xffff0000u));
    *reinterpret_cast<uint2*>(y + i) = o;
  } else {
    for (size_t j = i; j < n; ++j) y[j] = (uint16_t)(pack_bf16x2(a[j] + bf16_bits_to_f32(y[j]), 0.f) & 0xffffu);
  }
}

// F.normalize backwards (sinkhorn.py:70): xh = x / max(|x|, 1e-12);  d xh = da (fp32) + db (bf16, nullable);
// dx = (d xh - xh <xh, d xh>) / |x|.  One wave per row.
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ da,
                                                          const uint16_t* __restrict__ db, float* __restrict__ dx, int M, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * D;
  float ss = 0.f, dot = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float xv = xr[d];
    const float g = da[(size_t)row * D + d] + (db ? bf16_bits_to_f32(db[(size_t)row * D + d]) : 0.f);
    ss += xv * xv;
    dot += xv * g;
  }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  const float inv = 1.0f / nrm, proj = dot * inv * inv * inv;      // <xh, g> / |x| * (1 / |x|) in units of x
  for (int d = lane; d < D; d += 64) {
    const float g = da[(size_t)row * D + d] + (db ? bf16_bits_to_f32(db[(size_t)row * D + d]) : 0.f);
    dx[(size_t)row * D + d] = g * inv - xr[d] * proj;
  }
}

// log_optimal_transport (sinkhorn.py:41-56) backwards, one workgroup per image.  Forward (m = K rows, n = P columns, Z0 = scores/eps,
// log_mu = log_nu = norm = -log(K+P)):  u_t = norm - LSE_p(Z0 + v_{t-1}),  v_t = norm - LSE_k(Z0 + u_t),  plan = exp(Z0 + u_T + v_T - norm).
// Backward: dZf = dplan * plan;  du_T = rowsum(dZf), dv_T = colsum(dZf);  for t = T..1:
//   A_t[k,p] = exp(Z0 + u_t + v_t - norm)      (softmax of v_t's LSE):  du_t[k] -= sum_p dv_t[p] A_t[k,p]
//   B_t[k,p] = exp(Z0 + u_t + v_{t-1} - norm)  (softmax of u_t's LSE):  dv_{t-1}[p] = - sum_k du_t[k] B_t[k,p]     (t > 1)
//   dZ0 = dZf - sum_t (dv_t[p] A_t + du_t[k] B_t);  dscores = dZ0 / eps.
// Z0 lives in LDS ([K][PP], K*PP floats), the u_t / v_t / du_t / dv_t vectors too; the iterations are recomputed, not saved.
constexpr int SB_T = 1024, SB_MAXIT = 8;

// ZLDS = false: K x P does not fit the LDS (384 x 384 inputs: 144 x 576) -- Z0 is then re-read from the scores in global memory (an
// image's matrix, 330 KB, stays in L2 over the 2 T + 3 passes); only the vectors live in LDS.
template <bool ZLDS>
__global__ __launch_bounds__(SB_T) void sinkhorn_bwd_kernel(const float* __restrict__ scores, const float* __restrict__ dplan, int ldl,
                                                            float eps, int iters, uint16_t* __restrict__ ds, int ldo, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int P = N - 1, PP = P | 1;                // odd row stride: column walks hit distinct banks
  float* z = sm;                                   // [K][PP]  (ZLDS only)
  float* u = z + (ZLDS ? (size_t)K * PP : 0);      // [iters+1][K]   u_0 = 0
  float* v = u + (size_t)(iters + 1) * K;          // [iters+1][P]   v_0 = 0
  float* du = v + (size_t)(iters + 1) * P;         // [K]   current du_t
  float* dv = du + K;                              // [P]   current dv_t
  float* dus = dv + P;                             // [iters+1][K]  du_t kept for the final pass
  float* dvs = dus + (size_t)(iters + 1) * K;      // [iters+1][P]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = SB_T / 64;
  const int b = blockIdx.x;
  const float norm = -logf((float)(K + P));
  const float inv_eps = 1.0f / eps;
  const float* sc = scores + ((size_t)b * N + 1) * ldl;
  const float* dp = dplan + ((size_t)b * N + 1) * ldl;
  if (ZLDS)
    for (int e = tid; e < K * P; e += SB_T) {
      const int p = e / K, k = e - p * K;          // global reads along k (contiguous)
      z[(size_t)k * PP + p] = sc[(size_t)p * ldl + k] * inv_eps;
    }
#define Z_(k, p) (ZLDS ? z[(size_t)(k) * PP + (p)] : sc[(size_t)(p) * ldl + (k)] * inv_eps)
  for (int e = tid; e < K; e += SB_T) u[e] = 0.f;
  for (int e = tid; e < P; e += SB_T) v[e] = 0.f;
  __syncthreads();
  for (int t = 1; t <= iters; ++t) {
    for (int k = wave; k < K; k += nw) {           // u_t[k] = norm - LSE_p(z[k,p] + v_{t-1}[p]): one wave per row
      float mx = -INFINITY;
      for (int p = lane; p < P; p += 64) mx = fmaxf(mx, Z_(k, p) + v[(t - 1) * P + p]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      float s = 0.f;
      for (int p = lane; p < P; p += 64) s += expf(Z_(k, p) + v[(t - 1) * P + p] - mx);
      s = wave_sum(s);
      if (lane == 0) u[t * K + k] = norm - (mx + logf(s));
    }
    __syncthreads();
    for (int p = tid; p < P; p += SB_T) {          // v_t[p] = norm - LSE_k(z[k,p] + u_t[k]): one thread per column
      float mx = -INFINITY;
      for (int k = 0; k < K; ++k) mx = fmaxf(mx, Z_(k, p) + u[t * K + k]);
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += expf(Z_(k, p) + u[t * K + k] - mx);
      v[t * P + p] = norm - (mx + logf(s));
    }
    __syncthreads();
  }
  // dZf = dplan * plan, plan = exp(z + u_T + v_T - norm); du_T, dv_T
  for (int k = wave; k < K; k += nw) {
    float s = 0.f;
    for (int p = lane; p < P; p += 64)
      s += dp[(size_t)p * ldl + k] * expf(Z_(k, p) + u[iters * K + k] + v[iters * P + p] - norm);
    s = wave_sum(s);
    if (lane == 0) du[k] = s;
  }
  for (int p = tid; p < P; p += SB_T) {
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += dp[(size_t)p * ldl + k] * expf(Z_(k, p) + u[iters * K + k] + v[iters * P + p] - norm);
    dv[p] = s;
  }
  __syncthreads();
  for (int t = iters; t >= 1; --t) {
    for (int p = tid; p < P; p += SB_T) dvs[t * P + p] = dv[p];
    // du_t[k] -= sum_p dv_t[p] A_t[k,p]
    for (int k = wave; k < K; k += nw) {
      float s = 0.f;
      for (int p = lane; p < P; p += 64) s += dv[p] * expf(Z_(k, p) + u[t * K + k] + v[t * P + p] - norm);
      s = wave_sum(s);
      if (lane == 0) {
        const float d = du[k] - s;
        du[k] = d;
        dus[t * K + k] = d;
      }
    }
    __syncthreads();
    // dv_{t-1}[p] = - sum_k du_t[k] B_t[k,p];  u_{t-1} receives nothing else (u_t does not depend on u_{t-1})
    for (int p = tid; p < P; p += SB_T) {
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += du[k] * expf(Z_(k, p) + u[t * K + k] + v[(t - 1) * P + p] - norm);
      dv[p] = -s;
    }
    __syncthreads();
    for (int k = tid; k < K; k += SB_T) du[k] = 0.f;
    __syncthreads();
  }
  // dscores = (dZf - sum_t (dv_t A_t + du_t B_t)) / eps, written token-major (contiguous along k), CLS row zero
  uint16_t* o = ds + ((size_t)b * N + 1) * ldo;
  for (int e = tid; e < K * P; e += SB_T) {
    const int p = e / K, k = e - p * K;
    const float zz = Z_(k, p);
    float g = dp[(size_t)p * ldl + k] * expf(zz + u[iters * K + k] + v[iters * P + p] - norm);
    for (int t = 1; t <= iters; ++t) {
      const float base = zz + u[t * K + k] - norm;
      g -= dvs[t * P + p] * expf(base + v[t * P + p]) + dus[t * K + k] * expf(base + v[(t - 1) * P + p]);
    }
    o[(size_t)p * ldo + k] = (uint16_t)(pack_bf16x2(g * inv_eps, 0.f) & 0xffffu);
  }
  for (int k = tid; k < K; k += SB_T) ds[(size_t)b * N * ldo + k] = 0;
#undef Z_
}

inline size_t sinkhorn_bwd_lds(int N, int K, int iters, bool zlds) {
  const size_t P = N - 1, PP = P | 1;
  return ((zlds ? (size_t)K * PP : 0) + (size_t)2 * (iters + 1) * (K + P) + K + P) * sizeof(float);
}

}  // namespace

extern "C" int tr_soft_dweights(const float* g, const float* src, float* dwt, int ldl, int B, int N, int K, int D, tr_stream_t s) {
  TR_REQUIRE(g && src && dwt, TR_ERR_NULL, "tr_soft_dweights: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && K >= 1 && D > 0 && ldl >= K, TR_ERR_SHAPE, "tr_soft_dweights: bad shape");
  const int P = N - 1;
  tr_prof_note("soft_dweights", 2.0 * B * P * K * D, 4.0 * B * ((double)P * D + (double)K * D + (double)P * K));
  // dW[p][k] = <src[1+p], g[1+k]>:  A = src rows [P, D], Bm = g rows [K, D] (NT)
  const dim3 grid(((P + GT - 1) / GT) * ((K + GT - 1) / GT), B);
  hipLaunchKernelGGL(bgemm_f32_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(s), src + D, (long)N * D, D, g + D, (long)(K + 1) * D, D,
                     dwt + ldl, (long)N * ldl, ldl, P, K, D);
  TR_CHECK_LAUNCH("tr_soft_dweights");
  return TR_OK;
}

extern "C" int tr_soft_dsrc(const float* g, const float* wt, int ldl, float* dsrc, int B, int N, int K, int D, tr_stream_t s) {
  TR_REQUIRE(g && wt && dsrc, TR_ERR_NULL, "tr_soft_dsrc: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && K >= 1 && D > 0 && ldl >= K, TR_ERR_SHAPE, "tr_soft_dsrc: bad shape");
  const int P = N - 1;
  tr_prof_note("soft_dsrc", 2.0 * B * P * K * D, 4.0 * B * ((double)P * D + (double)K * D + (double)P * K));
  // dsrc[1+p][d] = sum_k W[1+p][k] g[1+k][d]:  A = W rows [P, K], Bm = g rows [K, D] (NN)
  const dim3 grid(((P + GT - 1) / GT) * ((D + GT - 1) / GT), B);
  hipLaunchKernelGGL(bgemm_f32_kernel<true>, grid, dim3(256), 0, static_cast<hipStream_t>(s), wt + ldl, (long)N * ldl, ldl, g + D, (long)(K + 1) * D, D,
                     dsrc + D, (long)N * D, D, P, D, K);
  TR_CHECK_LAUNCH("tr_soft_dsrc");
  return TR_OK;
}

extern "C" size_t tr_token_softmax_bwd_workspace_floats(int B, int K) { return (size_t)B * ((K + 31) / 32); }

extern "C" int tr_token_softmax_bwd(const float* wt, const float* dwt, const float* logits, int ldl, float scale, uint16_t* ds, int ldo,
                                    float* dscale, int accumulate, float* ws, size_t ws_floats, int B, int N, int K, tr_stream_t s) {
  TR_REQUIRE(wt && dwt && ds, TR_ERR_NULL, "tr_token_softmax_bwd: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && K >= 1 && ldl >= K && ldo >= K, TR_ERR_SHAPE, "tr_token_softmax_bwd: bad shape");
  TR_REQUIRE(dscale == nullptr || (logits && ws && ws_floats >= tr_token_softmax_bwd_workspace_floats(B, K)), TR_ERR_SHAPE,
             "tr_token_softmax_bwd: d scale needs the raw logits and %zu floats of workspace", tr_token_softmax_bwd_workspace_floats(B, K));
  const dim3 grid((K + 31) / 32, B);
  hipLaunchKernelGGL(token_softmax_bwd_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(s), wt, dwt, dscale ? logits : nullptr, ldl, scale, ds,
                     ldo, dscale ? ws : nullptr, N, K);
  TR_CHECK_LAUNCH("tr_token_softmax_bwd");
  if (dscale) return tr_reduce_partials_f32(ws, (int)tr_token_softmax_bwd_workspace_floats(B, K), 1, dscale, accumulate, s);
  return TR_OK;
}

extern "C" int tr_add_into_bf16(const float* a, uint16_t* y, size_t n, tr_stream_t s) {
  TR_REQUIRE(a && y, TR_ERR_NULL, "tr_add_into_bf16: null pointer");
  TR_REQUIRE(tr_aligned16(a) && (reinterpret_cast<uintptr_t>(y) & 7u) == 0, TR_ERR_ALIGN, "tr_add_into_bf16: pointers must be aligned");
  if (n == 0) return TR_OK;
  hipLaunchKernelGGL(add_into_bf16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, static_cast<hipStream_t>(s), a, y, n);
  TR_CHECK_LAUNCH("tr_add_into_bf16");
  return TR_OK;
}

extern "C" int tr_rownorm_bwd(const float* x, const float* da, const uint16_t* db, float* dx, int M, int D, tr_stream_t s) {
  TR_REQUIRE(x && da && dx, TR_ERR_NULL, "tr_rownorm_bwd: null pointer");
  TR_REQUIRE(M > 0 && D > 0, TR_ERR_SHAPE, "tr_rownorm_bwd: bad shape");
  hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(s), x, da, db, dx, M, D);
  TR_CHECK_LAUNCH("tr_rownorm_bwd");
  return TR_OK;
}

extern "C" int tr_sinkhorn_bwd(const float* scores, const float* dplan, int ldl, float eps, int iters, uint16_t* ds, int ldo, int B, int N,
                               int K, tr_stream_t s) {
  TR_REQUIRE(scores && dplan && ds, TR_ERR_NULL, "tr_sinkhorn_bwd: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && K >= 1 && ldl >= K && ldo >= K && iters >= 1 && iters <= SB_MAXIT && eps > 0.f, TR_ERR_SHAPE,
             "tr_sinkhorn_bwd: bad shape / iters (1..%d)", SB_MAXIT);
  const bool zlds = sinkhorn_bwd_lds(N, K, iters, true) <= 160 * 1024;      // else the score matrix is re-read from global memory (L2)
  const size_t lds = sinkhorn_bwd_lds(N, K, iters, zlds);
  TR_REQUIRE(lds <= 160 * 1024, TR_ERR_SHAPE, "tr_sinkhorn_bwd: K=%d, P=%d, %d iterations need %zu B of LDS for the scaling vectors alone", K, N - 1,
             iters, lds);
  if (zlds) {
    TR_RESERVE_LDS(reinterpret_cast<const void*>(sinkhorn_bwd_kernel<true>), lds, "tr_sinkhorn_bwd");
    hipLaunchKernelGGL(sinkhorn_bwd_kernel<true>, dim3(B), dim3(SB_T), lds, static_cast<hipStream_t>(s), scores, dplan, ldl, eps, iters, ds, ldo, N, K);
  } else {
    TR_RESERVE_LDS(reinterpret_cast<const void*>(sinkhorn_bwd_kernel<false>), lds, "tr_sinkhorn_bwd");
    hipLaunchKernelGGL(sinkhorn_bwd_kernel<false>, dim3(B), dim3(SB_T), lds, static_cast<hipStream_t>(s), scores, dplan, ldl, eps, iters, ds, ldo, N, K);
  }
  TR_CHECK_LAUNCH("tr_sinkhorn_bwd");
  return TR_OK;
}
