// fp32 VALIDATION path: the same ops as the bf16 fast path, in the reference's own arithmetic (fp32 operands, fp32
// accumulate, exact-erf GELU, expf softmax), so that the executor can be run end-to-end against the fp32 golden vectors
// of the reference: Top-K / EViT indices BIT-EXACT, logits to ~1e-4.  A bf16 pipeline cannot show that -- the synthetic
// golden models have a K-th/K+1-th score gap of ~1e-3 relative, below bf16 resolution.
//
// Correctness first, speed second (it runs DeiT-S at a few hundred images/s): plain LDS-tiled VALU SGEMM and a
// wave-per-query attention.  Everything still runs on the GPU through the C ABI -- this is not a CPU fallback.
//   tr_gemm_f32        nn.Linear:  topk.py:44,52, timm Mlp fc1/fc2, head topk.py:203, PatchEmbed topk.py:181
//   tr_attention_f32   softmax(q k^T * 64^-0.5) v + CLS row   topk.py:44-51,59
#include "tr_common.h"

namespace {

constexpr int FT = 64, FK = 16;   // 64x64 output tile, 16-deep K slabs, 256 threads x (4x4) outputs

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       const float* __restrict__ aux, int aux_i, int M, int N, int K) {
  __shared__ float sA[FK][FT + 4];
  __shared__ float sW[FK][FT + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int nNt = (N + FT - 1) / FT;
  const int m0 = (blockIdx.x / nNt) * FT, n0 = (blockIdx.x % nNt) * FT;
  const int lr = tid >> 2, lk = (tid & 3) * 4;             // staging: row lr, 4 consecutive k
  const float* ap = A + (size_t)min(m0 + lr, M - 1) * K + lk;
  const float* wp = W + (size_t)min(n0 + lr, N - 1) * K + lk;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += FK) {
    const float4 a = *reinterpret_cast<const float4*>(ap + k0);
    const float4 w = *reinterpret_cast<const float4*>(wp + k0);
    __syncthreads();
    sA[lk + 0][lr] = a.x; sA[lk + 1][lr] = a.y; sA[lk + 2][lr] = a.z; sA[lk + 3][lr] = a.w;
    sW[lk + 0][lr] = w.x; sW[lk + 1][lr] = w.y; sW[lk + 2][lr] = w.z; sW[lk + 3][lr] = w.w;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FK; ++k) {
      const float4 av = *reinterpret_cast<const float4*>(&sA[k][ty * 4]);
      const float4 wv = *reinterpret_cast<const float4*>(&sW[k][tx * 4]);
      const float a4[4] = {av.x, av.y, av.z, av.w}, w4[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], w4[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
    size_t orow = (size_t)m;
    const float* posrow = nullptr;
    if (EPI == TR_EPI_PATCH_F32) {
      const int b = m / aux_i, p = m - b * aux_i;
      orow = (size_t)b * (aux_i + 1) + 1 + p;
      posrow = aux + (size_t)(1 + p) * N;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float v = acc[i][j] + bias[n];
      if (EPI == TR_EPI_GELU_BF16) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));   // nn.GELU(): exact erf
      if (EPI == TR_EPI_PATCH_F32) v += posrow[n];
      out[orow * N + n] = v;
    }
  }
}

// One workgroup per (image, head); one wave per query row at a time.  K is kept TRANSPOSED in LDS (Kt[d][key]) so the lanes
// of a wave (= keys) read consecutive addresses; V row-major (lane = d for the P.V product).
template <bool POLICY>
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            float* __restrict__ cls_rows, const float* __restrict__ size,
                                                            float* __restrict__ colsum_part, int N, int H) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NP = (N + 63) & ~63;
  float* sKt = sm;                       // [64][NP]
  float* sV = sKt + 64 * NP;             // [N][64]
  float* sQ = sV + (size_t)NP * 64;      // [4 waves][64]
  float* sP = sQ + 4 * 64;               // [4 waves][NP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int ldq = 3 * H * 64;
  const float* base = qkv + (size_t)b * N * ldq;
  const int qcol = h * 64, kcol = H * 64 + h * 64, vcol = 2 * H * 64 + h * 64;
  for (int e = tid; e < NP * 64; e += 256) {
    const int key = e >> 6, d = e & 63;
    const float kv = key < N ? base[(size_t)key * ldq + kcol + d] : 0.f;
    const float vv = key < N ? base[(size_t)key * ldq + vcol + d] : 0.f;
    sKt[d * NP + key] = kv;
    sV[key * 64 + d] = vv;
  }
  __syncthreads();
  const int nkc = NP >> 6;               // key chunks of 64 per lane
  float colacc[4] = {0.f, 0.f, 0.f, 0.f}; // column sums of the softmax matrix over this wave's queries (kmedoids.py:240)
  for (int q = wave; q < N; q += 4) {
    sQ[wave * 64 + lane] = base[(size_t)q * ldq + qcol + lane];
    __builtin_amdgcn_wave_barrier();
    float s[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < nkc) {
        const int key = c * 64 + lane;
        float acc = 0.f;
#pragma unroll 16
        for (int d = 0; d < 64; ++d) acc = fmaf(sQ[wave * 64 + d], sKt[d * NP + key], acc);
        // (q @ k^T) * scale [+ size.log(), tome.py:48-49], scale = 64^-0.5
        acc = key < N ? acc * 0.125f + ((size && !POLICY) ? logf(size[(size_t)b * N + key]) : 0.f) : -INFINITY;
        s[c] = acc;
        mx = fmaxf(mx, acc);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float l = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s[c] = expf(s[c] - mx);           // exp(-inf) = 0 for padded keys / unused chunks
      if (POLICY) {                     // softmax_with_policy dyvit.py:39-51: exp * (policy + (1 - policy) * eye)
        const int key = c * 64 + lane;
        if (c < nkc && key < N && key != q) s[c] *= size[(size_t)b * N + key];
      }
      l += s[c];
    }
    l = wave_sum(l);
    const float inv = POLICY ? 1.0f / (l + 1e-6f) : 1.0f / l;
    if (POLICY) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nkc && c * 64 + lane < N) s[c] += 1e-6f / (float)N;     // (attn + eps/N) / (sum + eps)
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < nkc) {
        sP[wave * NP + c * 64 + lane] = s[c] * inv;                // softmax row (attn = attn.softmax(-1))
        colacc[c] += s[c] * inv;
      }
    __builtin_amdgcn_wave_barrier();
    float o = 0.f;
    for (int key = 0; key < N; ++key) o = fmaf(sP[wave * NP + key], sV[key * 64 + lane], o);
    out[((size_t)b * N + q) * (H * 64) + h * 64 + lane] = o;
    if (cls_rows != nullptr && q == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int key = c * 64 + lane;
        if (c < nkc && key < N) cls_rows[((size_t)b * H + h) * N + key] = s[c] * inv;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (colsum_part != nullptr) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int key = c * 64 + lane;
      if (c < nkc && key < N) colsum_part[(((size_t)b * H + h) * 4 + wave) * N + key] = colacc[c];
    }
  }
}

// Long sequences (256 < N <= 640: the 384^2 inputs, N = 577): K^T and V of a head no longer fit the LDS as fp32 (2 x 164 KB), so
// the same wave-per-query scheme reads K rows and V rows straight from global memory (they stay in L2: 2 x 148 KB per head).
// Only the softmax row lives in LDS.  Same arithmetic and summation order as attention_f32_kernel.
constexpr int F32_LONG_CHUNKS = 10;
__global__ __launch_bounds__(256) void attention_f32_long_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 float* __restrict__ cls_rows, const float* __restrict__ size,
                                                                 float* __restrict__ colsum_part, int N, int H) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int NP = (N + 63) & ~63;
  float* sQ = sm;                        // [4 waves][64]
  float* sP = sQ + 4 * 64;               // [4 waves][NP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int ldq = 3 * H * 64;
  const float* base = qkv + (size_t)b * N * ldq;
  const int qcol = h * 64, kcol = H * 64 + h * 64, vcol = 2 * H * 64 + h * 64;
  const int nkc = NP >> 6;
  float colacc[F32_LONG_CHUNKS];
#pragma unroll
  for (int c = 0; c < F32_LONG_CHUNKS; ++c) colacc[c] = 0.f;
  for (int q = wave; q < N; q += 4) {
    sQ[wave * 64 + lane] = base[(size_t)q * ldq + qcol + lane];
    __builtin_amdgcn_wave_barrier();
    float s[F32_LONG_CHUNKS];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < F32_LONG_CHUNKS; ++c) {
      s[c] = -INFINITY;
      if (c < nkc) {
        const int key = c * 64 + lane;
        const float* krow = base + (size_t)min(key, N - 1) * ldq + kcol;
        float acc = 0.f;
#pragma unroll 16
        for (int d = 0; d < 64; ++d) acc = fmaf(sQ[wave * 64 + d], krow[d], acc);
        acc = key < N ? acc * 0.125f + (size ? logf(size[(size_t)b * N + key]) : 0.f) : -INFINITY;
        s[c] = acc;
        mx = fmaxf(mx, acc);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float l = 0.f;
#pragma unroll
    for (int c = 0; c < F32_LONG_CHUNKS; ++c) {
      s[c] = expf(s[c] - mx);
      l += s[c];
    }
    l = wave_sum(l);
    const float inv = 1.0f / l;
#pragma unroll
    for (int c = 0; c < F32_LONG_CHUNKS; ++c)
      if (c < nkc) {
        sP[wave * NP + c * 64 + lane] = s[c] * inv;
        colacc[c] += s[c] * inv;
      }
    __builtin_amdgcn_wave_barrier();
    float o = 0.f;
    for (int key = 0; key < N; ++key) o = fmaf(sP[wave * NP + key], base[(size_t)key * ldq + vcol + lane], o);
    out[((size_t)b * N + q) * (H * 64) + h * 64 + lane] = o;
    if (cls_rows != nullptr && q == 0) {
#pragma unroll
      for (int c = 0; c < F32_LONG_CHUNKS; ++c) {
        const int key = c * 64 + lane;
        if (c < nkc && key < N) cls_rows[((size_t)b * H + h) * N + key] = s[c] * inv;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (colsum_part != nullptr) {
#pragma unroll
    for (int c = 0; c < F32_LONG_CHUNKS; ++c) {
      const int key = c * 64 + lane;
      if (c < nkc && key < N) colsum_part[(((size_t)b * H + h) * 4 + wave) * N + key] = colacc[c];
    }
  }
}

}  // namespace

extern "C" int tr_gemm_f32(const float* A, const float* W, const float* bias, float* out, const float* aux, int aux_i, int M,
                           int N, int K, int epilogue, tr_stream_t s) {
  TR_REQUIRE(A && W && bias && out, TR_ERR_NULL, "tr_gemm_f32: null pointer");
  TR_REQUIRE(M > 0 && N > 0 && K > 0 && K % FK == 0, TR_ERR_SHAPE, "tr_gemm_f32: need K %% %d == 0 (M=%d N=%d K=%d)", FK, M, N, K);
  TR_REQUIRE(tr_aligned16(A) && tr_aligned16(W), TR_ERR_ALIGN, "tr_gemm_f32: operands must be 16-byte aligned");
  if (epilogue == TR_EPI_PATCH_F32)
    TR_REQUIRE(aux && aux_i > 0 && M % aux_i == 0, TR_ERR_SHAPE, "tr_gemm_f32: PATCH epilogue needs pos_embed and P | M");
  const int nblocks = ((M + FT - 1) / FT) * ((N + FT - 1) / FT);
  hipStream_t st = static_cast<hipStream_t>(s);
  switch (epilogue) {
    case TR_EPI_F32: hipLaunchKernelGGL(gemm_f32_kernel<TR_EPI_F32>, dim3(nblocks), dim3(256), 0, st, A, W, bias, out, aux, aux_i, M, N, K); break;
    case TR_EPI_GELU_BF16: hipLaunchKernelGGL(gemm_f32_kernel<TR_EPI_GELU_BF16>, dim3(nblocks), dim3(256), 0, st, A, W, bias, out, aux, aux_i, M, N, K); break;
    case TR_EPI_PATCH_F32: hipLaunchKernelGGL(gemm_f32_kernel<TR_EPI_PATCH_F32>, dim3(nblocks), dim3(256), 0, st, A, W, bias, out, aux, aux_i, M, N, K); break;
    default: TR_REQUIRE(false, TR_ERR_SHAPE, "tr_gemm_f32: epilogue must be TR_EPI_F32, TR_EPI_GELU_BF16 (= GELU, fp32 out) or TR_EPI_PATCH_F32");
  }
  TR_CHECK_LAUNCH("tr_gemm_f32");
  return TR_OK;
}

extern "C" int tr_attention_f32(const float* qkv, float* out, float* cls_rows, const float* size, float* colsum_part, int B, int N,
                                int H, tr_stream_t s) {
  TR_REQUIRE(qkv && out, TR_ERR_NULL, "tr_attention_f32: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 1 && N <= 64 * F32_LONG_CHUNKS, TR_ERR_SHAPE, "tr_attention_f32: need 1 <= N <= %d (N=%d)", 64 * F32_LONG_CHUNKS, N);
  const int NP = (N + 63) & ~63;
  hipStream_t st = static_cast<hipStream_t>(s);
  if (N > 256) {
    const size_t lds_long = (size_t)(4 * 64 + 4 * NP) * sizeof(float);
    hipLaunchKernelGGL(attention_f32_long_kernel, dim3(B * H), dim3(256), lds_long, st, qkv, out, cls_rows, size, colsum_part, N, H);
    TR_CHECK_LAUNCH("tr_attention_f32");
    return TR_OK;
  }
  const size_t lds = (size_t)(64 * NP + NP * 64 + 4 * 64 + 4 * NP) * sizeof(float);
  TR_RESERVE_LDS(reinterpret_cast<const void*>(attention_f32_kernel<false>), lds, "tr_attention_f32");
  hipLaunchKernelGGL(attention_f32_kernel<false>, dim3(B * H), dim3(256), lds, st, qkv, out, cls_rows, size, colsum_part, N, H);
  TR_CHECK_LAUNCH("tr_attention_f32");
  return TR_OK;
}

extern "C" int tr_attention_policy_f32(const float* qkv, float* out, const float* policy, int B, int N, int H, tr_stream_t s) {
  TR_REQUIRE(qkv && out && policy, TR_ERR_NULL, "tr_attention_policy_f32: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 1 && N <= 256, TR_ERR_SHAPE, "tr_attention_policy_f32: need 1 <= N <= 256 (N=%d)", N);
  const int NP = (N + 63) & ~63;
  const size_t lds = (size_t)(64 * NP + NP * 64 + 4 * 64 + 4 * NP) * sizeof(float);
  hipStream_t st = static_cast<hipStream_t>(s);
  TR_RESERVE_LDS(reinterpret_cast<const void*>(attention_f32_kernel<true>), lds, "tr_attention_policy_f32");
  hipLaunchKernelGGL(attention_f32_kernel<true>, dim3(B * H), dim3(256), lds, st, qkv, out, static_cast<float*>(nullptr), policy,
                     static_cast<float*>(nullptr), N, H);
  TR_CHECK_LAUNCH("tr_attention_policy_f32");
  return TR_OK;
}
