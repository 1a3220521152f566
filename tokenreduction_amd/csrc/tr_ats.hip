// Adaptive Token Sampling (models/ats.py) on gfx950.
//
//   tr_ats_sample   AdaptiveTokenSampling.forward ats.py:52-84 for one block: significance score = sum_h cls_attn_h * |v_h|
//                   over the patch tokens, normalised; running sum (cdf), +0.1 on masked positions; for every point of the
//                   inverse-CDF grid the nearest cdf entry; per-image sorted unique; CLS id 0 in front.  The reference pads
//                   to the batch maximum (data dependent shape, ats.py:78); here every image is padded with id 0 to the static
//                   bound K -- padded rows are masked keys with exactly zero softmax weight, so valid rows are unchanged.
//                   One workgroup per image; everything after the |v| pass lives in LDS.
//   tr_ats_gather   x = batched_index_select(x, ids) (ats.py:157) and new_attn @ v = rows `ids` of attn @ v (ats.py:86,129):
//                   row gather of the fp32 residual stream and of the attention output.
// The nearest-entry search mirrors torch.cdist's matmul form (|a|^2 + |b|^2 - 2ab as a 3-term dot product, clamp 1e-30, sqrt):
// its fp32 rounding -- not the exact |a - b| -- decides among candidates closer than ~3e-4 to a grid point.
#include "tr_common.h"

namespace {

constexpr int ATS_MAX_P = 1024;

template <bool F32>
__global__ __launch_bounds__(256) void ats_sample_kernel(const float* __restrict__ cls_rows, const void* __restrict__ qkv,
                                                         const float* __restrict__ mask, const float* __restrict__ steps,
                                                         int n_steps, int32_t* __restrict__ ids, float* __restrict__ new_mask,
                                                         float* __restrict__ cdf_out, int N, int H, int K, float eps) {
  __shared__ float s_sig[ATS_MAX_P];
  __shared__ float s_cdf[ATS_MAX_P];
  __shared__ int s_flag[ATS_MAX_P + 1];
  __shared__ float s_total;
  extern __shared__ float s_part[];                   // [H][P]: one head's term of every token
  const int b = blockIdx.x, tid = threadIdx.x;
  const int P = N - 1;
  const int ldq = 3 * H * 64;
  // sig[p] = sum_h attn[b,h,0,1+p] * ||v[b,h,1+p,:]||_2     (heads summed in order, ats.py:58-63).  One (token, head) item per thread and
  // step -- the per-item arithmetic and the head order of the sum are those of the rolled loop over heads this replaces, which waited
  // for H dependent global round trips per token (round 4: 27-31 us per launch, half of it here).
  for (int item = tid; item < P * H; item += 256) {
    const int h = item / P, p = item - h * P;
    const size_t e0 = ((size_t)b * N + 1 + p) * ldq + 2 * H * 64 + h * 64;
    float ss = 0.f;
    if (F32) {
      const float4* vp = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(qkv) + e0);
      float4 u[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) u[c] = vp[c];
#pragma unroll
      for (int c = 0; c < 16; ++c) { ss = fmaf(u[c].x, u[c].x, ss); ss = fmaf(u[c].y, u[c].y, ss); ss = fmaf(u[c].z, u[c].z, ss); ss = fmaf(u[c].w, u[c].w, ss); }
    } else {
      const uint4* vp = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(qkv) + e0);
      uint4 u[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) u[c] = vp[c];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const unsigned int w4[4] = {u[c].x, u[c].y, u[c].z, u[c].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = __uint_as_float(w4[q] << 16), hi = __uint_as_float(w4[q] & 0xffff0000u);
          ss = fmaf(lo, lo, ss);
          ss = fmaf(hi, hi, ss);
        }
      }
    }
    s_part[item] = cls_rows[((size_t)b * H + h) * N + 1 + p] * sqrtf(ss);
  }
  __syncthreads();
  for (int p = tid; p < P; p += 256) {
    float sig = 0.f;
    for (int h = 0; h < H; ++h) sig += s_part[h * P + p];
    s_sig[p] = sig;
    s_flag[p + 1] = 0;
  }
  if (tid == 0) s_flag[0] = 0;
  __syncthreads();
  if (tid < 64) {                                    // sig.sum(-1)
    float t = 0.f;
    for (int p = tid; p < P; p += 64) t += s_sig[p];
    t = wave_sum(t);
    if (tid == 0) s_total = t + eps;
  }
  __syncthreads();
  const float denom = s_total;
  for (int p = tid; p < P; p += 256) s_cdf[p] = s_sig[p] / denom;      // the quotients in parallel; only the running sum is a chain
  __syncthreads();
  if (tid == 0) {                                    // cumsum in token order, then +0.1 on masked positions (ats.py:69-70)
    float run = 0.f;
    for (int p = 0; p < P; ++p) {
      run += s_cdf[p];
      s_cdf[p] = run;
    }
  }
  __syncthreads();
  for (int p = tid; p < P; p += 256) {
    float c = s_cdf[p];
    if (mask != nullptr && mask[(size_t)b * N + 1 + p] == 0.f) c += 0.1f;
    s_cdf[p] = c;
    if (cdf_out != nullptr) cdf_out[(size_t)b * P + p] = c;
  }
  __syncthreads();
  // nearest cdf entry per grid point; torch.cdist matmul form: [-2s, s^2, 1] . [c, 1, c^2], clamp_min(1e-30), sqrt; argmin
  // keeps the first minimum.  Four lanes per grid point, each walking every fourth entry; (distance, index) minima combined
  // lexicographically, which is the first minimum of the whole row.
  {
    const int q4 = tid & 3;
    for (int t = tid >> 2; t < ((n_steps + 63) & ~63); t += 64) {
      const float s = steps[min(t, n_steps - 1)];
      const float a0 = -2.0f * s, a1 = __fmul_rn(s, s);
      float best = INFINITY;
      int arg = 0x7fffffff;
      for (int p = q4; p < P; p += 4) {
        const float c = s_cdf[p];
        // k = 0,1,2 of the 3-term dot product, each product/sum rounded once (no contraction: c^2 is a rounded operand)
        float r = __fmul_rn(a0, c);
        r = __fadd_rn(r, a1);
        r = __fadd_rn(r, __fmul_rn(c, c));
        const float d = sqrtf(fmaxf(r, 1e-30f));
        if (d < best) { best = d; arg = p; }
      }
#pragma unroll
      for (int o = 1; o < 4; o <<= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oa = __shfl_xor(arg, o, 64);
        if (ob < best || (ob == best && oa < arg)) { best = ob; arg = oa; }
      }
      if (arg == 0x7fffffff) arg = 0;                 // every distance NaN: index 0, as a sequential first-minimum scan returns
      if (q4 == 0 && t < n_steps) s_flag[arg + 1] = 1;                             // sampled_token_ids = argmin + 1
    }
  }
  __syncthreads();
  // sorted unique ids, CLS id 0 in front, zero padding to K; new_mask = ids != 0 (CLS True)   (ats.py:77-84)
  for (int t = tid; t < K; t += 256) {
    ids[(size_t)b * K + t] = 0;
    new_mask[(size_t)b * K + t] = t == 0 ? 1.f : 0.f;
  }
  __syncthreads();
  for (int p = tid; p < P; p += 256) {
    if (!s_flag[p + 1]) continue;
    int pos = 1;
    for (int q = 0; q < p; ++q) pos += s_flag[q + 1];
    if (pos < K) {
      ids[(size_t)b * K + pos] = p + 1;
      new_mask[(size_t)b * K + pos] = 1.f;
    }
  }
}

// Dynamic width (ats.py:77-78: pad_sequence pads the unique ids to the BATCH maximum): width = 1 + max over the images of their unique sampled ids
// = max row sum of new_mask; one thread per image, atomicMax into a zeroed word.
__global__ __launch_bounds__(256) void ats_width_kernel(const float* __restrict__ new_mask, int32_t* __restrict__ width, int B, int K) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  int c = 0;
  for (int t = 0; t < K; ++t) c += new_mask[(size_t)b * K + t] != 0.f;
  atomicMax(width, c);
}
// the first Kw columns of ids / new_mask [B,K] as contiguous [B,Kw] arrays
__global__ __launch_bounds__(256) void ats_narrow_kernel(const int32_t* __restrict__ ids, const float* __restrict__ mask,
                                                         int32_t* __restrict__ ids_out, float* __restrict__ mask_out, int B, int K, int Kw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * Kw) return;
  const int b = i / Kw, t = i - b * Kw;
  ids_out[i] = ids[(size_t)b * K + t];
  mask_out[i] = mask[(size_t)b * K + t];
}

// one wave per output row
template <bool F32>
__global__ __launch_bounds__(256) void ats_gather_kernel(const float* __restrict__ x, const void* __restrict__ ao,
                                                         const int32_t* __restrict__ ids, float* __restrict__ x_out,
                                                         void* __restrict__ ao_out, int B, int N, int K, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * K) return;
  const int b = row / K;
  const size_t src = (size_t)b * N + ids[row];
  // every load of the row pair goes out before the first store (rolled copy loops waited for each 16-byte chunk in turn);
  // D <= 1024: at most 4 fp32 chunks and 2 (bf16) / 4 (fp32) chunks of the attention output per lane
  const float4* xs = reinterpret_cast<const float4*>(x + src * D);
  float4* xd = reinterpret_cast<float4*>(x_out + (size_t)row * D);
  const int n4 = D / 4;
  float4 xv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) xv[c] = xs[min(lane + 64 * c, n4 - 1)];
  if (F32) {
    const float4* as = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ao) + src * D);
    float4* ad = reinterpret_cast<float4*>(reinterpret_cast<float*>(ao_out) + (size_t)row * D);
    float4 av[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) av[c] = as[min(lane + 64 * c, n4 - 1)];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (lane + 64 * c < n4) { xd[lane + 64 * c] = xv[c]; ad[lane + 64 * c] = av[c]; }
  } else {
    const uint4* as = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(ao) + src * D);
    uint4* ad = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(ao_out) + (size_t)row * D);
    const int n8 = D / 8;
    uint4 av[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) av[c] = as[min(lane + 64 * c, n8 - 1)];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (lane + 64 * c < n4) xd[lane + 64 * c] = xv[c];
#pragma unroll
    for (int c = 0; c < 2; ++c)
      if (lane + 64 * c < n8) ad[lane + 64 * c] = av[c];
  }
}

}  // namespace

extern "C" int tr_ats_sample(const float* cls_rows, const void* qkv, int qkv_is_f32, const float* mask, const float* steps,
                             int n_steps, int32_t* ids, float* new_mask, float* cdf_out, int B, int N, int H, int K,
                             tr_stream_t s) {
  TR_REQUIRE(cls_rows && qkv && steps && ids && new_mask, TR_ERR_NULL, "tr_ats_sample: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 2 && N - 1 <= ATS_MAX_P, TR_ERR_SHAPE, "tr_ats_sample: need 2 <= N <= %d (N=%d)", ATS_MAX_P + 1, N);
  TR_REQUIRE(K >= 2 && n_steps >= 1 && n_steps <= K - 1, TR_ERR_SHAPE,
             "tr_ats_sample: %d grid points cannot exceed K-1 = %d (at most one new token per point, ats.py:48)", n_steps, K - 1);
  TR_REQUIRE(tr_aligned16(qkv), TR_ERR_ALIGN, "tr_ats_sample: qkv must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  const size_t lds = (size_t)H * (N - 1) * sizeof(float);            // one term per (head, token)
  TR_REQUIRE(lds <= 128 * 1024, TR_ERR_SHAPE, "tr_ats_sample: %d heads x %d tokens do not fit the per-head scratch (128 KiB of LDS)", H, N - 1);
  // more than the default 64 KiB of dynamic LDS (16 heads x 1024 tokens = 64 KiB; the CU has 160): raise the kernel's limit on demand
  if (qkv_is_f32) TR_RESERVE_LDS(reinterpret_cast<const void*>(ats_sample_kernel<true>), lds, "tr_ats_sample");
  else TR_RESERVE_LDS(reinterpret_cast<const void*>(ats_sample_kernel<false>), lds, "tr_ats_sample");
  if (qkv_is_f32)
    hipLaunchKernelGGL(ats_sample_kernel<true>, dim3(B), dim3(256), lds, st, cls_rows, qkv, mask, steps, n_steps, ids, new_mask, cdf_out, N,
                       H, K, 1e-6f);
  else
    hipLaunchKernelGGL(ats_sample_kernel<false>, dim3(B), dim3(256), lds, st, cls_rows, qkv, mask, steps, n_steps, ids, new_mask, cdf_out,
                       N, H, K, 1e-6f);
  TR_CHECK_LAUNCH("tr_ats_sample");
  return TR_OK;
}

extern "C" int tr_ats_gather(const float* x, const void* ao, int ao_is_f32, const int32_t* ids, float* x_out, void* ao_out, int B,
                             int N, int K, int D, tr_stream_t s) {
  TR_REQUIRE(x && ao && ids && x_out && ao_out, TR_ERR_NULL, "tr_ats_gather: null pointer");
  TR_REQUIRE(B > 0 && N >= 1 && K >= 1 && D > 0 && D % 8 == 0 && D <= 1024, TR_ERR_SHAPE, "tr_ats_gather: bad shape B=%d N=%d K=%d D=%d (D %% 8 == 0, D <= 1024)", B, N, K, D);
  TR_REQUIRE(x_out != x && ao_out != ao, TR_ERR_SHAPE, "tr_ats_gather: needs distinct outputs");
  TR_REQUIRE(tr_aligned16(x) && tr_aligned16(ao) && tr_aligned16(x_out) && tr_aligned16(ao_out), TR_ERR_ALIGN,
             "tr_ats_gather: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  const int nblocks = (B * K + 3) / 4;
  if (ao_is_f32) hipLaunchKernelGGL(ats_gather_kernel<true>, dim3(nblocks), dim3(256), 0, st, x, ao, ids, x_out, ao_out, B, N, K, D);
  else hipLaunchKernelGGL(ats_gather_kernel<false>, dim3(nblocks), dim3(256), 0, st, x, ao, ids, x_out, ao_out, B, N, K, D);
  TR_CHECK_LAUNCH("tr_ats_gather");
  return TR_OK;
}

extern "C" int tr_ats_width(const float* new_mask, int32_t* width, int B, int K, tr_stream_t s) {
  TR_REQUIRE(new_mask && width, TR_ERR_NULL, "tr_ats_width: null pointer");
  TR_REQUIRE(B > 0 && K >= 1, TR_ERR_SHAPE, "tr_ats_width: bad shape B=%d K=%d", B, K);
  hipStream_t st = static_cast<hipStream_t>(s);
  hipError_t e = hipMemsetAsync(width, 0, sizeof(int32_t), st);
  TR_REQUIRE(e == hipSuccess, TR_ERR_LAUNCH, "tr_ats_width: hipMemsetAsync: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(ats_width_kernel, dim3((B + 255) / 256), dim3(256), 0, st, new_mask, width, B, K);
  TR_CHECK_LAUNCH("tr_ats_width");
  return TR_OK;
}

extern "C" int tr_ats_narrow(const int32_t* ids, const float* new_mask, int32_t* ids_out, float* mask_out, int B, int K, int Kw, tr_stream_t s) {
  TR_REQUIRE(ids && new_mask && ids_out && mask_out, TR_ERR_NULL, "tr_ats_narrow: null pointer");
  TR_REQUIRE(B > 0 && K >= 1 && Kw >= 1 && Kw <= K, TR_ERR_SHAPE, "tr_ats_narrow: bad shape B=%d K=%d Kw=%d", B, K, Kw);
  TR_REQUIRE(static_cast<const void*>(ids) != static_cast<const void*>(ids_out) && new_mask != mask_out, TR_ERR_SHAPE,
             "tr_ats_narrow: needs distinct outputs");
  hipLaunchKernelGGL(ats_narrow_kernel, dim3((B * Kw + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(s), ids, new_mask, ids_out, mask_out, B, K,
                     Kw);
  TR_CHECK_LAUNCH("tr_ats_narrow");
  return TR_OK;
}
