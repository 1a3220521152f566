// ToMe -- token merging (models/tome.py) on gfx950: bipartite soft matching + size-weighted merge.
//
//   tr_tome_match             bipartite_soft_matching (tome.py:230-277, class_token=True) on metric = k.mean(1) (tome.py:58):
//                             cosine scores between even- and odd-position tokens, row max/argmax, descending rank of the
//                             row maxima, split into merged (src -> dst) and unmerged tokens.  One workgroup per image; the
//                             whole problem (<= 113 x 112 x 64 MACs) lives in LDS.  Integer outputs; ties: row argmax ->
//                             first index (torch CPU max), rank -> lowest index first (torch's argsort order is unspecified).
//   tr_tome_merge_layernorm   merge_wavg (tome.py:309-323): x = merge(x*size) / merge(size), size = merge(size), with the
//                             pending residual add (x + attn.proj output, tome.py:84) in front and norm2 (tome.py:101) behind,
//                             in ONE pass.  Sources are added to their destination in edge order (= the order torch's CPU
//                             scatter_add applies them), so the fp32 path reproduces the reference's rounding.
// HBM traffic is the algorithmic minimum: match reads the K third of qkv once (2*N*H*64 B per image, bf16) and writes
// 4*(na + r) B of indices; merge reads each input row once and writes each output row once.
#include "tr_common.h"

namespace {

constexpr int TOME_MAX_N = 224;      // tokens incl. CLS (224^2 inputs: 197)
constexpr int MST = 65;              // metric row stride in floats (odd: conflict-free column walks)

template <bool F32>
__device__ __forceinline__ float load_k(const void* qkv, size_t elem) {
  if (F32) return reinterpret_cast<const float*>(qkv)[elem];
  return bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(qkv)[elem]);
}

template <bool F32>
__global__ __launch_bounds__(256) void tome_match_kernel(const void* __restrict__ qkv, int32_t* __restrict__ unm_idx,
                                                         int32_t* __restrict__ src_idx, int32_t* __restrict__ dst_idx, int N, int H,
                                                         int r) {
  __shared__ float s_m[TOME_MAX_N * MST];
  __shared__ float s_max[(TOME_MAX_N + 1) / 2];
  __shared__ int s_arg[(TOME_MAX_N + 1) / 2];
  __shared__ int s_edge[(TOME_MAX_N + 1) / 2];
  __shared__ unsigned char s_unm[(TOME_MAX_N + 1) / 2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int na = (N + 1) >> 1, nb = N >> 1;
  const int ldq = 3 * H * 64;
  // metric = mean over heads of K (post-bias): sequential fp32 sum over h, then / H, like a strided torch mean
  for (int e = tid; e < N * 64; e += 256) {
    const int n = e >> 6, d = e & 63;
    float acc = 0.f;
    for (int h = 0; h < H; ++h) acc += load_k<F32>(qkv, ((size_t)b * N + n) * ldq + H * 64 + h * 64 + d);
    s_m[n * MST + d] = acc / (float)H;
  }
  __syncthreads();
  // metric / metric.norm(dim=-1)
  for (int n = tid; n < N; n += 256) {
    // validation path: accumulate in fp64 and round once -- within 0.5 ulp of the exact value, so the ranking can only differ
    // from the reference's fp32 matmul where its own rounding (a few ulp) decides, i.e. on near-ties below ~4e-7
    float nrm;
    if (F32) {
      double ss = 0.0;
      for (int d = 0; d < 64; ++d) ss += (double)s_m[n * MST + d] * (double)s_m[n * MST + d];
      nrm = (float)sqrt(ss);
    } else {
      float ss = 0.f;
      for (int d = 0; d < 64; ++d) ss = fmaf(s_m[n * MST + d], s_m[n * MST + d], ss);
      nrm = sqrtf(ss);
    }
    for (int d = 0; d < 64; ++d) s_m[n * MST + d] = s_m[n * MST + d] / nrm;
  }
  __syncthreads();
  // row i of scores = a_i . b_j (a = even tokens, b = odd tokens); row 0 is CLS: -inf (never merged)
  for (int i = tid; i < na; i += 256) {
    float best = -INFINITY;
    int arg = 0;
    if (i > 0) {
      const float* ai = s_m + (2 * i) * MST;
      for (int j = 0; j < nb; ++j) {
        const float* bj = s_m + (2 * j + 1) * MST;
        float acc;
        if (F32) {
          double a64 = 0.0;
          for (int d = 0; d < 64; ++d) a64 += (double)ai[d] * (double)bj[d];
          acc = (float)a64;
        } else {
          acc = 0.f;
          for (int d = 0; d < 64; ++d) acc = fmaf(ai[d], bj[d], acc);
        }
        if (acc > best) { best = acc; arg = j; }     // strict >: first index wins ties
      }
    }
    s_max[i] = best;
    s_arg[i] = arg;
  }
  __syncthreads();
  // descending rank of the row maxima (ties: lowest index first) = argsort(descending)
  for (int i = tid; i < na; i += 256) {
    const float vi = s_max[i];
    int rank = 0;
    for (int j = 0; j < na; ++j) {
      const float vj = s_max[j];
      rank += (vj > vi) || (vj == vi && j < i);
    }
    s_edge[rank] = i;
    s_unm[i] = rank >= r;
  }
  __syncthreads();
  for (int e = tid; e < r; e += 256) {
    const int i = s_edge[e];
    src_idx[(size_t)b * r + e] = i;
    dst_idx[(size_t)b * r + e] = s_arg[i];
  }
  // unmerged tokens, ascending (tome.py:275-277: keeps the class token first)
  for (int i = tid; i < na; i += 256) {
    if (!s_unm[i]) continue;
    int pos = 0;
    for (int j = 0; j < i; ++j) pos += s_unm[j];
    unm_idx[(size_t)b * (na - r) + pos] = i;
  }
}

constexpr int LNC = 4;   // float4 chunks per lane -> D <= 1024

template <bool F32>
__device__ __forceinline__ float4 load_d4(const void* base, size_t elem) {
  if (F32) return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem);
  const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem);
  return make_float4(bf16_bits_to_f32((unsigned short)(u.x & 0xffffu)), bf16_bits_to_f32((unsigned short)(u.x >> 16)),
                     bf16_bits_to_f32((unsigned short)(u.y & 0xffffu)), bf16_bits_to_f32((unsigned short)(u.y >> 16)));
}

// one wave per OUTPUT row: [unmerged even tokens | all odd tokens]
template <bool F32>
__global__ __launch_bounds__(256) void tome_merge_layernorm_kernel(const float* __restrict__ x, const void* __restrict__ delta,
                                                                   const float* __restrict__ size_in, const int32_t* __restrict__ unm_idx,
                                                                   const int32_t* __restrict__ src_idx, const int32_t* __restrict__ dst_idx,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ x_out, float* __restrict__ size_out,
                                                                   void* __restrict__ y, int N, int r, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int na = (N + 1) >> 1, nb = N >> 1;
  const int n_unm = na - r, N_out = n_unm + nb;
  const int rblocks = (N_out + 3) >> 2;
  const int b = blockIdx.x / rblocks;
  const int p = (blockIdx.x % rblocks) * 4 + (threadIdx.x >> 6);
  if (p >= N_out) return;
  const int nchunks = D >> 2;
  const float* xb = x + (size_t)b * N * D;
  const size_t dbase = (size_t)b * N * D;
  const float* sb = size_in ? size_in + (size_t)b * N : nullptr;
  float4 v[LNC];
  float sz;
  // (x + pending residual) * size of one input token, accumulated into v
  auto add_token = [&](int t, bool first) __attribute__((always_inline)) {
    const float s = sb ? sb[t] : 1.0f;
#pragma unroll
    for (int c = 0; c < LNC; ++c)
      if (lane + 64 * c < nchunks) {
        float4 a = *reinterpret_cast<const float4*>(xb + (size_t)t * D + 4 * (lane + 64 * c));
        if (delta) {
          const float4 d = load_d4<F32>(delta, dbase + (size_t)t * D + 4 * (lane + 64 * c));
          a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
        }
        if (first) v[c] = make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
        else { v[c].x += a.x * s; v[c].y += a.y * s; v[c].z += a.z * s; v[c].w += a.w * s; }
      }
    if (first) sz = s; else sz += s;
  };
  if (p < n_unm) {
    add_token(2 * unm_idx[(size_t)b * n_unm + p], true);
  } else {
    const int j = p - n_unm;
    add_token(2 * j + 1, true);
    for (int e = 0; e < r; ++e)                       // edge order = the order torch's CPU scatter_add applies the sources
      if (dst_idx[(size_t)b * r + e] == j) add_token(2 * src_idx[(size_t)b * r + e], false);
  }
#pragma unroll
  for (int c = 0; c < LNC; ++c)
    if (lane + 64 * c < nchunks) { v[c].x /= sz; v[c].y /= sz; v[c].z /= sz; v[c].w /= sz; }
  const size_t orow = (size_t)b * N_out + p;
  if (lane == 0) size_out[orow] = sz;
#pragma unroll
  for (int c = 0; c < LNC; ++c)
    if (lane + 64 * c < nchunks) *reinterpret_cast<float4*>(x_out + orow * D + 4 * (lane + 64 * c)) = v[c];
  // norm2 (same arithmetic as tr_norm.hip: two-pass statistics in registers)
  float s1 = 0.f;
#pragma unroll
  for (int c = 0; c < LNC; ++c)
    if (lane + 64 * c < nchunks) s1 += (v[c].x + v[c].y) + (v[c].z + v[c].w);
  const float mean = wave_sum(s1) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < LNC; ++c)
    if (lane + 64 * c < nchunks) {
      const float a = v[c].x - mean, bb = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
      q += (a * a + bb * bb) + (cc * cc + d * d);
    }
  const float var = wave_sum(q) / (float)D + eps;
  const float rstd = F32 ? 1.0f / sqrtf(var) : rsqrtf(var);
#pragma unroll
  for (int c = 0; c < LNC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * ch);
      const float4 be = *reinterpret_cast<const float4*>(beta + 4 * ch);
      const float o0 = (v[c].x - mean) * rstd * g.x + be.x, o1 = (v[c].y - mean) * rstd * g.y + be.y;
      const float o2 = (v[c].z - mean) * rstd * g.z + be.z, o3 = (v[c].w - mean) * rstd * g.w + be.w;
      if (F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + orow * D + 4 * ch) = make_float4(o0, o1, o2, o3);
      } else {
        uint2 pk;
        pk.x = pack_bf16x2(o0, o1);
        pk.y = pack_bf16x2(o2, o3);
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(y) + orow * D + 4 * ch) = pk;
      }
    }
  }
}

}  // namespace

extern "C" int tr_tome_match(const void* qkv, int qkv_is_f32, int32_t* unm_idx, int32_t* src_idx, int32_t* dst_idx, int B, int N,
                             int H, int r, tr_stream_t s) {
  TR_REQUIRE(qkv && unm_idx && src_idx && dst_idx, TR_ERR_NULL, "tr_tome_match: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 3 && N <= TOME_MAX_N, TR_ERR_SHAPE, "tr_tome_match: need 3 <= N <= %d (N=%d)", TOME_MAX_N, N);
  TR_REQUIRE(r >= 1 && r <= (N - 1) / 2, TR_ERR_SHAPE, "tr_tome_match: r=%d must be in [1, (N-1)/2] for N=%d (tome.py:253)", r, N);
  hipStream_t st = static_cast<hipStream_t>(s);
  if (qkv_is_f32) hipLaunchKernelGGL(tome_match_kernel<true>, dim3(B), dim3(256), 0, st, qkv, unm_idx, src_idx, dst_idx, N, H, r);
  else hipLaunchKernelGGL(tome_match_kernel<false>, dim3(B), dim3(256), 0, st, qkv, unm_idx, src_idx, dst_idx, N, H, r);
  TR_CHECK_LAUNCH("tr_tome_match");
  return TR_OK;
}

extern "C" int tr_tome_merge_layernorm(const float* x, const void* delta, int f32_path, const float* size_in, const int32_t* unm_idx,
                                       const int32_t* src_idx, const int32_t* dst_idx, const float* gamma, const float* beta,
                                       float* x_out, float* size_out, void* y, int B, int N, int r, int D, float eps, tr_stream_t s) {
  TR_REQUIRE(x && unm_idx && src_idx && dst_idx && gamma && beta && x_out && size_out && y, TR_ERR_NULL, "tr_tome_merge_layernorm: null pointer");
  TR_REQUIRE(B > 0 && N >= 3 && D > 0 && D % 4 == 0 && D <= 256 * LNC, TR_ERR_SHAPE, "tr_tome_merge_layernorm: bad shape B=%d N=%d D=%d", B, N, D);
  TR_REQUIRE(r >= 1 && r <= (N - 1) / 2, TR_ERR_SHAPE, "tr_tome_merge_layernorm: r=%d out of range for N=%d", r, N);
  TR_REQUIRE(x_out != x, TR_ERR_SHAPE, "tr_tome_merge_layernorm: needs a distinct x_out");
  TR_REQUIRE(tr_aligned16(x) && tr_aligned16(delta) && tr_aligned16(x_out) && tr_aligned16(y) && tr_aligned16(gamma) && tr_aligned16(beta),
             TR_ERR_ALIGN, "tr_tome_merge_layernorm: pointers must be 16-byte aligned");
  const int N_out = N - r;
  const int rblocks = (N_out + 3) / 4;
  hipStream_t st = static_cast<hipStream_t>(s);
  if (f32_path)
    hipLaunchKernelGGL(tome_merge_layernorm_kernel<true>, dim3(B * rblocks), dim3(256), 0, st, x, delta, size_in, unm_idx, src_idx,
                       dst_idx, gamma, beta, x_out, size_out, y, N, r, D, eps);
  else
    hipLaunchKernelGGL(tome_merge_layernorm_kernel<false>, dim3(B * rblocks), dim3(256), 0, st, x, delta, size_in, unm_idx, src_idx,
                       dst_idx, gamma, beta, x_out, size_out, y, N, r, D, eps);
  TR_CHECK_LAUNCH("tr_tome_merge_layernorm");
  return TR_OK;
}
