// Reducers that run BEFORE a block: DyViT's score predictor tail (models/dyvit.py) and SiT's token slimming (models/sit.py).
// The GEMMs of both modules go through tr_gemm_*; this file holds the three non-GEMM pieces, all HBM-bound:
//
//   tr_pool_broadcast    PredictorLG.forward dyvit.py:115-118 with policy == 1 (eval): the second half of the channels is
//                        replaced, in place, by its mean over the image's patch tokens (+ eps outside the fraction).
//   tr_dyvit_score       out_conv.4 + LogSoftmax dyvit.py:108-109 and score = pred_score[:,:,0] dyvit.py:231: two dot products
//                        per token, log-softmax over the pair, first component -> scores [B,N] fp32 (column 0 = CLS, unused).
//   tr_sit_merge         TokenSlimmingModule.forward sit.py:37-39: softmax(logits * scale) over the TOKEN axis, out = W^T x,
//                        CLS row copied through (sit.py:117-119); optional soft-assignment output [B,K,P] (viz, sit.py:124).
#include "tr_common.h"

namespace {

template <bool F32>
__device__ __forceinline__ float load1(const void* p, size_t e) {
  if (F32) return reinterpret_cast<const float*>(p)[e];
  return bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(p)[e]);
}
template <bool F32>
__device__ __forceinline__ void store1(void* p, size_t e, float v) {
  if (F32) reinterpret_cast<float*>(p)[e] = v;
  else reinterpret_cast<uint16_t*>(p)[e] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
}

// grid (ceil(C/2 / 64), B); lanes = adjacent channels (coalesced rows), 4 waves split the tokens
template <bool F32>
__global__ __launch_bounds__(256) void pool_broadcast_kernel(void* __restrict__ h, int N, int C, float eps) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int c = C / 2 + blockIdx.x * 64 + lane;
  const bool ok = c < C;
  const size_t base = (size_t)b * N * C;
  float acc = 0.f;
  if (ok)
    for (int n = 1 + wave; n < N; n += 4) acc += load1<F32>(h, base + (size_t)n * C + c);    // patch tokens only (x[:, 1:])
  part[wave][lane] = acc;
  __syncthreads();
  const float g = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / (float)(N - 1) + eps;
  if (ok)
    for (int n = wave; n < N; n += 4) store1<F32>(h, base + (size_t)n * C + c, g);
}

// 16 lanes per token row, 4 rows per wave
template <bool F32>
__global__ __launch_bounds__(256) void dyvit_score_kernel(const void* __restrict__ h, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ scores, int M,
                                                          int C) {
  const int sub = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  float a0 = 0.f, a1 = 0.f;
  if (row < M)
    for (int c = sub; c < C; c += 16) {
      const float v = load1<F32>(h, (size_t)row * C + c);
      a0 = fmaf(v, w[c], a0);
      a1 = fmaf(v, w[C + c], a1);
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    a0 += __shfl_xor(a0, o, 64);
    a1 += __shfl_xor(a1, o, 64);
  }
  if (row < M && sub == 0) {
    const float l0 = a0 + bias[0], l1 = a1 + bias[1];
    const float m = fmaxf(l0, l1);
    // torch log_softmax: x - max - log(sum(exp(x - max)))
    scores[row] = (l0 - m) - logf(expf(l0 - m) + expf(l1 - m));
  }
}

constexpr int SKC = 32;    // clusters per workgroup
constexpr int SJ_MAX = 4;  // D columns per thread (D <= 1024)

// grid (ceil(K/32), B), 256 threads, SJ = ceil(D/256) columns per thread.  Dynamic LDS: w[P][SKC] fp32.
// SOFTMAX (SiT): w = softmax over tokens of logits*scale.  !SOFTMAX (Sinkhorn): `logits` already holds the weights [B,N,ldl].
// src = the rows that are summed (x for SiT, the unit-norm tokens for Sinkhorn); the CLS row always comes from x.
template <int SJ, bool SOFTMAX>
__global__ __launch_bounds__(256) void sit_merge_kernel(const float* __restrict__ logits, int ldl, float scale,
                                                        const float* __restrict__ x, const float* __restrict__ src,
                                                        float* __restrict__ x_out, float* __restrict__ soft, int N, int K, int D) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];     // [P][SKC]
  __shared__ float s_red[8][SKC];
  __shared__ float s_max[SKC], s_inv[SKC];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, k0 = blockIdx.x * SKC;
  const int P = N - 1;
  const int kk = tid & 31, pg = tid >> 5;                         // column kk, token group pg (8 groups)
  const bool kval = k0 + kk < K;
  const float* lg = logits + ((size_t)b * N + 1) * ldl + k0;      // patch rows only
  // softmax over tokens of (logit * scale), column by column (F.softmax(weight * self.scale, dim=1), sit.py:38)
  float mx = -INFINITY;
  for (int p = pg; p < P; p += 8) {
    const float v = kval ? lg[(size_t)p * ldl + kk] * scale : 0.f;
    s_w[p * SKC + kk] = v;
    mx = fmaxf(mx, v);
  }
  if (SOFTMAX) {
  s_red[pg][kk] = mx;
  __syncthreads();
  if (tid < SKC) {
    float m = s_red[0][tid];
#pragma unroll
    for (int g = 1; g < 8; ++g) m = fmaxf(m, s_red[g][tid]);
    s_max[tid] = m;
  }
  __syncthreads();
  float sum = 0.f;
  const float m = s_max[kk];
  for (int p = pg; p < P; p += 8) {
    const float e = expf(s_w[p * SKC + kk] - m);
    s_w[p * SKC + kk] = e;
    sum += e;
  }
  s_red[pg][kk] = sum;
  __syncthreads();
  if (tid < SKC) {
    float t = s_red[0][tid];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += s_red[g][tid];
    s_inv[tid] = 1.0f / t;
  }
  __syncthreads();
  const float inv = s_inv[kk];
  for (int p = pg; p < P; p += 8) s_w[p * SKC + kk] *= inv;
  }
  __syncthreads();
  if (soft != nullptr)                                            // [B,K,P], lanes along p
    for (int e = tid; e < SKC * P; e += 256) {
      const int k = e / P, p = e - k * P;
      if (k0 + k < K) soft[((size_t)b * K + k0 + k) * P + p] = s_w[p * SKC + k];
    }
  // out[k][d] = sum_p w[p][k] * x[1+p][d]   (torch.bmm(weight, x), sit.py:39), p ascending in fp32
  float acc[SJ][SKC];
#pragma unroll
  for (int j = 0; j < SJ; ++j)
#pragma unroll
    for (int k = 0; k < SKC; ++k) acc[j][k] = 0.f;
  const float* xb = src + ((size_t)b * N + 1) * D;
  for (int p = 0; p < P; ++p) {
    float xv[SJ];
#pragma unroll
    for (int j = 0; j < SJ; ++j) xv[j] = (tid + 256 * j < D) ? xb[(size_t)p * D + tid + 256 * j] : 0.f;
#pragma unroll
    for (int k4 = 0; k4 < SKC / 4; ++k4) {
      const float4 wv = *reinterpret_cast<const float4*>(&s_w[p * SKC + 4 * k4]);
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        acc[j][4 * k4 + 0] = fmaf(wv.x, xv[j], acc[j][4 * k4 + 0]);
        acc[j][4 * k4 + 1] = fmaf(wv.y, xv[j], acc[j][4 * k4 + 1]);
        acc[j][4 * k4 + 2] = fmaf(wv.z, xv[j], acc[j][4 * k4 + 2]);
        acc[j][4 * k4 + 3] = fmaf(wv.w, xv[j], acc[j][4 * k4 + 3]);
      }
    }
  }
  float* ob = x_out + ((size_t)b * (K + 1) + 1 + k0) * D;
#pragma unroll
  for (int j = 0; j < SJ; ++j) {
    const int d = tid + 256 * j;
    if (d < D) {
#pragma unroll
      for (int k = 0; k < SKC; ++k)
        if (k0 + k < K) ob[(size_t)k * D + d] = acc[j][k];
      if (blockIdx.x == 0) x_out[(size_t)b * (K + 1) * D + d] = x[(size_t)b * N * D + d];     // global (CLS) token
    }
  }
}

// ---- fast path of the soft merges (SiT, PatchMerger, Sinkhorn) for the bf16 executor --------------------------------------
// softmax over the token axis, in place on the token-major logits (F.softmax(weight * scale, dim=1), sit.py:38).
// grid (ceil(K/32), B): 32 output columns per workgroup (lanes along k: coalesced), 8 token groups reduced through LDS
// PER > 0: a thread's (at most PER) logits stay in registers -- one read pass with every load in flight, one write pass (round 4: the three
// rolled passes re-read the column with a dependent load per element: 20 us per launch, 2.1 x the algorithmic bytes).  PER = 0: any length.
// The max and the sum are taken over the same elements in the same order in both forms (thread's elements ascending, then the 8 groups).
template <int PER>
__global__ __launch_bounds__(256) void token_softmax_kernel(float* __restrict__ logits, int ldl, float scale, float* __restrict__ soft,
                                                            int N, int K) {
  __shared__ float s_red[8][32];
  __shared__ float s_max[32], s_inv[32];
  const int tid = threadIdx.x, kk = tid & 31, pg = tid >> 5;
  const int b = blockIdx.y, k = blockIdx.x * 32 + kk;
  const bool kval = k < K;
  const int P = N - 1;
  float* lg = logits + ((size_t)b * N + 1) * ldl + blockIdx.x * 32 + min(kk, K - 1 - (int)blockIdx.x * 32);      // clamped: loads stay branch-free
  float v[PER > 0 ? PER : 1];
  float mx = -INFINITY;
  if (PER > 0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = lg[(size_t)min(pg + 8 * i, P - 1) * ldl] * scale;
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if (pg + 8 * i < P) mx = fmaxf(mx, v[i]);
  } else if (kval) {
    for (int p = pg; p < P; p += 8) mx = fmaxf(mx, lg[(size_t)p * ldl] * scale);
  }
  s_red[pg][kk] = mx;
  __syncthreads();
  if (tid < 32) {
    float m = s_red[0][tid];
#pragma unroll
    for (int g = 1; g < 8; ++g) m = fmaxf(m, s_red[g][tid]);
    s_max[tid] = m;
  }
  __syncthreads();
  const float m = s_max[kk];
  float sum = 0.f;
  if (PER > 0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      v[i] = expf(v[i] - m);
      if (pg + 8 * i < P) sum += v[i];
    }
  } else if (kval) {
    for (int p = pg; p < P; p += 8) sum += expf(lg[(size_t)p * ldl] * scale - m);
  }
  __syncthreads();
  s_red[pg][kk] = sum;
  __syncthreads();
  if (tid < 32) {
    float t = s_red[0][tid];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += s_red[g][tid];
    s_inv[tid] = 1.0f / t;
  }
  __syncthreads();
  const float inv = s_inv[kk];
  if (!kval) return;
  if (PER > 0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = pg + 8 * i;
      if (p < P) {
        const float w = v[i] * inv;
        lg[(size_t)p * ldl] = w;
        if (soft != nullptr) soft[((size_t)b * K + k) * P + p] = w;
      }
    }
  } else {
    for (int p = pg; p < P; p += 8) {
      const float w = expf(lg[(size_t)p * ldl] * scale - m) * inv;
      lg[(size_t)p * ldl] = w;
      if (soft != nullptr) soft[((size_t)b * K + k) * P + p] = w;
    }
  }
}

// out[b][1+k][:] = sum_p w[b][1+p][k] * src[b][1+p][:] on MFMA: both operands are split into bf16 hi + lo (3 products,
// relative error ~2^-16).  The token axis is the contraction and the SLOW axis of both operands in memory, so both are staged
// ROW-major (token rows, 16- and 8-byte LDS writes) and read TRANSPOSED with ds_read_b64_tr_b16 -- the layout and the read pattern of V
// in attention16_kernel.  (Until round 4 the staging transposed by hand: one 2-byte LDS write per element, 64 per thread and 32-token
// slab, for 27 MFMAs per wave: 75 us per launch, 7 % of a SiT forward.)  Workgroup = one image x 64 feature columns, wave = 16 of
// them (the permuted 16 of the transposed read: lane group g, register r <-> column 32 (w >> 1) + 8 g + 4 (w & 1) + r) x all K (<= 192)
// centres; tokens walked in slabs of 32: k-slot 8 g + j of a fragment <-> token 16 (j >> 2) + 4 g + (j & 3) for BOTH operands.
constexpr int MK_MAX = 192;
constexpr int MWS = 544;                 // weight image row pitch in bytes: 272 centres, and 544 = 32 (mod 256): the 16 rows of a transposed
                                         // read fall into 8 distinct 32-byte windows, two rows each (tools/lds_sim.py's criterion: 2 passes)
typedef __attribute__((ext_vector_type(4))) short m_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short m_s16x8_t;
__device__ __forceinline__ bf16x8 m_lds_tr_pair16(const unsigned char* p0, const unsigned char* p1) {
  const m_s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) m_s16x4_t*)(p0));
  const m_s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) m_s16x4_t*)(p1));
  const m_s16x8_t c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}
// four fp32 -> four bf16 hi (8 bytes) + four bf16 lo (8 bytes)
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  const unsigned h0 = pack_bf16x2(v.x, v.y), h1 = pack_bf16x2(v.z, v.w);
  lo.x = pack_bf16x2(v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u));
  lo.y = pack_bf16x2(v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u));
  hi.x = h0;
  hi.y = h1;
}

__global__ __launch_bounds__(256) void softmerge_mfma_kernel(const float* __restrict__ wt, int ldl, const float* __restrict__ x,
                                                             const float* __restrict__ src, float* __restrict__ x_out, int N, int K,
                                                             int D) {
  __shared__ __attribute__((aligned(16))) unsigned char sSh[32 * 128], sSl[32 * 128], sWh[32 * MWS], sWl[32 * MWS];
  const int P = N - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4, q4 = li >> 2, p4 = li & 3;
  // The feature slices of an image (6 at D = 384, 12 at D = 768) all read the image's whole weight matrix: give them consecutive ids of the
  // XCD-contiguous order, so that ONE L2 fetches it (r04i: 155.7 MB fetched per launch at SiT-S against 71.7 MB of operands -- 1.75 x the
  // algorithmic bytes -- with the slices dealt round-robin over the eight L2s)
  const int nsl = (D + 63) >> 6;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = lid / nsl, d0 = (lid - b * nsl) * 64;
  const int nkf = (K + 15) >> 4, Kp = nkf * 16;
  const float* sb = src + ((size_t)b * N + 1) * D;
  const float* wb = wt + ((size_t)b * N + 1) * ldl;
  f32x4 acc[MK_MAX / 16];
#pragma unroll
  for (int i = 0; i < MK_MAX / 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int sp = tid >> 3, sc = tid & 7;                           // src staging: token sp of the slab, 16-byte chunk sc (8 feature columns)
  // Register-staged prefetch: the loads of slab p0+32 (2 float4 of src, up to 6 float4 of weights per thread) are issued before
  // the MFMAs of slab p0.
  constexpr int WIT = 32 * (MK_MAX / 4) / 256;                     // 6: float4 weight loads per thread and slab, at most
  const int kq4 = Kp >> 2;                                         // float4 groups per token
  float4 s0, s1, w4[WIT];
  auto load_slab = [&](int p0) __attribute__((always_inline)) {
    s0 = make_float4(0.f, 0.f, 0.f, 0.f);
    s1 = s0;
    if (p0 + sp < P && d0 + 8 * sc < D) {
      s0 = *reinterpret_cast<const float4*>(sb + (size_t)(p0 + sp) * D + d0 + 8 * sc);
      s1 = *reinterpret_cast<const float4*>(sb + (size_t)(p0 + sp) * D + d0 + 8 * sc + 4);
    }
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int idx = tid + 256 * it;
      const int p = idx / kq4, kq = (idx - p * kq4) * 4;
      w4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < 32 * kq4 && p0 + p < P && kq < ldl) w4[it] = *reinterpret_cast<const float4*>(wb + (size_t)(p0 + p) * ldl + kq);
    }
  };
  // transposed-read addresses (attention16_kernel's V image): src block `wave`; lane 4 q4 + p4 addresses token row 4 g + q4 (+ 16), piece p4
  const int dbw = wave;
  const int cS = 4 * (dbw >> 1) + p4, hbS = dbw & 1;
  const unsigned aoff = (unsigned)((4 * g + q4) * 128 + ((cS ^ (((q4 >> 1) & 1) << 2)) << 4) + ((hbS ^ (g & 1)) << 3));
  const unsigned woff = (unsigned)((4 * g + q4) * MWS + 8 * p4);    // + 32 kf for the centre block
  load_slab(0);
  for (int p0 = 0; p0 < P; p0 += 32) {
    lds_barrier();                                                 // previous slab's fragment reads are done (LDS only)
    {
      uint2 h0, l0, h1, l1;
      split4(s0, h0, l0);
      split4(s1, h1, l1);
      uint4 hv = make_uint4(h0.x, h0.y, h1.x, h1.y), lv = make_uint4(l0.x, l0.y, l1.x, l1.y);
      if ((sp >> 2) & 1) { hv = make_uint4(hv.z, hv.w, hv.x, hv.y); lv = make_uint4(lv.z, lv.w, lv.x, lv.y); }     // halves swapped on rows 4..7 (mod 8)
      const unsigned o = (unsigned)(sp * 128 + ((sc ^ (((sp >> 1) & 1) << 2)) << 4));
      *reinterpret_cast<uint4*>(sSh + o) = hv;
      *reinterpret_cast<uint4*>(sSl + o) = lv;
    }
#pragma unroll
    for (int it = 0; it < WIT; ++it) {                             // weights: 4 consecutive centres of one token per thread
      const int idx = tid + 256 * it;
      if (idx < 32 * kq4) {
        const int p = idx / kq4, kq = (idx - p * kq4) * 4;
        float4 v = w4[it];
        if (kq + 0 >= K) v.x = 0.f;
        if (kq + 1 >= K) v.y = 0.f;
        if (kq + 2 >= K) v.z = 0.f;
        if (kq + 3 >= K) v.w = 0.f;
        uint2 h, l;
        split4(v, h, l);
        *reinterpret_cast<uint2*>(sWh + p * MWS + kq * 2) = h;
        *reinterpret_cast<uint2*>(sWl + p * MWS + kq * 2) = l;
      }
    }
    if (p0 + 32 < P) load_slab(p0 + 32);
    lds_barrier();                                                 // the prefetch above stays in flight under the MFMAs
    const bf16x8 sh = m_lds_tr_pair16(sSh + aoff, sSh + aoff + 2048);
    const bf16x8 sl = m_lds_tr_pair16(sSl + aoff, sSl + aoff + 2048);
#pragma unroll
    for (int kf = 0; kf < MK_MAX / 16; ++kf)
      if (kf < nkf) {
        const bf16x8 wh = m_lds_tr_pair16(sWh + woff + 32 * kf, sWh + woff + 32 * kf + 16 * MWS);
        const bf16x8 wl = m_lds_tr_pair16(sWl + woff + 32 * kf, sWl + woff + 32 * kf + 16 * MWS);
        acc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sl, wh, acc[kf], 0, 0, 0);
        acc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sh, wl, acc[kf], 0, 0, 0);
        acc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sh, wh, acc[kf], 0, 0, 0);
      }
  }
  // acc[kf][e]: feature column d0 + 32 (wave >> 1) + 8 g + 4 (wave & 1) + e, centre kf * 16 + li
  const int d = d0 + 32 * (dbw >> 1) + 8 * g + 4 * (dbw & 1);
  if (d < D) {
#pragma unroll
    for (int kf = 0; kf < MK_MAX / 16; ++kf) {
      const int k = kf * 16 + li;
      if (kf < nkf && k < K)
        *reinterpret_cast<float4*>(x_out + ((size_t)b * (K + 1) + 1 + k) * D + d) =
            make_float4(acc[kf][0], acc[kf][1], acc[kf][2], acc[kf][3]);
    }
  }
  if (tid < 16 && d0 + 4 * tid < D)                                 // global (CLS) token
    *reinterpret_cast<float4*>(x_out + (size_t)b * (K + 1) * D + d0 + 4 * tid) =
        *reinterpret_cast<const float4*>(x + (size_t)b * N * D + d0 + 4 * tid);
}

// ---- Sinkhorn (models/sinkhorn.py) ------------------------------------------------------------------------------------
// one wave per token row: xh = x / max(|x|_2, 1e-12)   (F.normalize, sinkhorn.py:70), fp32 copy + GEMM-operand copy
template <bool F32>
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, float* __restrict__ xh, void* __restrict__ xh_lp,
                                                      int M, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * D;
  if ((D & 3) == 0 && D <= 1024) {
    // the row stays in registers: one 16-byte load per lane and chunk, one pass (the scalar version below read it twice, 4 B per lane)
    const int nch = D >> 2;
    float4 v[4];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const float4*>(xr + 4 * min(lane + 64 * c, nch - 1));   // branch-free: one batch
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (lane + 64 * c < nch) {
        ss = fmaf(v[c].x, v[c].x, ss); ss = fmaf(v[c].y, v[c].y, ss); ss = fmaf(v[c].z, v[c].z, ss); ss = fmaf(v[c].w, v[c].w, ss);
      }
    ss = wave_sum(ss);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (lane + 64 * c < nch) {
        const size_t e = (size_t)row * D + 4 * (lane + 64 * c);
        const float4 o = make_float4(v[c].x / nrm, v[c].y / nrm, v[c].z / nrm, v[c].w / nrm);
        *reinterpret_cast<float4*>(xh + e) = o;
        if (F32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(xh_lp) + e) = o;
        } else {
          uint2 pk;
          pk.x = pack_bf16x2(o.x, o.y);
          pk.y = pack_bf16x2(o.z, o.w);
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(xh_lp) + e) = pk;
        }
      }
    return;
  }
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) ss = fmaf(xr[d], xr[d], ss);
  ss = wave_sum(ss);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  for (int d = lane; d < D; d += 64) {
    const float v = xr[d] / nrm;
    xh[(size_t)row * D + d] = v;
    store1<F32>(xh_lp, (size_t)row * D + d, v);
  }
}

// one workgroup (16 waves) per image; Z[k][p] = scores[p][k] / eps lives in LDS (K*P floats) through the iterations
// (log_optimal_transport sinkhorn.py:41-56 over log_sinkhorn_iterations :25-38).  u-step: one wave per centre row;
// v-step: four lanes per token column, combined with two shuffles.
constexpr int SKT = 1024;
__global__ __launch_bounds__(SKT) void sinkhorn_kernel(const float* __restrict__ scores, int ldl, float eps, int iters,
                                                       float* __restrict__ wt, float* __restrict__ soft, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float s_z[];      // [K][P], then u[K], v[P]
  const int P = N - 1;
  float* s_u = s_z + (size_t)K * P;
  float* s_v = s_u + K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const float* sc = scores + ((size_t)b * N + 1) * ldl;
  for (int e = tid; e < K * P; e += SKT) {
    const int p = e / K, k = e - p * K;                            // lanes along k: coalesced global reads
    s_z[k * P + p] = sc[(size_t)p * ldl + k] / eps;
  }
  for (int p = tid; p < P; p += SKT) s_v[p] = 0.f;
  for (int k = tid; k < K; k += SKT) s_u[k] = 0.f;
  const float norm = -logf((float)K + (float)P);                   // log_mu = log_nu = -log(m + n)
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    for (int k = wave; k < K; k += SKT / 64) {                     // u = log_mu - logsumexp_p(Z + v)
      const float* zr = s_z + (size_t)k * P;
      float m = -INFINITY;
      for (int p = lane; p < P; p += 64) m = fmaxf(m, zr[p] + s_v[p]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      float t = 0.f;
      for (int p = lane; p < P; p += 64) t += expf(zr[p] + s_v[p] - m);
      t = wave_sum(t);
      if (lane == 0) s_u[k] = norm - (m + logf(t));
    }
    __syncthreads();
    const int sub = tid & 3;
    for (int p0 = 0; p0 < P; p0 += SKT / 4) {                      // v = log_nu - logsumexp_k(Z + u)
      const int p = p0 + (tid >> 2);
      const bool ok = p < P;
      float m = -INFINITY;
      if (ok)
        for (int k = sub; k < K; k += 4) m = fmaxf(m, s_z[k * P + p] + s_u[k]);
      m = fmaxf(m, __shfl_xor(m, 1, 64));
      m = fmaxf(m, __shfl_xor(m, 2, 64));
      float t = 0.f;
      if (ok)
        for (int k = sub; k < K; k += 4) t += expf(s_z[k * P + p] + s_u[k] - m);
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      if (ok && sub == 0) s_v[p] = norm - (m + logf(t));
    }
    __syncthreads();
  }
  // W = exp(Z + u + v - norm): token-major copy for the merge, cluster-major copy for Soft_Assignment_Maps
  for (int e = tid; e < K * P; e += SKT) {
    const int k = e / P, p = e - k * P;
    const float w = expf(s_z[e] + s_u[k] + s_v[p] - norm);
    s_z[e] = w;
    if (soft != nullptr) soft[((size_t)b * K + k) * P + p] = w;
  }
  __syncthreads();
  float* wb = wt + ((size_t)b * N + 1) * ldl;
  for (int e = tid; e < K * P; e += SKT) {
    const int p = e / K, k = e - p * K;
    wb[(size_t)p * ldl + k] = s_z[k * P + p];
  }
}

// Same iterations when K*P floats exceed the LDS (384^2 inputs: 144 x 576 = 332 KB): Z stays in the token-major scores buffer
// (L2 / Infinity-Cache resident) and is re-read by every half-iteration, by all 16 waves of one workgroup per image:
//   u-step  thread = (centre k, token segment): lanes along k (coalesced 256-byte rows), the 1024 threads split the tokens into
//           1024 / ceil64(K) segments; two passes (max, then sum of exponentials) with the partials combined through LDS in
//           segment order -- the same max-then-sum formulation as sinkhorn_kernel;
//   v-step  one wave per token row, four rows in flight per wave, lanes along k.
// All loads of a pass are issued in batches of eight (independent addresses).  The first version of this kernel ran 256 threads
// per image with one thread walking all tokens of a centre: 1384 us at B = 64, K = 144, P = 576 -- a quarter of the 384^2 forward.
constexpr int SGT = 1024;
constexpr int SG_KMAX = 640;                 // centres the wide instantiation holds per token row (K = 518 at 384 x 384, keep_rate 0.9)
// KC: 64-centre groups a lane holds of one token row in the v-step (4: K <= 256, every registered model; 10: K <= 640)
template <int KC>
__global__ __launch_bounds__(SGT) void sinkhorn_global_kernel(const float* __restrict__ scores, int ldl, float eps, int iters,
                                                              float* __restrict__ wt, float* __restrict__ soft, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float s_uv[];     // u[K], v[P], partials[nseg][KW]
  const int P = N - 1;
  const int KW = (K + 63) & ~63;             // centre columns rounded up to whole waves
  const int nseg = SGT / KW;                 // token segments of the u-step
  float* s_u = s_uv;
  float* s_v = s_uv + K;
  float* s_part = s_v + P;                   // [nseg][KW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const float* sc = scores + ((size_t)b * N + 1) * ldl;
  const float inv_eps = 1.0f / eps;
  for (int p = tid; p < P; p += SGT) s_v[p] = 0.f;
  for (int k = tid; k < K; k += SGT) s_u[k] = 0.f;
  const float norm = -logf((float)K + (float)P);
  const int kc = tid % KW, seg = tid / KW;   // u-step role (threads with seg >= nseg or kc >= K only help at the barriers)
  const bool u_active = seg < nseg && kc < K;
  const int seg_len = (P + nseg - 1) / nseg;
  const int p_lo = min(seg * seg_len, P), p_hi = min(p_lo + seg_len, P);
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    // ---- u = log_mu - logsumexp_p(Z + v)
    float m = -INFINITY;
    if (u_active) {
      for (int p0 = p_lo; p0 < p_hi; p0 += 8) {
        float z[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) z[q] = sc[(size_t)min(p0 + q, p_hi - 1) * ldl + kc];
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (p0 + q < p_hi) m = fmaxf(m, z[q] * inv_eps + s_v[p0 + q]);
      }
      s_part[seg * KW + kc] = m;
    }
    __syncthreads();
    float t = 0.f;
    if (u_active) {
      for (int sg = 0; sg < nseg; ++sg) m = fmaxf(m, s_part[sg * KW + kc]);
      for (int p0 = p_lo; p0 < p_hi; p0 += 8) {
        float z[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) z[q] = sc[(size_t)min(p0 + q, p_hi - 1) * ldl + kc];
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (p0 + q < p_hi) t += expf(z[q] * inv_eps + s_v[p0 + q] - m);
      }
    }
    __syncthreads();                          // every thread has read the maxima: the partial buffer can take the sums
    if (u_active) s_part[seg * KW + kc] = t;
    __syncthreads();
    if (u_active && seg == 0) {
      float tt = 0.f;
      for (int sg = 0; sg < nseg; ++sg) tt += s_part[sg * KW + kc];      // fixed order
      s_u[kc] = norm - (m + logf(tt));
    }
    __syncthreads();
    // ---- v = log_nu - logsumexp_k(Z + u): wave per token row, four rows per step
    for (int p0 = wave * 4; p0 < P; p0 += (SGT / 64) * 4) {
      float z[4][KC];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const int k = lane + 64 * c;
          z[r][c] = sc[(size_t)min(p0 + r, P - 1) * ldl + min(k, K - 1)];
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float mm = -INFINITY;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const int k = lane + 64 * c;
          z[r][c] = (k < K) ? z[r][c] * inv_eps + s_u[min(k, K - 1)] : -INFINITY;
          mm = fmaxf(mm, z[r][c]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o, 64));
        float ts = 0.f;
#pragma unroll
        for (int c = 0; c < KC; ++c) ts += expf(z[r][c] - mm);     // exp(-inf) = 0 for the padding lanes
        ts = wave_sum(ts);
        if (lane == 0 && p0 + r < P) s_v[p0 + r] = norm - (mm + logf(ts));
      }
    }
    __syncthreads();
  }
  float* wb = wt + ((size_t)b * N + 1) * ldl;
  for (int e = tid; e < K * P; e += SGT) {
    const int p = e / K, k = e - p * K;
    const float w = expf(sc[(size_t)p * ldl + k] * inv_eps + s_u[k] + s_v[p] - norm);   // read-then-write by the same thread: wt may alias scores
    wb[(size_t)p * ldl + k] = w;
    if (soft != nullptr) soft[((size_t)b * K + k) * P + p] = w;
  }
}

// The same iterations with the transport plan in REGISTERS (round 4): 384 x 384 inputs at keep_rate 0.25 have K = 144 centres x P = 576
// tokens = 332 KB per image -- twice the LDS, but 8 waves x 64 lanes x 216 registers hold it.  sinkhorn_global_kernel above re-reads
// the plan from L2 in every half-iteration (4 passes per iteration, 248 us at B = 64: 5.6 % of the DeiT-B 384^2 forward for one launch).
// Wave w owns the RPW token rows p = w * RPW + r; a lane holds centre columns lane + 64 c (c < KC) of each: Z is read ONCE, pre-scaled
// by 1 / eps.  u-step: each lane reduces its columns over the wave's rows, the 8 wave partials meet in LDS (fixed order);
// v-step: a token row is spread over one wave -- row-wide max and sum by DPP + v_permlane swaps (full-rate VALU; 2 x RPW ds_bpermute
// reductions per half-iteration would queue on the LDS pipe).  Same max-then-sum formulation as the other two kernels; the v-step's
// arithmetic is theirs, the u-step's partial order differs (8 row ranges instead of 1024 / ceil64(K) segments): last-bit differences.
__device__ __forceinline__ float wave_max_valu(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  const auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned a0 = a[0], a1 = a[1];
  v = fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1));
  const auto c = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned c0 = c[0], c1 = c[1];
  return fmaxf(__builtin_bit_cast(float, c0), __builtin_bit_cast(float, c1));
}
__device__ __forceinline__ float wave_sum_valu(float v) {
  v = row16_sum(v);
  const auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned a0 = a[0], a1 = a[1];
  v = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
  const auto c = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned c0 = c[0], c1 = c[1];
  return __builtin_bit_cast(float, c0) + __builtin_bit_cast(float, c1);
}
#ifndef TR_SR_RPW
#define TR_SR_RPW 72
#endif
#ifndef TR_SR_NW
#define TR_SR_NW 8
#endif
constexpr int SR_NW = TR_SR_NW, SR_RPW = TR_SR_RPW, SR_KC = 3;           // 8 waves x 72 rows = 576 tokens, 3 x 64 = 192 centres: 216 plan registers per lane
                                                            // (two waves per SIMD: 256 registers each; 12 x 48 spilled 57 of its 168)
__global__ __launch_bounds__(64 * SR_NW) void sinkhorn_regs_kernel(const float* __restrict__ scores, int ldl, float eps, int iters,
                                                                   float* __restrict__ wt, float* __restrict__ soft, int N, int K) {
  constexpr int KW = 64 * SR_KC;
  __shared__ float s_u[KW], s_v[SR_NW * SR_RPW], s_part[SR_NW][KW];
  const int P = N - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const float* sc = scores + ((size_t)b * N + 1) * ldl;
  // everything in the log2 domain (Z / eps, u, v and the normaliser times log2(e)): exp and log are then ONE transcendental instruction
  // each (v_exp_f32 / v_log_f32) instead of libm's range-checked sequences -- the kernel is bound by VALU issue (1512 exponentials per lane)
  constexpr float L2E = 1.44269504088896340736f;
  const float inv_eps = L2E / eps;
  const float norm = -__builtin_amdgcn_logf((float)K + (float)P);
  const int p0 = __builtin_amdgcn_readfirstlane(wave * SR_RPW);
  // Buffer addressing: the row offset is a SCALAR (soffset), the column a per-lane byte offset -- with flat addresses the compiler kept a
  // 64-bit address pair per row and matrix alive (232 registers for 120 of plan).  The launcher sends exactly P = SR_NW * SR_RPW here (384 x 384 inputs).
  const __amdgpu_buffer_rsrc_t zsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sc), 0, (int)((size_t)P * ldl * 4), 0x00020000);
  unsigned kof[SR_KC];
#pragma unroll
  for (int c = 0; c < SR_KC; ++c) kof[c] = 4u * (unsigned)min(lane + 64 * c, K - 1);
  float z[SR_RPW][SR_KC];
#pragma unroll
  for (int r = 0; r < SR_RPW; ++r)
#pragma unroll
    for (int c = 0; c < SR_KC; ++c)
      z[r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zsrc, kof[c], (p0 + r) * ldl * 4, 0)) * inv_eps;
  for (int i = tid; i < SR_NW * SR_RPW; i += 64 * SR_NW) s_v[i] = 0.f;
  if (tid < KW) s_u[tid] = 0.f;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    // ---- u = log_mu - logsumexp_p(Z + v)
    float m[SR_KC];
#pragma unroll
    for (int c = 0; c < SR_KC; ++c) m[c] = -INFINITY;
#pragma unroll
    for (int r = 0; r < SR_RPW; ++r) {
      const float vr = s_v[p0 + r];
#pragma unroll
      for (int c = 0; c < SR_KC; ++c) m[c] = fmaxf(m[c], z[r][c] + vr);
      if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);       // bounds what the scheduler hoists: the plan already fills the register file
    }
#pragma unroll
    for (int c = 0; c < SR_KC; ++c) s_part[wave][lane + 64 * c] = m[c];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < SR_KC; ++c) {
      float mm = -INFINITY;
#pragma unroll
      for (int w = 0; w < SR_NW; ++w) mm = fmaxf(mm, s_part[w][lane + 64 * c]);
      m[c] = mm;
    }
    float t[SR_KC];
#pragma unroll
    for (int c = 0; c < SR_KC; ++c) t[c] = 0.f;
#pragma unroll
    for (int r = 0; r < SR_RPW; ++r) {
      const float vr = s_v[p0 + r];
#pragma unroll
      for (int c = 0; c < SR_KC; ++c) t[c] += __builtin_amdgcn_exp2f(z[r][c] + vr - m[c]);
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                          // every wave has read the maxima: the partial buffer can take the sums
#pragma unroll
    for (int c = 0; c < SR_KC; ++c) s_part[wave][lane + 64 * c] = t[c];
    __syncthreads();
    if (tid < KW) {
      float tt = 0.f;
#pragma unroll
      for (int w = 0; w < SR_NW; ++w) tt += s_part[w][tid];      // fixed order
      float mk = -INFINITY;                                        // this thread's column is tid = lane + 64 * wave: its maximum again
      mk = m[0];
      if (wave == 1) mk = m[1];
      if (wave == 2) mk = m[2];
      s_u[tid] = norm - (mk + __builtin_amdgcn_logf(tt));
    }
    __syncthreads();
    // ---- v = log_nu - logsumexp_k(Z + u): a token row is one wave's registers
    float uk[SR_KC];
#pragma unroll
    for (int c = 0; c < SR_KC; ++c) uk[c] = (lane + 64 * c < K) ? s_u[lane + 64 * c] : -INFINITY;
#pragma unroll
    for (int r = 0; r < SR_RPW; ++r) {
      float a[SR_KC], mm = -INFINITY;
#pragma unroll
      for (int c = 0; c < SR_KC; ++c) {
        a[c] = z[r][c] + uk[c];
        mm = fmaxf(mm, a[c]);
      }
      mm = wave_max_valu(mm);
      float ts = 0.f;
#pragma unroll
      for (int c = 0; c < SR_KC; ++c) ts += __builtin_amdgcn_exp2f(a[c] - mm);       // exp2(-inf) = 0 for the padding columns
      ts = wave_sum_valu(ts);
      if (lane == 0) s_v[p0 + r] = norm - (mm + __builtin_amdgcn_logf(ts));
      if ((r & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  // W = exp(Z + u + v - norm): token-major for the merge, cluster-major for Soft_Assignment_Maps
  // stores through range-checked buffers: a column past K gets an offset beyond the buffer and is dropped
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(wt + ((size_t)b * N + 1) * ldl, 0, (int)((size_t)P * ldl * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t ssrc =
      __builtin_amdgcn_make_buffer_rsrc(soft != nullptr ? soft + (size_t)b * K * P : wt, 0, soft != nullptr ? (int)((size_t)K * P * 4) : 0, 0x00020000);
  float uk[SR_KC];
  unsigned wof[SR_KC], sof[SR_KC];
#pragma unroll
  for (int c = 0; c < SR_KC; ++c) {
    const int k = lane + 64 * c;
    uk[c] = s_u[min(k, KW - 1)];
    wof[c] = k < K ? 4u * (unsigned)k : 0x80000000u;
    sof[c] = k < K ? 4u * (unsigned)k * (unsigned)P : 0x80000000u;
  }
#pragma unroll
  for (int r = 0; r < SR_RPW; ++r) {
    const float vr = s_v[p0 + r];
#pragma unroll
    for (int c = 0; c < SR_KC; ++c) {
      const float w = __builtin_amdgcn_exp2f(z[r][c] + uk[c] + vr - norm);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, w), wsrc, wof[c], (p0 + r) * ldl * 4, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, w), ssrc, sof[c], (p0 + r) * 4, 0);
    }
    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace

extern "C" int tr_pool_broadcast(void* h, int is_f32, int B, int N, int C, float eps, tr_stream_t s) {
  TR_REQUIRE(h, TR_ERR_NULL, "tr_pool_broadcast: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && C >= 2 && C % 2 == 0, TR_ERR_SHAPE, "tr_pool_broadcast: bad shape B=%d N=%d C=%d", B, N, C);
  const dim3 grid((C / 2 + 63) / 64, B);
  hipStream_t st = static_cast<hipStream_t>(s);
  if (is_f32) hipLaunchKernelGGL(pool_broadcast_kernel<true>, grid, dim3(256), 0, st, h, N, C, eps);
  else hipLaunchKernelGGL(pool_broadcast_kernel<false>, grid, dim3(256), 0, st, h, N, C, eps);
  TR_CHECK_LAUNCH("tr_pool_broadcast");
  return TR_OK;
}

extern "C" int tr_dyvit_score(const void* h, int is_f32, const float* w, const float* bias, float* scores, int M, int C,
                              tr_stream_t s) {
  TR_REQUIRE(h && w && bias && scores, TR_ERR_NULL, "tr_dyvit_score: null pointer");
  TR_REQUIRE(M > 0 && C > 0, TR_ERR_SHAPE, "tr_dyvit_score: bad shape M=%d C=%d", M, C);
  hipStream_t st = static_cast<hipStream_t>(s);
  const int nblocks = (M + 15) / 16;
  if (is_f32) hipLaunchKernelGGL(dyvit_score_kernel<true>, dim3(nblocks), dim3(256), 0, st, h, w, bias, scores, M, C);
  else hipLaunchKernelGGL(dyvit_score_kernel<false>, dim3(nblocks), dim3(256), 0, st, h, w, bias, scores, M, C);
  TR_CHECK_LAUNCH("tr_dyvit_score");
  return TR_OK;
}

extern "C" int tr_softassign_merge(const float* logits, int ldl, float scale, const float* x, const float* src, float* x_out,
                                   float* soft, int B, int N, int K, int D, tr_stream_t s);

extern "C" int tr_sit_merge(const float* logits, int ldl, float scale, const float* x, float* x_out, float* soft, int B, int N,
                            int K, int D, tr_stream_t s) {
  return tr_softassign_merge(logits, ldl, scale, x, x, x_out, soft, B, N, K, D, s);
}

extern "C" int tr_softassign_merge(const float* logits, int ldl, float scale, const float* x, const float* src, float* x_out,
                                   float* soft, int B, int N, int K, int D, tr_stream_t s) {
  TR_REQUIRE(logits && x && src && x_out, TR_ERR_NULL, "tr_sit_merge: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && K >= 1 && D >= 1 && D <= 256 * SJ_MAX && ldl >= K, TR_ERR_SHAPE,
             "tr_sit_merge: bad shape B=%d N=%d K=%d D=%d ldl=%d (D <= %d)", B, N, K, D, ldl, 256 * SJ_MAX);
  TR_REQUIRE(x_out != x && x_out != src, TR_ERR_SHAPE, "tr_sit_merge: needs a distinct x_out");
  const size_t lds = (size_t)(N - 1) * SKC * sizeof(float);
  TR_REQUIRE(lds <= 150 * 1024, TR_ERR_SHAPE, "tr_sit_merge: %d tokens need %zu B of LDS (max 150 KiB)", N - 1, lds);
  hipStream_t st = static_cast<hipStream_t>(s);
  const dim3 grid((K + SKC - 1) / SKC, B);
#define TR_SIT_LAUNCH(J)                                                                                                          \
  do {                                                                                                                            \
    TR_RESERVE_LDS(reinterpret_cast<const void*>(sit_merge_kernel<J, true>), lds, "tr_sit_merge");                                \
    hipLaunchKernelGGL((sit_merge_kernel<J, true>), grid, dim3(256), lds, st, logits, ldl, scale, x, src, x_out, soft, N, K, D);  \
  } while (0)
  switch ((D + 255) / 256) {
    case 1: TR_SIT_LAUNCH(1); break;
    case 2: TR_SIT_LAUNCH(2); break;
    case 3: TR_SIT_LAUNCH(3); break;
    default: TR_SIT_LAUNCH(4); break;
  }
#undef TR_SIT_LAUNCH
  TR_CHECK_LAUNCH("tr_sit_merge");
  return TR_OK;
}

extern "C" int tr_rownorm(const float* x, float* xh, void* xh_lp, int lp_is_f32, int M, int D, tr_stream_t s) {
  TR_REQUIRE(x && xh && xh_lp, TR_ERR_NULL, "tr_rownorm: null pointer");
  TR_REQUIRE(M > 0 && D > 0, TR_ERR_SHAPE, "tr_rownorm: bad shape M=%d D=%d", M, D);
  hipStream_t st = static_cast<hipStream_t>(s);
  if (lp_is_f32) hipLaunchKernelGGL(rownorm_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, st, x, xh, xh_lp, M, D);
  else hipLaunchKernelGGL(rownorm_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, st, x, xh, xh_lp, M, D);
  TR_CHECK_LAUNCH("tr_rownorm");
  return TR_OK;
}

extern "C" int tr_sinkhorn(const float* scores, int ldl, float eps, int iters, float* wt, float* soft, int B, int N, int K,
                           tr_stream_t s) {
  TR_REQUIRE(scores && wt, TR_ERR_NULL, "tr_sinkhorn: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && K >= 1 && ldl >= K && iters >= 0 && eps > 0.f, TR_ERR_SHAPE,
             "tr_sinkhorn: bad arguments B=%d N=%d K=%d ldl=%d iters=%d eps=%g", B, N, K, ldl, iters, (double)eps);
  const size_t lds = ((size_t)K * (N - 1) + K + (N - 1)) * sizeof(float);
  hipStream_t st = static_cast<hipStream_t>(s);
  if (lds > 158 * 1024) {
    static const bool regs_off = getenv("TR_SINKHORN_REGS_OFF") != nullptr;       // lab switch (same-box A/B)
    if (!regs_off && K <= 64 * SR_KC && N - 1 == SR_NW * SR_RPW) {                   // the plan fits the register file of one CU
      hipLaunchKernelGGL(sinkhorn_regs_kernel, dim3(B), dim3(64 * SR_NW), 0, st, scores, ldl, eps, iters, wt, soft, N, K);
      TR_CHECK_LAUNCH("tr_sinkhorn");
      return TR_OK;
    }
    TR_REQUIRE(K <= SG_KMAX, TR_ERR_SHAPE, "tr_sinkhorn: K=%d > %d centres with K*P beyond the LDS is not supported", K, SG_KMAX);
    const int kw = (K + 63) & ~63;
    const size_t lds_g = ((size_t)K + (N - 1) + (size_t)(SGT / kw) * kw) * sizeof(float);
    if (K <= 256) hipLaunchKernelGGL(sinkhorn_global_kernel<4>, dim3(B), dim3(SGT), lds_g, st, scores, ldl, eps, iters, wt, soft, N, K);
    else hipLaunchKernelGGL(sinkhorn_global_kernel<SG_KMAX / 64>, dim3(B), dim3(SGT), lds_g, st, scores, ldl, eps, iters, wt, soft, N, K);
    TR_CHECK_LAUNCH("tr_sinkhorn");
    return TR_OK;
  }
  TR_RESERVE_LDS(reinterpret_cast<const void*>(sinkhorn_kernel), lds, "tr_sinkhorn");
  hipLaunchKernelGGL(sinkhorn_kernel, dim3(B), dim3(SKT), lds, st, scores, ldl, eps, iters, wt, soft, N, K);
  TR_CHECK_LAUNCH("tr_sinkhorn");
  return TR_OK;
}

extern "C" int tr_weighted_merge(const float* wt, int ldl, const float* x, const float* src, float* x_out, int B, int N, int K,
                                 int D, tr_stream_t s) {
  TR_REQUIRE(wt && x && src && x_out, TR_ERR_NULL, "tr_weighted_merge: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && K >= 1 && D >= 1 && D <= 256 * SJ_MAX && ldl >= K, TR_ERR_SHAPE,
             "tr_weighted_merge: bad shape B=%d N=%d K=%d D=%d ldl=%d", B, N, K, D, ldl);
  TR_REQUIRE(x_out != x && x_out != src, TR_ERR_SHAPE, "tr_weighted_merge: needs a distinct x_out");
  const size_t lds = (size_t)(N - 1) * SKC * sizeof(float);
  TR_REQUIRE(lds <= 150 * 1024, TR_ERR_SHAPE, "tr_weighted_merge: %d tokens need %zu B of LDS (max 150 KiB)", N - 1, lds);
  hipStream_t st = static_cast<hipStream_t>(s);
  const dim3 grid((K + SKC - 1) / SKC, B);
#define TR_WM_LAUNCH(J)                                                                                                           \
  do {                                                                                                                            \
    TR_RESERVE_LDS(reinterpret_cast<const void*>(sit_merge_kernel<J, false>), lds, "tr_weighted_merge");                          \
    hipLaunchKernelGGL((sit_merge_kernel<J, false>), grid, dim3(256), lds, st, wt, ldl, 1.0f, x, src, x_out,                      \
                       static_cast<float*>(nullptr), N, K, D);                                                                    \
  } while (0)
  switch ((D + 255) / 256) {
    case 1: TR_WM_LAUNCH(1); break;
    case 2: TR_WM_LAUNCH(2); break;
    case 3: TR_WM_LAUNCH(3); break;
    default: TR_WM_LAUNCH(4); break;
  }
#undef TR_WM_LAUNCH
  TR_CHECK_LAUNCH("tr_weighted_merge");
  return TR_OK;
}

extern "C" int tr_softassign_merge_fast(float* logits, int ldl, float scale, int apply_softmax, const float* x, const float* src,
                                        float* x_out, float* soft, int B, int N, int K, int D, tr_stream_t s) {
  TR_REQUIRE(logits && x && src && x_out, TR_ERR_NULL, "tr_softassign_merge_fast: null pointer");
  TR_REQUIRE(B > 0 && N >= 2 && K >= 1 && D >= 4 && D % 4 == 0 && ldl >= K && ldl % 4 == 0, TR_ERR_SHAPE,
             "tr_softassign_merge_fast: bad shape B=%d N=%d K=%d D=%d ldl=%d", B, N, K, D, ldl);
  TR_REQUIRE(x_out != x && x_out != src, TR_ERR_SHAPE, "tr_softassign_merge_fast: needs a distinct x_out");
  TR_REQUIRE(tr_aligned16(logits) && tr_aligned16(x) && tr_aligned16(src) && tr_aligned16(x_out), TR_ERR_ALIGN,
             "tr_softassign_merge_fast: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  if (apply_softmax) {
    if (N - 1 <= 8 * 25) hipLaunchKernelGGL(token_softmax_kernel<25>, dim3((K + 31) / 32, B), dim3(256), 0, st, logits, ldl, scale, soft, N, K);       // 224^2 inputs
    else hipLaunchKernelGGL(token_softmax_kernel<0>, dim3((K + 31) / 32, B), dim3(256), 0, st, logits, ldl, scale, soft, N, K);
  }
  if (K > MK_MAX) {          // more outputs than the MFMA kernel's accumulators hold (384 x 384 inputs at high keep rates): the VALU merge
    TR_CHECK_LAUNCH("tr_softassign_merge_fast");
    return tr_weighted_merge(logits, ldl, x, src, x_out, B, N, K, D, s);
  }
  hipLaunchKernelGGL(softmerge_mfma_kernel, dim3(((D + 63) / 64) * B), dim3(256), 0, st, logits, ldl, x, src, x_out, N, K, D);
  TR_CHECK_LAUNCH("tr_softassign_merge_fast");
  return TR_OK;
}
