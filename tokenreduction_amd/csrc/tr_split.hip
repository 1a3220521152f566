// "bf16x3" precision: fp32 operands on the MATRIX CORES.  Every fp32 value v is split into hi = bf16(v) and lo = bf16(v - hi)
// (v = hi + lo to ~2^-17 relative), and a product a*w is taken as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi in the fp32 MFMA accumulator
// (the dropped lo*lo term is ~2^-18 relative).  Three MFMAs per product instead of one buy back the 8 bits a bf16 operand drops:
// the trunk then agrees with the reference's fp32 arithmetic to ~1e-5 relative per Linear -- enough for north_star's "logits within
// 1e-3 abs" and for bit-exact token decisions on all but knife-edge scores -- at MFMA speed instead of the VALU speed of the
// TR_PREC_FP32 validation twins (tr_fp32.hip).  TR_PREC_BF16X3 runs the fp32 executor (fp32 activations everywhere, every
// non-GEMM op is the fp32 twin) with these two kernels in place of tr_gemm_f32 / tr_attention_f32.
//   tr_gemm_split       nn.Linear  topk.py:44,52, timm Mlp fc1/fc2, head topk.py:203, PatchEmbed topk.py:181
//   tr_attention_split  softmax(q k^T * 64^-0.5 [+ log size]) v, CLS row, column sums   topk.py:44-51,59, tome.py:48-49
#include "tr_common.h"

namespace {

struct HiLo {
  unsigned hi, lo;   // packed bf16x2 each
};
__device__ __forceinline__ HiLo split2(float a, float b) {
  HiLo r;
  r.hi = pack_bf16x2(a, b);
  const float ah = __builtin_bit_cast(float, r.hi << 16), bh = __builtin_bit_cast(float, r.hi & 0xffff0000u);
  r.lo = pack_bf16x2(a - ah, b - bh);
  return r;
}
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ f32x4 mfma3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);     // small terms first
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// GEMM: out[M,N] = A[M,K] W[N,K]^T + bias, fp32 in / fp32 out.  128 x 128 tile per workgroup, four waves of 64 x 64, 32-deep K
// slabs.  The slab is fetched as fp32 into registers one step ahead, split into hi/lo on its way into LDS (rows of 32 bf16 =
// 64 B, padded to 80 B: the 16 rows a ds_read_b128 fragment read touches land on 16 distinct 4-bank groups).
constexpr int SBM = 128, SBN = 128, SBK = 32, SROW = 80;
constexpr int S_TILE = 128 * SROW;   // bytes of one hi or lo image of a 128-row operand slab

template <int EPI>
__global__ __launch_bounds__(256) void gemm_split_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         const float* __restrict__ aux, int aux_i, int M, int N, int K, int nNt) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[4 * S_TILE];   // A hi, A lo, W hi, W lo
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, frow = lane & 15, fq = lane >> 4;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / nNt) * SBM, n0 = (bid % nNt) * SBN;
  const int lrow = tid >> 1, lhalf = tid & 1;
  const float* ap = A + (size_t)min(m0 + lrow, M - 1) * K + lhalf * 16;
  const float* wp = W + (size_t)min(n0 + lrow, N - 1) * K + lhalf * 16;
  f32x4 ra[4], rw[4];
  auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ra[c] = *reinterpret_cast<const f32x4*>(ap + k0 + 4 * c);
      rw[c] = *reinterpret_cast<const f32x4*>(wp + k0 + 4 * c);
    }
  };
  auto put = [&](unsigned char* base, const f32x4 (&r)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const HiLo p0 = split2(r[2 * c][0], r[2 * c][1]), p1 = split2(r[2 * c][2], r[2 * c][3]);
      const HiLo p2 = split2(r[2 * c + 1][0], r[2 * c + 1][1]), p3 = split2(r[2 * c + 1][2], r[2 * c + 1][3]);
      unsigned char* d = base + lrow * SROW + (2 * lhalf + c) * 16;
      *reinterpret_cast<u32x4*>(d) = u32x4{p0.hi, p1.hi, p2.hi, p3.hi};
      *reinterpret_cast<u32x4*>(d + S_TILE) = u32x4{p0.lo, p1.lo, p2.lo, p3.lo};
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += SBK) {
    __syncthreads();                        // the previous slab has been read
    put(sm, ra);
    put(sm + 2 * S_TILE, rw);
    if (k0 + SBK < K) fetch(k0 + SBK);
    lds_barrier();
    bf16x8 wh[4], wl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned char* p = sm + 2 * S_TILE + (wn * 64 + i * 16 + frow) * SROW + fq * 16;
      wh[i] = *reinterpret_cast<const bf16x8*>(p);
      wl[i] = *reinterpret_cast<const bf16x8*>(p + S_TILE);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned char* p = sm + (wm * 64 + j * 16 + frow) * SROW + fq * 16;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(p);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(p + S_TILE);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = mfma3(wh[i], wl[i], ah, al, acc[i][j]);
    }
  }
  // accumulator (i, j): rows = output columns n0 + wn*64 + 16 i + 4 fq + r, column = token m0 + wm*64 + 16 j + frow
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + frow;
    if (m >= M) continue;
    size_t orow = (size_t)m;
    const float* posrow = nullptr;
    if (EPI == TR_EPI_PATCH_F32) {
      const int b = m / aux_i, p = m - b * aux_i;
      orow = (size_t)b * (aux_i + 1) + 1 + p;
      posrow = aux + (size_t)(1 + p) * N;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + 4 * fq;
      if (n >= N) continue;                 // N % 4 == 0 (launcher)
      f32x4 v = acc[i][j] + *reinterpret_cast<const f32x4*>(bias + n);
      if (EPI == TR_EPI_GELU_BF16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752f));   // nn.GELU(): exact erf
      }
      if (EPI == TR_EPI_PATCH_F32) v += *reinterpret_cast<const f32x4*>(posrow + n);
      *reinterpret_cast<f32x4*>(out + orow * N + n) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention, N <= 224: one workgroup per (image, head).  K (row-major) and V (transposed, key-contiguous) of the head live in LDS
// as hi/lo bf16 images; each wave takes 16 queries at a time with the whole score row in registers (exact row max / normaliser).
// S^T = K Q^T puts the key on the accumulator row, so a lane holds keys {16 t + 4 fq + r} of query frow; the P.V product walks the
// keys in that same order (two key tiles per 32-deep MFMA step), which makes the softmax registers the MFMA operand directly.
constexpr int A_MAXT = 14;            // key tiles of 16: N <= 224
constexpr int A_KROW = 144;           // 64 bf16 = 128 B + 16 B pad

template <int NT, bool COLSUM>
__global__ __launch_bounds__(256) void attention_split_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                              float* __restrict__ cls_rows, const float* __restrict__ size,
                                                              float* __restrict__ colsum_part, int N, int H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NP = NT * 16;
  constexpr int VROW = NP * 2 + 8;      // one d row of V^T: NP bf16 + 8 B pad
  unsigned char* sKh = smem;
  unsigned char* sKl = sKh + NP * A_KROW;
  unsigned char* sVh = sKl + NP * A_KROW;
  unsigned char* sVl = sVh + 64 * VROW;
  float* sBias = reinterpret_cast<float*>(sVl + 64 * VROW);   // [NP] log size or -inf for padded keys
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int ldq = 3 * H * 64;
  const float* base = qkv + (size_t)b * N * ldq;
  const int qcol = h * 64, kcol = H * 64 + h * 64, vcol = 2 * H * 64 + h * 64;
  for (int e = tid; e < NP * 16; e += 256) {
    const int key = e >> 4, d4 = (e & 15) * 4;
    f32x4 kv = f32x4{0.f, 0.f, 0.f, 0.f}, vv = kv;
    if (key < N) {
      kv = *reinterpret_cast<const f32x4*>(base + (size_t)key * ldq + kcol + d4);
      vv = *reinterpret_cast<const f32x4*>(base + (size_t)key * ldq + vcol + d4);
    }
    const HiLo k0 = split2(kv[0], kv[1]), k1 = split2(kv[2], kv[3]);
    *reinterpret_cast<u32x2*>(sKh + key * A_KROW + d4 * 2) = u32x2{k0.hi, k1.hi};
    *reinterpret_cast<u32x2*>(sKl + key * A_KROW + d4 * 2) = u32x2{k0.lo, k1.lo};
    const HiLo v0 = split2(vv[0], vv[1]), v1 = split2(vv[2], vv[3]);
    unsigned short* th = reinterpret_cast<unsigned short*>(sVh);
    unsigned short* tl = reinterpret_cast<unsigned short*>(sVl);
    const int o = key;
    th[(d4 + 0) * (VROW / 2) + o] = (unsigned short)(v0.hi & 0xffff);
    th[(d4 + 1) * (VROW / 2) + o] = (unsigned short)(v0.hi >> 16);
    th[(d4 + 2) * (VROW / 2) + o] = (unsigned short)(v1.hi & 0xffff);
    th[(d4 + 3) * (VROW / 2) + o] = (unsigned short)(v1.hi >> 16);
    tl[(d4 + 0) * (VROW / 2) + o] = (unsigned short)(v0.lo & 0xffff);
    tl[(d4 + 1) * (VROW / 2) + o] = (unsigned short)(v0.lo >> 16);
    tl[(d4 + 2) * (VROW / 2) + o] = (unsigned short)(v1.lo & 0xffff);
    tl[(d4 + 3) * (VROW / 2) + o] = (unsigned short)(v1.lo >> 16);
  }
  for (int key = tid; key < NP; key += 256)
    sBias[key] = key < N ? (size ? logf(size[(size_t)b * N + key]) : 0.f) : -INFINITY;
  __syncthreads();

  float colacc[COLSUM ? NT : 1][4];
#pragma unroll
  for (int t = 0; t < (COLSUM ? NT : 1); ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) colacc[t][r] = 0.f;

  const int nqt = (N + 15) >> 4;
  for (int qt = wave; qt < nqt; qt += 4) {
    const int q = qt * 16 + frow;
    const bool qok = q < N;
    // Q fragment: row q, head dims 32 ks + 8 fq .. +8, hi / lo
    bf16x8 qh[2], ql[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, c = a;
      if (qok) {
        a = *reinterpret_cast<const f32x4*>(base + (size_t)q * ldq + qcol + 32 * ks + 8 * fq);
        c = *reinterpret_cast<const f32x4*>(base + (size_t)q * ldq + qcol + 32 * ks + 8 * fq + 4);
      }
      const HiLo p0 = split2(a[0], a[1]), p1 = split2(a[2], a[3]), p2 = split2(c[0], c[1]), p3 = split2(c[2], c[3]);
      qh[ks] = __builtin_bit_cast(bf16x8, u32x4{p0.hi, p1.hi, p2.hi, p3.hi});
      ql[ks] = __builtin_bit_cast(bf16x8, u32x4{p0.lo, p1.lo, p2.lo, p3.lo});
    }
    f32x4 s[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned char* p = sKh + (t * 16 + frow) * A_KROW + (4 * ks + fq) * 16;
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(p);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(p + NP * A_KROW);
        a = mfma3(kh, kl, qh[ks], ql[ks], a);
      }
      const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + t * 16 + 4 * fq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[r] = a[r] * 0.125f + bv[r];       // (q @ k^T) * 64^-0.5 [+ size.log(), tome.py:48-49]; -inf on padded keys
        mx = fmaxf(mx, a[r]);
      }
      s[t] = a;
      __builtin_amdgcn_sched_barrier(0);    // keep the K fragments of later tiles from being hoisted (register pressure)
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] = expf(s[t][r] - mx);
        l += s[t][r];
      }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] *= inv;                     // attn = attn.softmax(-1)
        if (COLSUM && qok) colacc[COLSUM ? t : 0][r] += s[t][r];
      }
    if (cls_rows != nullptr && qt == 0 && frow == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = t * 16 + 4 * fq + r;
          if (key < N) cls_rows[((size_t)b * H + h) * N + key] = s[t][r];
        }
    }
    // O^T[d][q] = sum_key V^T[d][key] P[q][key]: 32 keys per step = tiles 2u, 2u+1, lane slot order [tile 2u: 4 fq + r | tile 2u+1: 4 fq + r]
    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < (NT + 1) / 2; ++u) {
      const f32x4 pa = s[2 * u];
      const f32x4 pb = (2 * u + 1 < NT) ? s[2 * u + 1] : f32x4{0.f, 0.f, 0.f, 0.f};
      const HiLo p0 = split2(pa[0], pa[1]), p1 = split2(pa[2], pa[3]), p2 = split2(pb[0], pb[1]), p3 = split2(pb[2], pb[3]);
      const bf16x8 ph = __builtin_bit_cast(bf16x8, u32x4{p0.hi, p1.hi, p2.hi, p3.hi});
      const bf16x8 pl = __builtin_bit_cast(bf16x8, u32x4{p0.lo, p1.lo, p2.lo, p3.lo});
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned char* p = sVh + (i * 16 + frow) * VROW + (32 * u + 4 * fq) * 2;
        u32x2 a0 = *reinterpret_cast<const u32x2*>(p), a1 = u32x2{0u, 0u};
        u32x2 b0 = *reinterpret_cast<const u32x2*>(p + 64 * VROW), b1 = u32x2{0u, 0u};
        if (2 * u + 1 < NT) {
          a1 = *reinterpret_cast<const u32x2*>(p + 32);
          b1 = *reinterpret_cast<const u32x2*>(p + 64 * VROW + 32);
        }
        const bf16x8 vh = __builtin_bit_cast(bf16x8, u32x4{a0[0], a0[1], a1[0], a1[1]});
        const bf16x8 vl = __builtin_bit_cast(bf16x8, u32x4{b0[0], b0[1], b1[0], b1[1]});
        o[i] = mfma3(vh, vl, ph, pl, o[i]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (qok) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(out + ((size_t)b * N + q) * (H * 64) + h * 64 + i * 16 + 4 * fq) = o[i];
    }
  }
  if (COLSUM) {                             // column sums of the softmax matrix over this wave's queries (kmedoids.py:240)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = colacc[COLSUM ? t : 0][r];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        const int key = t * 16 + 4 * fq + r;
        if (frow == 0 && key < N) colsum_part[(((size_t)b * H + h) * 4 + wave) * N + key] = v;
      }
  }
}

template <int NT, bool COLSUM>
int launch_attention_split_(const float* qkv, float* out, float* cls_rows, const float* size, float* colsum_part, int B, int N, int H,
                           hipStream_t st) {
  constexpr int NP = NT * 16;
  const size_t lds = (size_t)2 * NP * A_KROW + (size_t)2 * 64 * (NP * 2 + 8) + (size_t)NP * 4;
  TR_RESERVE_LDS(reinterpret_cast<const void*>(attention_split_kernel<NT, COLSUM>), lds, "tr_attention_split");
  hipLaunchKernelGGL((attention_split_kernel<NT, COLSUM>), dim3(B * H), dim3(256), lds, st, qkv, out, cls_rows, size, colsum_part, N, H);
  return TR_OK;
}
template <int NT>
int launch_attention_split(const float* qkv, float* out, float* cls_rows, const float* size, float* colsum_part, int B, int N, int H,
                           hipStream_t st) {
  return colsum_part ? launch_attention_split_<NT, true>(qkv, out, cls_rows, size, colsum_part, B, N, H, st)
                     : launch_attention_split_<NT, false>(qkv, out, cls_rows, size, colsum_part, B, N, H, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention beyond 224 tokens (384^2 inputs: N = 577; round 4 -- until then bf16x3 fell back to the fp32 VALU kernel here: 14 ms per
// launch, 79 % of a 53-ms DeiT-B forward at B = 64).  K and V hi/lo of a head no longer fit the LDS, so the keys are walked in chunks of
// 128, TWICE per group of 128 queries (8 waves x 16): pass 1 finds every query's row maximum and normaliser (running max / rescaled
// sum), pass 2 recomputes the scores, turns them into FINAL probabilities p = exp(s - m) / l and feeds P.V -- so the CLS row and the
// column sums are exact side outputs, as in the one-chunk kernel above, whose fragment layouts and arithmetic this reuses.  One
// workgroup per (image, head) walks the query groups; column sums are kept per wave in LDS and leave as the four rows the consumer adds.
constexpr int AL_CH = 128, AL_CT = AL_CH / 16;        // keys per chunk, 16-key tiles per chunk
constexpr int AL_VROW = AL_CH * 2 + 8;                 // one d row of a V^T chunk: 128 bf16 + 8 B pad
constexpr int AL_NW = 8;                               // waves

template <bool COLSUM>
__global__ __launch_bounds__(64 * AL_NW, 1) void attention_split_long_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                            float* __restrict__ cls_rows, const float* __restrict__ size,
                                                                            float* __restrict__ colsum_part, int N, int H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sKh = smem;
  unsigned char* sKl = sKh + AL_CH * A_KROW;
  unsigned char* sVh = sKl + AL_CH * A_KROW;
  unsigned char* sVl = sVh + 64 * AL_VROW;
  float* sBias = reinterpret_cast<float*>(sVl + 64 * AL_VROW);   // [AL_CH] log size, or -inf for keys past N
  float* sCol = sBias + AL_CH;                                     // COLSUM: [AL_NW][nch * AL_CH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int ldq = 3 * H * 64;
  const float* base = qkv + (size_t)b * N * ldq;
  const int qcol = h * 64, kcol = H * 64 + h * 64, vcol = 2 * H * 64 + h * 64;
  const int nch = (N + AL_CH - 1) / AL_CH, NPk = nch * AL_CH;
  if (COLSUM)
    for (int i = tid; i < AL_NW * NPk; i += 64 * AL_NW) sCol[i] = 0.f;

  // K (and V) of chunk c into the LDS images; every thread passes both barriers
  auto stage = [&](int c, bool with_v) __attribute__((always_inline)) {
    __syncthreads();                               // the previous chunk's fragment reads are done
    for (int e = tid; e < AL_CH * 16; e += 64 * AL_NW) {
      const int kl_ = e >> 4, d4 = (e & 15) * 4, key = c * AL_CH + kl_;
      f32x4 kv = f32x4{0.f, 0.f, 0.f, 0.f}, vv = kv;
      if (key < N) {
        kv = *reinterpret_cast<const f32x4*>(base + (size_t)key * ldq + kcol + d4);
        if (with_v) vv = *reinterpret_cast<const f32x4*>(base + (size_t)key * ldq + vcol + d4);
      }
      const HiLo k0 = split2(kv[0], kv[1]), k1 = split2(kv[2], kv[3]);
      *reinterpret_cast<u32x2*>(sKh + kl_ * A_KROW + d4 * 2) = u32x2{k0.hi, k1.hi};
      *reinterpret_cast<u32x2*>(sKl + kl_ * A_KROW + d4 * 2) = u32x2{k0.lo, k1.lo};
      if (with_v) {
        const HiLo v0 = split2(vv[0], vv[1]), v1 = split2(vv[2], vv[3]);
        unsigned short* th = reinterpret_cast<unsigned short*>(sVh);
        unsigned short* tl = reinterpret_cast<unsigned short*>(sVl);
        th[(d4 + 0) * (AL_VROW / 2) + kl_] = (unsigned short)(v0.hi & 0xffff);
        th[(d4 + 1) * (AL_VROW / 2) + kl_] = (unsigned short)(v0.hi >> 16);
        th[(d4 + 2) * (AL_VROW / 2) + kl_] = (unsigned short)(v1.hi & 0xffff);
        th[(d4 + 3) * (AL_VROW / 2) + kl_] = (unsigned short)(v1.hi >> 16);
        tl[(d4 + 0) * (AL_VROW / 2) + kl_] = (unsigned short)(v0.lo & 0xffff);
        tl[(d4 + 1) * (AL_VROW / 2) + kl_] = (unsigned short)(v0.lo >> 16);
        tl[(d4 + 2) * (AL_VROW / 2) + kl_] = (unsigned short)(v1.lo & 0xffff);
        tl[(d4 + 3) * (AL_VROW / 2) + kl_] = (unsigned short)(v1.lo >> 16);
      }
    }
    for (int kl_ = tid; kl_ < AL_CH; kl_ += 64 * AL_NW) {
      const int key = c * AL_CH + kl_;
      sBias[kl_] = key < N ? (size ? logf(size[(size_t)b * N + key]) : 0.f) : -INFINITY;
    }
    __syncthreads();
  };

  const int nqg = (N + 16 * AL_NW - 1) / (16 * AL_NW);
  for (int qg = 0; qg < nqg; ++qg) {
    const int q = (qg * AL_NW + wave) * 16 + frow;
    const bool qok = q < N;
    bf16x8 qh[2], ql[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, c = a;
      if (qok) {
        a = *reinterpret_cast<const f32x4*>(base + (size_t)q * ldq + qcol + 32 * ks + 8 * fq);
        c = *reinterpret_cast<const f32x4*>(base + (size_t)q * ldq + qcol + 32 * ks + 8 * fq + 4);
      }
      const HiLo p0 = split2(a[0], a[1]), p1 = split2(a[2], a[3]), p2 = split2(c[0], c[1]), p3 = split2(c[2], c[3]);
      qh[ks] = __builtin_bit_cast(bf16x8, u32x4{p0.hi, p1.hi, p2.hi, p3.hi});
      ql[ks] = __builtin_bit_cast(bf16x8, u32x4{p0.lo, p1.lo, p2.lo, p3.lo});
    }
    // scores of the staged chunk: s[t][r] = key 16 t + 4 fq + r of the chunk, query frow -- scaled, key bias added (-inf past N)
    f32x4 s[AL_CT];
    auto scores = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < AL_CT; ++t) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const unsigned char* p = sKh + (t * 16 + frow) * A_KROW + (4 * ks + fq) * 16;
          const bf16x8 kh = *reinterpret_cast<const bf16x8*>(p);
          const bf16x8 kl = *reinterpret_cast<const bf16x8*>(p + AL_CH * A_KROW);
          a = mfma3(kh, kl, qh[ks], ql[ks], a);
        }
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + t * 16 + 4 * fq);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = a[r] * 0.125f + bv[r];
        s[t] = a;
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // ---- pass 1: row maximum and normaliser
    float m = -INFINITY, l = 0.f;                  // l: this lane's share (its 4 keys of every tile); summed over fq at the end
    for (int c = 0; c < nch; ++c) {
      stage(c, false);
      scores();
      float cm = -INFINITY;
#pragma unroll
      for (int t = 0; t < AL_CT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) cm = fmaxf(cm, s[t][r]);
      cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
      cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
      const float mn = fmaxf(m, cm);
      float add = 0.f;
      if (mn != -INFINITY) {
#pragma unroll
        for (int t = 0; t < AL_CT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) add += expf(s[t][r] - mn);
        l = l * expf(m - mn) + add;                // exp(-inf - mn) = 0 on the first chunk
      }
      m = mn;
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // ---- pass 2: final probabilities, side outputs, P.V
    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nch; ++c) {
      stage(c, true);
      scores();
#pragma unroll
      for (int t = 0; t < AL_CT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[t][r] = expf(s[t][r] - m) * inv;       // attn = attn.softmax(-1); 0 on keys past N
      if (cls_rows != nullptr && q == 0) {
#pragma unroll
        for (int t = 0; t < AL_CT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = c * AL_CH + t * 16 + 4 * fq + r;
            if (key < N) cls_rows[((size_t)b * H + h) * N + key] = s[t][r];
          }
      }
      if (COLSUM) {                                // this wave's 16 queries, summed over the frow lanes; one LDS row per wave
#pragma unroll
        for (int t = 0; t < AL_CT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = qok ? s[t][r] : 0.f;
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            if (frow == 0) sCol[wave * NPk + c * AL_CH + t * 16 + 4 * fq + r] += v;
          }
      }
#pragma unroll
      for (int u = 0; u < AL_CT / 2; ++u) {
        const f32x4 pa = s[2 * u], pb = s[2 * u + 1];
        const HiLo p0 = split2(pa[0], pa[1]), p1 = split2(pa[2], pa[3]), p2 = split2(pb[0], pb[1]), p3 = split2(pb[2], pb[3]);
        const bf16x8 ph = __builtin_bit_cast(bf16x8, u32x4{p0.hi, p1.hi, p2.hi, p3.hi});
        const bf16x8 pl = __builtin_bit_cast(bf16x8, u32x4{p0.lo, p1.lo, p2.lo, p3.lo});
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned char* p = sVh + (i * 16 + frow) * AL_VROW + (32 * u + 4 * fq) * 2;
          const u32x2 a0 = *reinterpret_cast<const u32x2*>(p), a1 = *reinterpret_cast<const u32x2*>(p + 32);
          const u32x2 b0 = *reinterpret_cast<const u32x2*>(p + 64 * AL_VROW), b1 = *reinterpret_cast<const u32x2*>(p + 64 * AL_VROW + 32);
          const bf16x8 vh = __builtin_bit_cast(bf16x8, u32x4{a0[0], a0[1], a1[0], a1[1]});
          const bf16x8 vl = __builtin_bit_cast(bf16x8, u32x4{b0[0], b0[1], b1[0], b1[1]});
          o[i] = mfma3(vh, vl, ph, pl, o[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (qok) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(out + ((size_t)b * N + q) * (H * 64) + h * 64 + i * 16 + 4 * fq) = o[i];
    }
  }
  if (COLSUM) {                                    // rows w and w + 4 of the per-wave sums -> the consumer's four rows (fixed order)
    __syncthreads();
    for (int i = tid; i < 4 * N; i += 64 * AL_NW) {
      const int w = i / N, key = i - w * N;
      colsum_part[(((size_t)b * H + h) * 4 + w) * N + key] = sCol[w * NPk + key] + sCol[(w + 4) * NPk + key];
    }
  }
}

}  // namespace

extern "C" int tr_gemm_split(const float* A, const float* W, const float* bias, float* out, const float* aux, int aux_i, int M,
                             int N, int K, int epilogue, tr_stream_t s) {
  TR_REQUIRE(A && W && bias && out, TR_ERR_NULL, "tr_gemm_split: null pointer");
  TR_REQUIRE(M > 0 && N > 0 && K > 0, TR_ERR_SHAPE, "tr_gemm_split: M,N,K must be positive (got %d,%d,%d)", M, N, K);
  if (K % SBK != 0 || N % 4 != 0)           // shapes the MFMA tiling does not cover: the VALU twin computes the same Linear
    return tr_gemm_f32(A, W, bias, out, aux, aux_i, M, N, K, epilogue, s);
  TR_REQUIRE(tr_aligned16(A) && tr_aligned16(W) && tr_aligned16(bias) && tr_aligned16(out), TR_ERR_ALIGN,
             "tr_gemm_split: pointers must be 16-byte aligned");
  if (epilogue == TR_EPI_PATCH_F32)
    TR_REQUIRE(aux && aux_i > 0 && M % aux_i == 0 && tr_aligned16(aux), TR_ERR_SHAPE,
               "tr_gemm_split: PATCH epilogue needs pos_embed and P | M");
  const int nMt = (M + SBM - 1) / SBM, nNt = (N + SBN - 1) / SBN;
  hipStream_t st = static_cast<hipStream_t>(s);
  tr_prof_note("gemm_split", 6.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
  switch (epilogue) {
    case TR_EPI_F32: hipLaunchKernelGGL(gemm_split_kernel<TR_EPI_F32>, dim3(nMt * nNt), dim3(256), 0, st, A, W, bias, out, aux, aux_i, M, N, K, nNt); break;
    case TR_EPI_GELU_BF16: hipLaunchKernelGGL(gemm_split_kernel<TR_EPI_GELU_BF16>, dim3(nMt * nNt), dim3(256), 0, st, A, W, bias, out, aux, aux_i, M, N, K, nNt); break;
    case TR_EPI_PATCH_F32: hipLaunchKernelGGL(gemm_split_kernel<TR_EPI_PATCH_F32>, dim3(nMt * nNt), dim3(256), 0, st, A, W, bias, out, aux, aux_i, M, N, K, nNt); break;
    default: TR_REQUIRE(false, TR_ERR_SHAPE, "tr_gemm_split: epilogue must be TR_EPI_F32, TR_EPI_GELU_BF16 (= GELU, fp32 out) or TR_EPI_PATCH_F32");
  }
  TR_CHECK_LAUNCH("tr_gemm_split");
  return TR_OK;
}

extern "C" int tr_attention_split(const float* qkv, float* out, float* cls_rows, const float* size, float* colsum_part, int B, int N,
                                  int H, tr_stream_t s) {
  TR_REQUIRE(qkv && out, TR_ERR_NULL, "tr_attention_split: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 1, TR_ERR_SHAPE, "tr_attention_split: need B, H, N >= 1");
  TR_REQUIRE(tr_aligned16(qkv) && tr_aligned16(out), TR_ERR_ALIGN, "tr_attention_split: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  if (N > 16 * A_MAXT) {                     // K/V hi+lo of a head no longer fit the LDS: keys in chunks of 128, two passes
    // up to 1024 keys (the length tests/test_hip_split.py covers, with column sums and CLS rows); beyond: the fp32 VALU kernel, which
    // itself takes N <= 640 and raises above -- one limit, stated in models.py's docstring
    if (N > 1024) return tr_attention_f32(qkv, out, cls_rows, size, colsum_part, B, N, H, s);
    tr_prof_note("attention_split_long", 18.0 * B * H * (double)N * N * 64, 16.0 * B * N * H * 64);
    const int nch = (N + AL_CH - 1) / AL_CH;
    const size_t lds = (size_t)2 * AL_CH * A_KROW + (size_t)2 * 64 * AL_VROW + (size_t)AL_CH * 4 + (colsum_part ? (size_t)AL_NW * nch * AL_CH * 4 : 0);
    if (colsum_part) {
      TR_RESERVE_LDS(reinterpret_cast<const void*>(attention_split_long_kernel<true>), lds, "tr_attention_split");
      hipLaunchKernelGGL(attention_split_long_kernel<true>, dim3(B * H), dim3(64 * AL_NW), lds, st, qkv, out, cls_rows, size, colsum_part, N, H);
    } else {
      TR_RESERVE_LDS(reinterpret_cast<const void*>(attention_split_long_kernel<false>), lds, "tr_attention_split");
      hipLaunchKernelGGL(attention_split_long_kernel<false>, dim3(B * H), dim3(64 * AL_NW), lds, st, qkv, out, cls_rows, size, colsum_part, N, H);
    }
    TR_CHECK_LAUNCH("tr_attention_split");
    return TR_OK;
  }
  tr_prof_note("attention_split", 12.0 * B * H * (double)N * N * 64, 16.0 * B * N * H * 64);
  const int nt = (N + 15) / 16;
  int rc;
  if (nt <= 5) rc = launch_attention_split<5>(qkv, out, cls_rows, size, colsum_part, B, N, H, st);
  else if (nt <= 7) rc = launch_attention_split<7>(qkv, out, cls_rows, size, colsum_part, B, N, H, st);
  else if (nt <= 9) rc = launch_attention_split<9>(qkv, out, cls_rows, size, colsum_part, B, N, H, st);
  else if (nt <= 11) rc = launch_attention_split<11>(qkv, out, cls_rows, size, colsum_part, B, N, H, st);
  else if (nt <= 13) rc = launch_attention_split<13>(qkv, out, cls_rows, size, colsum_part, B, N, H, st);
  else rc = launch_attention_split<14>(qkv, out, cls_rows, size, colsum_part, B, N, H, st);
  if (rc != TR_OK) return rc;
  TR_CHECK_LAUNCH("tr_attention_split");
  return TR_OK;
}
