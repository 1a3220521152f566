// Backward kernels of the DeiT block for gfx950 (the training half of the hot path: engine.py:50-76 fwd -> loss -> backward).
//
// The reference's backward is whatever torch.autograd derives from the eager forward (topk.py:83-99, timm Mlp, nn.LayerNorm,
// nn.Linear); these kernels are the hand-written gradients of the SAME forward the HIP executor runs:
//   tr_wgrad_bf16        dW[N,K] (+)= dY[M,N]^T X[M,K]        (nn.Linear weight gradient; MFMA, both operands read TRANSPOSED
//                                                              from row-major LDS images with ds_read_b64_tr_b16, split over M)
//   tr_colsum_bf16       db[N]  (+)= sum_m dY[m,n]             (nn.Linear bias gradient)
//   tr_gelu_bf16 / tr_gelu_bwd_bf16                            (timm Mlp act: the training forward keeps the pre-activation)
//   tr_layernorm_bwd     g (+)= dLN(dy; x, gamma), d_gamma, d_beta, bf16 copy of g for the next GEMMs; optional row scatter
//                        (the backward of the Top-K gather topk.py:89-93 and of EViT's fused token evit.py:111-123)
//   tr_head_bwd          classifier nn.Linear on the CLS rows (topk.py:203)
//   tr_embed_bwd         d pos_embed / d cls_token (topk.py:183-186)
//   tr_evit_fuse_bwd     gradient of extra = sum_j x[compl_j] * cls_attn[compl_j] (evit.py:117-120) wrt x rows and wrt cls_attn
//   tr_tome_merge_bwd    gradient of merge_wavg (tome.py:309-323) wrt x
// dgrad (dX = dY W) reuses tr_gemm_bf16 on a transposed copy of W.  All reductions over tokens are two-stage with fixed
// summation order (per-workgroup partials + one reduce kernel): results are bitwise reproducible run to run, no float atomics.
#include "tr_common.h"
#include "tr_rowops.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// two transposed 4x16 block reads -> one 16x16x32 MFMA operand (8 bf16: k = 8*(lane>>4) + e, row/col = lane&15); cdna guide T10
__device__ __forceinline__ bf16x8 lds_tr_pair(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
  const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient.  Output tile 128 (dW rows = dY columns) x 128 (dW columns = X columns); the token dimension is walked in
// slabs of 64 rows and split over blockIdx.y.  LDS image of a slab: [64 tokens][128 columns] bf16 = 256-B rows, 16-byte chunk
// index XOR ((row&3)<<2 | (row>>2)&3) -- conflict-free for the transposed reads (cdna guide T10, image (b)).
constexpr int WB = 128, WM = 64;
__device__ __forceinline__ int wswz(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// BIAS: the workgroups of the first column tile (k0 == 0) also sum their 128 dY columns over their token range -- the bias gradient
// db[n] = sum_m dY[m][n] -- from the registers the slab passes through anyway (bpart[split][N]; a separate column-sum kernel re-read
// dY: 1.0 of a 14 ms training step).
template <bool BIAS>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const uint16_t* __restrict__ Y, long ldy, int yskip,
                                                       const uint16_t* __restrict__ X, long ldx, float* __restrict__ part,
                                                       float* __restrict__ bpart, int M, int N, int K, int nNt, int sps) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[2][2][WM * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = (blockIdx.x % nNt) * WB, k0 = (blockIdx.x / nNt) * WB;
  const int nslab = (M + WM - 1) / WM;
  const int s_begin = blockIdx.y * sps, s_end = min(nslab, s_begin + sps);

  int srow[4], sch[4];
  size_t ycol[4], xcol[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int c = tid + 256 * it;
    srow[it] = c >> 4;
    sch[it] = c & 15;
    ycol[it] = (size_t)min(n0 + sch[it] * 8, N - 8);     // out-of-range columns: clamped (their outputs are never stored)
    xcol[it] = (size_t)min(k0 + sch[it] * 8, K - 8);
  }
  uint4 yreg[4], xreg[4];
  auto load_slab = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int m = s * WM + srow[it];
      const int mc = min(m, M - 1);
      const size_t my = yskip > 0 ? (size_t)mc + (size_t)(mc / yskip) + 1 : (size_t)mc;   // patch rows of a [B, P+1, D] tensor
      yreg[it] = *reinterpret_cast<const uint4*>(Y + my * ldy + ycol[it]);
      xreg[it] = *reinterpret_cast<const uint4*>(X + (size_t)mc * ldx + xcol[it]);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it)
      if (s * WM + srow[it] >= M) {          // rows past the end contribute zero
        yreg[it] = make_uint4(0u, 0u, 0u, 0u);
        xreg[it] = make_uint4(0u, 0u, 0u, 0u);
      }
  };
  const bool do_bias = BIAS && k0 == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // this thread's 8 columns (chunk tid & 15), its 4 rows of every slab
  auto write_slab = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      *reinterpret_cast<uint4*>(&sm[buf][0][wswz(srow[it], sch[it])]) = yreg[it];
      *reinterpret_cast<uint4*>(&sm[buf][1][wswz(srow[it], sch[it])]) = xreg[it];
    }
    if (do_bias) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const unsigned int w[4] = {yreg[it].x, yreg[it].y, yreg[it].z, yreg[it].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bsum[2 * e] += __uint_as_float(w[e] << 16);
          bsum[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
        }
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  if (s_begin < s_end) {
    load_slab(s_begin);
    write_slab(0);
  }
  __syncthreads();
  for (int s = s_begin; s < s_end; ++s) {
    const int buf = (s - s_begin) & 1;
    if (s + 1 < s_end) load_slab(s + 1);           // global loads of the next slab fly under this slab's MFMAs
    const unsigned char* sy = &sm[buf][0][0];
    const unsigned char* sx = &sm[buf][1][0];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = 32 * ks + 8 * g + q;          // this lane's row of the first 4-row block; the second is 4 rows on
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = (wn * 64 + i * 16) / 8 + (p >> 1);
        af[i] = lds_tr_pair(sy + wswz(r0, ch) + 8 * (p & 1), sy + wswz(r0 + 4, ch) + 8 * (p & 1));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ch = (wk * 64 + j * 16) / 8 + (p >> 1);
        bf[j] = lds_tr_pair(sx + wswz(r0, ch) + 8 * (p & 1), sx + wswz(r0 + 4, ch) + 8 * (p & 1));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < s_end) write_slab(buf ^ 1);
    __syncthreads();
  }
  if (do_bias) {
    // 16 threads (tid >> 4) hold partial sums of the same 8 columns: combined through LDS in thread order (fixed order)
    float* sb = reinterpret_cast<float*>(&sm[0][0][0]);           // [16][128]; the slab buffers are free after the loop's last barrier
#pragma unroll
    for (int e = 0; e < 8; ++e) sb[(tid >> 4) * 128 + (tid & 15) * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 128) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += sb[r * 128 + tid];
      if (n0 + tid < N) bpart[(size_t)blockIdx.y * N + n0 + tid] = a;
    }
  }
  // partial of this token range: part[split][n][k]; accumulator element r of tile (i,j) = row n 4g+r, column k lane&15
  float* po = part + (size_t)blockIdx.y * N * K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wk * 64 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + 4 * g + r;
        if (n < N && k < K) po[(size_t)n * K + k] = acc[i][j][r];
      }
    }
}

// dst[e] = (accumulate ? dst[e] : 0) + sum_s part[s][e].  Block = 64 float4 chunks x 4 partial-lanes: lane y sums partials
// y, y+4, ... (independent loads, 4 in flight), the four lane sums are combined in lane order -- a fixed order, so the result is
// bitwise reproducible.  (One thread per chunk walking all S partials was a chain of up to 1024 dependent round trips: 53 % of a
// training step in the first profile.)
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ part, int S, size_t count, float* __restrict__ dst,
                                                             int accumulate) {
  __shared__ float4 red[3][64];
  const int cx = threadIdx.x & 63, y = threadIdx.x >> 6;
  const size_t e = ((size_t)blockIdx.x * 64 + cx) * 4;
  const bool full = e + 4 <= count, any = e < count;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (full) {
    int s = y;
    for (; s + 12 < S; s += 16) {
      const float4 v0 = *reinterpret_cast<const float4*>(part + (size_t)s * count + e);
      const float4 v1 = *reinterpret_cast<const float4*>(part + (size_t)(s + 4) * count + e);
      const float4 v2 = *reinterpret_cast<const float4*>(part + (size_t)(s + 8) * count + e);
      const float4 v3 = *reinterpret_cast<const float4*>(part + (size_t)(s + 12) * count + e);
      a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < S; s += 4) {
      const float4 v = *reinterpret_cast<const float4*>(part + (size_t)s * count + e);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  } else if (any) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = y; s < S; s += 4)
      for (int i = 0; i < 4; ++i)
        if (e + i < count) t[i] += part[(size_t)s * count + e + i];
    a = make_float4(t[0], t[1], t[2], t[3]);
  }
  if (y > 0) red[y - 1][cx] = a;
  __syncthreads();
  if (y > 0 || !any) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const float4 v = red[w][cx];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (full) {
    if (accumulate) {
      const float4 d = *reinterpret_cast<const float4*>(dst + e);
      a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
    }
    *reinterpret_cast<float4*>(dst + e) = a;
  } else {
    const float t[4] = {a.x, a.y, a.z, a.w};
    for (int i = 0; i < 4; ++i)
      if (e + i < count) dst[e + i] = (accumulate ? dst[e + i] : 0.f) + t[i];
  }
}

// the same for TWO results in one launch (a Linear's weight and bias gradient, a LayerNorm's d_gamma and d_beta): blocks
// 0 .. nb0-1 reduce (part0, count0, dst0), the others (part1, count1, dst1); counts are multiples of 4
__global__ __launch_bounds__(256) void partial_reduce2_kernel(const float* __restrict__ part0, size_t count0, float* __restrict__ dst0,
                                                              const float* __restrict__ part1, size_t count1, float* __restrict__ dst1,
                                                              int S, int nb0, int accumulate) {
  __shared__ float4 red[3][64];
  const bool second = (int)blockIdx.x >= nb0;
  const float* part = second ? part1 : part0;
  const size_t count = second ? count1 : count0;
  float* dst = second ? dst1 : dst0;
  const int cx = threadIdx.x & 63, y = threadIdx.x >> 6;
  const size_t e = ((size_t)(second ? blockIdx.x - nb0 : blockIdx.x) * 64 + cx) * 4;
  const bool ok = e + 4 <= count;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
    int s = y;
    for (; s + 12 < S; s += 16) {
      const float4 v0 = *reinterpret_cast<const float4*>(part + (size_t)s * count + e);
      const float4 v1 = *reinterpret_cast<const float4*>(part + (size_t)(s + 4) * count + e);
      const float4 v2 = *reinterpret_cast<const float4*>(part + (size_t)(s + 8) * count + e);
      const float4 v3 = *reinterpret_cast<const float4*>(part + (size_t)(s + 12) * count + e);
      a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < S; s += 4) {
      const float4 v = *reinterpret_cast<const float4*>(part + (size_t)s * count + e);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  if (y > 0) red[y - 1][cx] = a;
  __syncthreads();
  if (y > 0 || !ok) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const float4 v = red[w][cx];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (accumulate) {
    const float4 d = *reinterpret_cast<const float4*>(dst + e);
    a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
  }
  *reinterpret_cast<float4*>(dst + e) = a;
}

// TALL variant of the two-result reduce: few elements (a LayerNorm's d_gamma / d_beta: D each), many partials (one per workgroup of the
// producing kernel: 512).  Block = 16 float4 chunks x 16 partial-lanes: lane y sums partials y, y + 16, ... four loads in flight, the
// sixteen lane sums are added in lane order (fixed).  With the 64 x 4 layout above the same job was 4 workgroups walking 128 partials each
// in 32 dependent round trips: 13 us, 26 times per training step.
__global__ __launch_bounds__(256) void partial_reduce2_tall_kernel(const float* __restrict__ part0, size_t count0, float* __restrict__ dst0,
                                                                   const float* __restrict__ part1, size_t count1, float* __restrict__ dst1,
                                                                   int S, int nb0, int accumulate) {
  __shared__ float4 red[15][16];
  const bool second = (int)blockIdx.x >= nb0;
  const float* part = second ? part1 : part0;
  const size_t count = second ? count1 : count0;
  float* dst = second ? dst1 : dst0;
  const int cx = threadIdx.x & 15, y = threadIdx.x >> 4;
  const size_t e = ((size_t)(second ? blockIdx.x - nb0 : blockIdx.x) * 16 + cx) * 4;
  const bool ok = e + 4 <= count;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
    int s = y;
    for (; s + 48 < S; s += 64) {
      const float4 v0 = *reinterpret_cast<const float4*>(part + (size_t)s * count + e);
      const float4 v1 = *reinterpret_cast<const float4*>(part + (size_t)(s + 16) * count + e);
      const float4 v2 = *reinterpret_cast<const float4*>(part + (size_t)(s + 32) * count + e);
      const float4 v3 = *reinterpret_cast<const float4*>(part + (size_t)(s + 48) * count + e);
      a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < S; s += 16) {
      const float4 v = *reinterpret_cast<const float4*>(part + (size_t)s * count + e);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  if (y > 0) red[y - 1][cx] = a;
  __syncthreads();
  if (y > 0 || !ok) return;
#pragma unroll
  for (int w = 0; w < 15; ++w) {
    const float4 v = red[w][cx];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (accumulate) {
    const float4 d = *reinterpret_cast<const float4*>(dst + e);
    a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
  }
  *reinterpret_cast<float4*>(dst + e) = a;
}

// The LayerNorm parameter gradients of a whole backward pass, reduced in ONE launch: every tr_layernorm_bwd of the pass leaves its
// [grid][D] partials of d gamma and d beta in its own slice of a region and adds two segments here; the executor flushes after its
// last block.  The per-call partial_reduce2_tall launch was 4.8 us x 24 per DeiT-B training step for a few hundred KB each.  Same block
// layout and the same fixed summation order as partial_reduce2_tall_kernel, so the sums are bit-identical to the undeferred ones.
constexpr int LN_DEFER_SEGS = 32;
struct TallSegs {
  const float* part[LN_DEFER_SEGS];
  float* dst[LN_DEFER_SEGS];
  int count[LN_DEFER_SEGS], S[LN_DEFER_SEGS], acc[LN_DEFER_SEGS];
  int blk0[LN_DEFER_SEGS + 1];          // first block of segment i; blk0[n] = blocks in all
};
__global__ __launch_bounds__(256) void partial_reduce_tall_segs_kernel(const TallSegs T) {
  __shared__ float4 red[15][16];
  // static indices only (a dynamically indexed kernel argument is copied to scratch): walk the table with selects
  const float* part = T.part[0];
  float* dst = T.dst[0];
  int count = T.count[0], S = T.S[0], accumulate = T.acc[0], b0 = 0;
#pragma unroll
  for (int i = 1; i < LN_DEFER_SEGS; ++i) {
    const bool in = (int)blockIdx.x >= T.blk0[i];
    part = in ? T.part[i] : part;
    dst = in ? T.dst[i] : dst;
    count = in ? T.count[i] : count;
    S = in ? T.S[i] : S;
    accumulate = in ? T.acc[i] : accumulate;
    b0 = in ? T.blk0[i] : b0;
  }
  const int cx = threadIdx.x & 15, y = threadIdx.x >> 4;
  const size_t e = ((size_t)(blockIdx.x - b0) * 16 + cx) * 4;
  const bool ok = e + 4 <= (size_t)count;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
    int s = y;
    for (; s + 48 < S; s += 64) {
      const float4 v0 = *reinterpret_cast<const float4*>(part + (size_t)s * count + e);
      const float4 v1 = *reinterpret_cast<const float4*>(part + (size_t)(s + 16) * count + e);
      const float4 v2 = *reinterpret_cast<const float4*>(part + (size_t)(s + 32) * count + e);
      const float4 v3 = *reinterpret_cast<const float4*>(part + (size_t)(s + 48) * count + e);
      a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < S; s += 16) {
      const float4 v = *reinterpret_cast<const float4*>(part + (size_t)s * count + e);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  if (y > 0) red[y - 1][cx] = a;
  __syncthreads();
  if (y > 0 || !ok) return;
#pragma unroll
  for (int w = 0; w < 15; ++w) {
    const float4 v = red[w][cx];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (accumulate) {
    const float4 d = *reinterpret_cast<const float4*>(dst + e);
    a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
  }
  *reinterpret_cast<float4*>(dst + e) = a;
}

// the same for up to EIGHT results with their own partial counts (the weight and bias gradients of up to four Linear layers whose weight
// gradients ran as one launch): block ranges [0, nb0), [nb0, nb0 + nb1), ...; counts are multiples of 4; unused segments have nb = 0
struct RSeg {
  const float* part;
  float* dst;
  size_t count;
  int S, nb;
};
struct RSegs {
  RSeg s[8];
};
template <typename T>
__device__ __forceinline__ T rsel8(int i, T a, T b, T c, T d, T e, T f, T g, T h) {      // by value (see qsel4 in tr_wgrad_pc.hip)
  return i >= 4 ? (i >= 6 ? (i == 7 ? h : g) : (i == 5 ? f : e)) : (i >= 2 ? (i == 3 ? d : c) : (i ? b : a));
}
__global__ __launch_bounds__(256) void partial_reduce8_kernel(const RSegs segs, int accumulate) {
  __shared__ float4 red[3][64];
  int b = blockIdx.x, seg = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int nb = segs.s[i].nb;
    if (seg == i && b >= nb) {
      b -= nb;
      seg = i + 1;
    }
  }
#define RS(f) rsel8(seg, segs.s[0].f, segs.s[1].f, segs.s[2].f, segs.s[3].f, segs.s[4].f, segs.s[5].f, segs.s[6].f, segs.s[7].f)
  const float* part = RS(part);
  float* dst = RS(dst);
  const size_t count = RS(count);
  const int S = RS(S);
#undef RS
  const int cx = threadIdx.x & 63, y = threadIdx.x >> 6;
  const size_t e = ((size_t)b * 64 + cx) * 4;
  const bool ok = e + 4 <= count;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
    int s = y;
    for (; s + 12 < S; s += 16) {
      const float4 v0 = *reinterpret_cast<const float4*>(part + (size_t)s * count + e);
      const float4 v1 = *reinterpret_cast<const float4*>(part + (size_t)(s + 4) * count + e);
      const float4 v2 = *reinterpret_cast<const float4*>(part + (size_t)(s + 8) * count + e);
      const float4 v3 = *reinterpret_cast<const float4*>(part + (size_t)(s + 12) * count + e);
      a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < S; s += 4) {
      const float4 v = *reinterpret_cast<const float4*>(part + (size_t)s * count + e);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  if (y > 0) red[y - 1][cx] = a;
  __syncthreads();
  if (y > 0 || !ok) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const float4 v = red[w][cx];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (accumulate) {
    const float4 d = *reinterpret_cast<const float4*>(dst + e);
    a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
  }
  *reinterpret_cast<float4*>(dst + e) = a;
}

// column sums of a bf16 matrix: grid (ceil(N/512), S); thread = 2 adjacent columns; part[s][n]
__global__ __launch_bounds__(256) void colsum_kernel(const uint16_t* __restrict__ Y, long ldy, int yskip, float* __restrict__ part, int M,
                                                     int N, int rows_per_split) {
  const int n = blockIdx.x * 512 + threadIdx.x * 2;
  if (n >= N) return;
  const int r0 = blockIdx.y * rows_per_split, r1 = min(M, r0 + rows_per_split);
  float a0 = 0.f, a1 = 0.f;
  int m = r0;
  for (; m + 4 <= r1; m += 4) {
    unsigned int u[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const size_t my = yskip > 0 ? (size_t)(m + t) + (size_t)((m + t) / yskip) + 1 : (size_t)(m + t);
      u[t] = *reinterpret_cast<const unsigned int*>(Y + my * ldy + n);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a0 += __uint_as_float(u[t] << 16);
      a1 += __uint_as_float(u[t] & 0xffff0000u);
    }
  }
  for (; m < r1; ++m) {
    const size_t my = yskip > 0 ? (size_t)m + (size_t)(m / yskip) + 1 : (size_t)m;
    const unsigned int u = *reinterpret_cast<const unsigned int*>(Y + my * ldy + n);
    a0 += __uint_as_float(u << 16);
    a1 += __uint_as_float(u & 0xffff0000u);
  }
  float* po = part + (size_t)blockIdx.y * N + n;
  po[0] = a0;
  po[1] = a1;
}

// ---- GELU: the training forward keeps fc1's pre-activation (bf16) and applies the SAME fit the fused eval epilogue uses
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const uint16_t* __restrict__ pre, uint16_t* __restrict__ h, size_t nchunks) {
  const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= nchunks) return;
  const uint4 u = *reinterpret_cast<const uint4*>(pre + 8 * c);
  const unsigned int w[4] = {u.x, u.y, u.z, u.w};
  unsigned int o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 v = gelu2(f32x2{__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)});
    o[i] = pack_bf16x2(v[0], v[1]);
  }
  *reinterpret_cast<uint4*>(h + 8 * c) = make_uint4(o[0], o[1], o[2], o[3]);
}

// dh *= gelu'(pre): the derivative of the fit the forward applied (gelu2_grad, tr_common.h)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const uint16_t* __restrict__ pre, uint16_t* __restrict__ dh, size_t nchunks) {
  const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= nchunks) return;
  const uint4 u = *reinterpret_cast<const uint4*>(pre + 8 * c);
  const uint4 d = *reinterpret_cast<const uint4*>(dh + 8 * c);
  const unsigned int w[4] = {u.x, u.y, u.z, u.w}, dd[4] = {d.x, d.y, d.z, d.w};
  unsigned int o[4];
  dgelu_line(dd, w, o);
  *reinterpret_cast<uint4*>(dh + 8 * c) = make_uint4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm backward, one wave per row (rows strided over the grid), fused with the residual stream's gradient:
//   xhat = (x - mean) * rstd;  dyg = dy * gamma;  dx = rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat))
//   g_new[row] = g_in[row] + dx   -> g_out (fp32) and gb_out (bf16: the dY operand of the next dgrad / wgrad GEMMs)
//   d_gamma += dy * xhat, d_beta += dy   (per-workgroup partials part[2][wg][D], reduced in a second kernel)
// Row scatter (idx != NULL): the LayerNorm ran on gathered rows [B, K+1(+1)] (topk.py:89-95); row r of image b goes to row
// (r == 0 ? 0 : 1 + idx[b,r-1]) of the [B, n_out] gradient, and EViT's fused row r == K+1 to g_fused[b] (fp32 [B, D]).
// ADD (K-Medoids: the gathered ids may repeat -- an empty cluster's medoid is token 0, kmedoids.py:74-79): the scattered rows are
// ADDED to the zero-filled destination with float atomics and no bf16 copy is written (the caller converts afterwards).
template <int NCH, bool ADD>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const uint16_t* __restrict__ dy, const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ g_in, long ldgi,
                                                     float* __restrict__ g_out, long ldgo, uint16_t* __restrict__ gb_out,
                                                     const int32_t* __restrict__ idx, int K, int n_in, int n_out,
                                                     float* __restrict__ g_fused, float* __restrict__ part, int M, int D, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunks = D >> 2;
  float4 gm[NCH], ag[NCH], ab[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    gm[c] = *reinterpret_cast<const float4*>(gamma + 4 * min(lane + 64 * c, nchunks - 1));
    ag[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    float4 v[NCH], d[NCH], gi[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = min(lane + 64 * c, nchunks - 1);
      v[c] = ln_nt_load4(x + (size_t)row * ldx + 4 * ch);
      d[c] = bf16x4_to_f32(*reinterpret_cast<const uint2*>(dy + (size_t)row * D + 4 * ch));
      gi[c] = g_in != nullptr ? ln_nt_load4(g_in + (size_t)row * ldgi + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (lane + 64 * c < nchunks) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    const float mean = wave_sum(s) / (float)D;
    float qv = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (lane + 64 * c < nchunks) {
        v[c].x -= mean; v[c].y -= mean; v[c].z -= mean; v[c].w -= mean;
        qv += (v[c].x * v[c].x + v[c].y * v[c].y) + (v[c].z * v[c].z + v[c].w * v[c].w);
      }
    const float rstd = rsqrtf(wave_sum(qv) / (float)D + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (lane + 64 * c < nchunks) {
        v[c].x *= rstd; v[c].y *= rstd; v[c].z *= rstd; v[c].w *= rstd;                       // xhat
        ag[c].x += d[c].x * v[c].x; ag[c].y += d[c].y * v[c].y; ag[c].z += d[c].z * v[c].z; ag[c].w += d[c].w * v[c].w;
        ab[c].x += d[c].x; ab[c].y += d[c].y; ab[c].z += d[c].z; ab[c].w += d[c].w;
        d[c].x *= gm[c].x; d[c].y *= gm[c].y; d[c].z *= gm[c].z; d[c].w *= gm[c].w;           // dy * gamma
        s1 += (d[c].x + d[c].y) + (d[c].z + d[c].w);
        s2 += (d[c].x * v[c].x + d[c].y * v[c].y) + (d[c].z * v[c].z + d[c].w * v[c].w);
      }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
    // destination row
    size_t orow = (size_t)row;
    bool fused = false;
    if (idx != nullptr) {
      const int b = row / n_in, r = row - b * n_in;
      if (r == K + 1) fused = true;
      else orow = (size_t)b * n_out + (r == 0 ? 0 : 1 + idx[(size_t)b * K + (r - 1)]);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunks) {
        float4 o;
        o.x = gi[c].x + rstd * (d[c].x - m1 - v[c].x * m2);
        o.y = gi[c].y + rstd * (d[c].y - m1 - v[c].y * m2);
        o.z = gi[c].z + rstd * (d[c].z - m1 - v[c].z * m2);
        o.w = gi[c].w + rstd * (d[c].w - m1 - v[c].w * m2);
        if (fused) {
          *reinterpret_cast<float4*>(g_fused + (size_t)(row / n_in) * D + 4 * ch) = o;
        } else if (ADD) {
          float* gp = g_out + orow * ldgo + 4 * ch;
          atomicAdd(gp, o.x); atomicAdd(gp + 1, o.y); atomicAdd(gp + 2, o.z); atomicAdd(gp + 3, o.w);
        } else {
          ln_nt_store4(o, g_out + orow * ldgo + 4 * ch);
          if (gb_out != nullptr) {
            uint2 pk;
            pk.x = pack_bf16x2(o.x, o.y);
            pk.y = pack_bf16x2(o.z, o.w);
            *reinterpret_cast<uint2*>(gb_out + orow * D + 4 * ch) = pk;
          }
        }
      }
    }
  }
  // per-workgroup partial of d_gamma / d_beta: waves combined in wave order
  __shared__ float4 red[3][2][64 * NCH];
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      red[wave - 1][0][lane + 64 * c] = ag[c];
      red[wave - 1][1][lane + 64 * c] = ab[c];
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunks) {
        float4 a = ag[c], b = ab[c];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float4 a2 = red[w][0][ch], b2 = red[w][1][ch];
          a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
          b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
        }
        *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * D + 4 * ch) = a;                          // part[0][wg][D]
        *reinterpret_cast<float4*>(part + ((size_t)gridDim.x + blockIdx.x) * D + 4 * ch) = b;            // part[1][wg][D]
      }
    }
  }
}

// ---- classifier (topk.py:203) backward on the CLS rows.  dxn[b][d] = sum_c dlogits[b][c] W[c][d]: grid (ceil(B/4), D/64); a block holds
// 64 columns (32 lanes x 2) of FOUR images and cuts the classes into 8 interleaved parts, combined through LDS in part order (C = 1000
// is not a multiple of the MFMA GEMM's K step; the product is tiny: B x C x D).  Every weight element a lane loads serves four images,
// five class steps are in flight per lane.  The weight gradient goes through tr_wgrad_bf16 / tr_colsum_bf16 on a bf16 copy of dlogits.
__global__ __launch_bounds__(256) void head_dx_kernel(const float* __restrict__ dlogits, const uint16_t* __restrict__ W, uint16_t* __restrict__ dxn,
                                                      int B, int C, int D) {
  __shared__ float red[7][4][64];
  const int b0 = blockIdx.x * 4, lane = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int d = min(blockIdx.y * 64 + 2 * lane, D - 2);
  const float* dl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dl[i] = dlogits + (size_t)min(b0 + i, B - 1) * C;
  float a[4][2] = {};
  int c = part;
  for (; c + 32 < C; c += 40) {
    unsigned w[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) w[u] = *reinterpret_cast<const unsigned*>(W + (size_t)(c + 8 * u) * D + d);
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const float w0 = __uint_as_float(w[u] << 16), w1 = __uint_as_float(w[u] & 0xffff0000u);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float g = dl[i][c + 8 * u];
        a[i][0] += g * w0;
        a[i][1] += g * w1;
      }
    }
  }
  for (; c < C; c += 8) {
    const unsigned w = *reinterpret_cast<const unsigned*>(W + (size_t)c * D + d);
    const float w0 = __uint_as_float(w << 16), w1 = __uint_as_float(w & 0xffff0000u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float g = dl[i][c];
      a[i][0] += g * w0;
      a[i][1] += g * w1;
    }
  }
  if (part > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[part - 1][i][2 * lane] = a[i][0];
      red[part - 1][i][2 * lane + 1] = a[i][1];
    }
  }
  __syncthreads();
  if (part > 0 || blockIdx.y * 64 + 2 * lane >= D) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (b0 + i >= B) break;
    float s0 = a[i][0], s1 = a[i][1];
#pragma unroll
    for (int p = 0; p < 7; ++p) {
      s0 += red[p][i][2 * lane];
      s1 += red[p][i][2 * lane + 1];
    }
    *reinterpret_cast<unsigned*>(dxn + (size_t)(b0 + i) * D + d) = pack_bf16x2(s0, s1);
  }
}

// ---- d pos_embed[n][:] = sum_b g[b][n][:], d cls_token = sum_b g[b][0][:]   (topk.py:183-186); block = 64 chunks x 4 batch lanes
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ g, float* __restrict__ dpos, float* __restrict__ dcls, int B,
                                                        int N, int D, int accumulate) {
  __shared__ float4 red[3][64];
  const int cx = threadIdx.x & 63, y = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + cx;
  const int nch = N * (D >> 2);
  const int tc = min(t, nch - 1);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = y; b < B; b += 4) {
    const float4 v = *reinterpret_cast<const float4*>(g + (size_t)b * N * D + 4 * (size_t)tc);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (y > 0) red[y - 1][cx] = a;
  __syncthreads();
  if (y > 0 || t >= nch) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const float4 v = red[w][cx];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  float4* o = reinterpret_cast<float4*>(dpos + 4 * (size_t)t);
  float4 r = a;
  if (accumulate) { const float4 p = *o; r.x += p.x; r.y += p.y; r.z += p.z; r.w += p.w; }
  *o = r;
  if (t < (D >> 2)) {
    float4* oc = reinterpret_cast<float4*>(dcls + 4 * (size_t)t);
    float4 rc = a;
    if (accumulate) { const float4 p = *oc; rc.x += p.x; rc.y += p.y; rc.z += p.z; rc.w += p.w; }
    *oc = rc;
  }
}

// ---- EViT fused token backward (evit.py:117-120): extra = sum_j xm[1+c_j] * s[c_j], xm = x + delta (post-attention stream).
// One workgroup per image; wave w takes complement tokens w, w+4, ...:
//   g_out[b, 1+c_j, :] = s[c_j] * g_fused[b]   (fp32 + bf16 copy; these rows get no other gradient: the token is dropped)
//   dscore[b, 1+c_j]   = <xm[1+c_j], g_fused[b]>   (d cls_attn, head-mean -> the attention backward divides by H)
template <int NCH>
__global__ __launch_bounds__(256) void evit_fuse_bwd_kernel(const float* __restrict__ x, const uint16_t* __restrict__ delta,
                                                            const int32_t* __restrict__ compl_idx, const float* __restrict__ scores,
                                                            const float* __restrict__ g_fused, float* __restrict__ g_out,
                                                            uint16_t* __restrict__ gb_out, float* __restrict__ dscore, int N, int K, int D) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunks = D >> 2, P = N - 1, nc = P - K;
  float4 gf[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) gf[c] = *reinterpret_cast<const float4*>(g_fused + (size_t)b * D + 4 * min(lane + 64 * c, nchunks - 1));
  for (int j = wave; j < nc; j += 4) {
    const int t = compl_idx[(size_t)b * nc + j];
    const float sc = scores[(size_t)b * P + t];
    const size_t rbase = ((size_t)b * N + 1 + t) * D;
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunks) {
        float4 xv = *reinterpret_cast<const float4*>(x + rbase + 4 * ch);
        if (delta != nullptr) {
          const float4 dv = bf16x4_to_f32(*reinterpret_cast<const uint2*>(delta + rbase + 4 * ch));
          xv.x += dv.x; xv.y += dv.y; xv.z += dv.z; xv.w += dv.w;
        }
        dot += (xv.x * gf[c].x + xv.y * gf[c].y) + (xv.z * gf[c].z + xv.w * gf[c].w);
        const float4 o = make_float4(sc * gf[c].x, sc * gf[c].y, sc * gf[c].z, sc * gf[c].w);
        *reinterpret_cast<float4*>(g_out + rbase + 4 * ch) = o;
        uint2 pk;
        pk.x = pack_bf16x2(o.x, o.y);
        pk.y = pack_bf16x2(o.z, o.w);
        *reinterpret_cast<uint2*>(gb_out + rbase + 4 * ch) = pk;
      }
    }
    dot = wave_sum(dot);
    if (lane == 0) dscore[(size_t)b * N + 1 + t] = dot;
  }
}

// ---- ToMe merge_wavg backward (tome.py:309-323): x_out[o] = sum_{i in o} x[i] size[i] / size_out[o]
//   g_in[b, i, :] = size_in[b,i] / size_out[b,o(i)] * g_out_rows[b, o(i), :]        (sizes carry no gradient)
// One wave per INPUT row; o(i): unmerged A-token -> its rank in unm_idx; B-token (odd position) -> (na - r) + (i >> 1);
// merged A-token (src) -> the slot of its dst B-token.  inv_map[b][i] is built by the first kernel.
__global__ __launch_bounds__(256) void tome_invmap_kernel(const int32_t* __restrict__ unm_idx, const int32_t* __restrict__ src_idx,
                                                          const int32_t* __restrict__ dst_idx, int32_t* __restrict__ inv_map, int N, int r) {
  const int b = blockIdx.x;
  const int na = (N + 1) >> 1, nb = N >> 1, nu = na - r;
  int32_t* im = inv_map + (size_t)b * N;
  for (int t = threadIdx.x; t < nu; t += 256) im[2 * unm_idx[(size_t)b * nu + t]] = t;
  for (int t = threadIdx.x; t < nb; t += 256) im[2 * t + 1] = nu + t;
  for (int t = threadIdx.x; t < r; t += 256) im[2 * src_idx[(size_t)b * r + t]] = nu + dst_idx[(size_t)b * r + t];
}
template <int NCH>
__global__ __launch_bounds__(256) void tome_merge_bwd_kernel(const float* __restrict__ g_merged, const float* __restrict__ size_in,
                                                             const float* __restrict__ size_out, const int32_t* __restrict__ inv_map,
                                                             float* __restrict__ g_out, uint16_t* __restrict__ gb_out, int N, int N_out,
                                                             int D, int M) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int b = row / N, i = row - b * N;
  const int o = inv_map[(size_t)b * N + i];
  const float w = (size_in != nullptr ? size_in[(size_t)b * N + i] : 1.0f) / size_out[(size_t)b * N_out + o];
  const int nchunks = D >> 2;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
      const float4 v = *reinterpret_cast<const float4*>(g_merged + ((size_t)b * N_out + o) * D + 4 * ch);
      const float4 ov = make_float4(w * v.x, w * v.y, w * v.z, w * v.w);
      *reinterpret_cast<float4*>(g_out + (size_t)row * D + 4 * ch) = ov;
      uint2 pk;
      pk.x = pack_bf16x2(ov.x, ov.y);
      pk.y = pack_bf16x2(ov.z, ov.w);
      *reinterpret_cast<uint2*>(gb_out + (size_t)row * D + 4 * ch) = pk;
    }
  }
}

// ---- DPC-KNN CTM backward (merge_tokens dpcknn.py:103-132 with token_weight = exp(score(x)), CTM.forward :155-157):
//   x_c = sum_{i in c} x_i w_i / W_c,  W_c = sum_{i in c} w_i + 1e-6,  w_i = exp(x_i . sw + sb)   (w_i = 1 with equal_weight)
//   d x_i = (w_i / W_c) g_c + dlog_i sw,   dlog_i = w_i <g_c, x_i - x_c> / W_c,   d sw = sum_i dlog_i x_i,   d sb = sum_i dlog_i
// (the clustering itself runs under no_grad, dpcknn.py:56).  One workgroup per image: thread c sums its cluster's weights in
// token order (fixed order), then one wave per token; the image's d sw / d sb partial goes to part[b][D+4].
template <int NCH>
__global__ __launch_bounds__(256) void cluster_merge_bwd_kernel(const float* __restrict__ g_in, const float* __restrict__ x0,
                                                                const float* __restrict__ x1, const float* __restrict__ wtok,
                                                                const int32_t* __restrict__ assign, const float* __restrict__ sw,
                                                                float* __restrict__ g_out, uint16_t* __restrict__ gb_out,
                                                                float* __restrict__ part, int N, int K, int D) {
  // grid (B, SPLIT): the SPLIT workgroups of an image take its tokens round robin (one per image left half of the CUs idle at B = 128 and
  // walked 49 dependent row gathers per wave: 112 us at DeiT-B); each recomputes the cluster weights, each writes its own d sw / d sb partial
  __shared__ float sW[640];          // summed token weight of every cluster (K <= 640: 384 x 384 inputs at keep_rate 0.9 have 518)
  __shared__ float4 red[3][64 * NCH];
  __shared__ float redb[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int P = N - 1, nchunks = D >> 2;
  const bool weighted = sw != nullptr;
  const int32_t* as = assign + (size_t)b * P;
  // cluster weights, summed in token order (fixed): assignments and token weights staged in LDS first when they fit (every built input
  // size: P <= 576) -- walking them in global memory, one dependent branch per token, was half of this kernel once the tokens were split
  __shared__ int sA[640];
  __shared__ float sT[640];
  const bool staged = P <= 640;
  if (staged) {
    for (int i = threadIdx.x; i < P; i += 256) {
      sA[i] = as[i];
      sT[i] = weighted ? wtok[(size_t)b * P + i] : 1.0f;
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < K; c += 256) {
    float a = 0.f;
    if (staged) {
      for (int i = 0; i < P; ++i) a += sA[i] == c ? sT[i] : 0.f;
    } else {
      for (int i = 0; i < P; ++i)
        if (as[i] == c) a += weighted ? wtok[(size_t)b * P + i] : 1.0f;
    }
    sW[c] = a + 1e-6f;
  }
  __syncthreads();
  float4 swv[NCH], acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    swv[c] = weighted ? *reinterpret_cast<const float4*>(sw + 4 * min(lane + 64 * c, nchunks - 1)) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float accb = 0.f;
  if (wave == 0 && blockIdx.y == 0) {      // CLS row passes through
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunks) {
        const float4 v = *reinterpret_cast<const float4*>(g_in + (size_t)b * (K + 1) * D + 4 * ch);
        *reinterpret_cast<float4*>(g_out + (size_t)b * N * D + 4 * ch) = v;
        uint2 pk;
        pk.x = pack_bf16x2(v.x, v.y);
        pk.y = pack_bf16x2(v.z, v.w);
        *reinterpret_cast<uint2*>(gb_out + (size_t)b * N * D + 4 * ch) = pk;
      }
    }
  }
  for (int i = blockIdx.y * 4 + wave; i < P; i += 4 * gridDim.y) {
    const int c = as[i];
    const float wi = weighted ? wtok[(size_t)b * P + i] : 1.0f;
    const float iw = 1.0f / sW[c];
    const float* gr = g_in + ((size_t)b * (K + 1) + 1 + c) * D;
    const float* xi = x0 + ((size_t)b * N + 1 + i) * D;
    const float* xc = x1 + ((size_t)b * (K + 1) + 1 + c) * D;
    float4 gv[NCH], xv[NCH];
    float dot = 0.f;
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
      const int ch = min(lane + 64 * cc, nchunks - 1);
      gv[cc] = *reinterpret_cast<const float4*>(gr + 4 * ch);
      xv[cc] = *reinterpret_cast<const float4*>(xi + 4 * ch);
      const float4 cv = *reinterpret_cast<const float4*>(xc + 4 * ch);
      if (lane + 64 * cc < nchunks)
        dot += (gv[cc].x * (xv[cc].x - cv.x) + gv[cc].y * (xv[cc].y - cv.y)) + (gv[cc].z * (xv[cc].z - cv.z) + gv[cc].w * (xv[cc].w - cv.w));
    }
    const float dlog = weighted ? wave_sum(dot) * iw * wi : 0.f;
    const float sc = wi * iw;
    accb += dlog;
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
      const int ch = lane + 64 * cc;
      if (ch < nchunks) {
        const float4 o = make_float4(sc * gv[cc].x + dlog * swv[cc].x, sc * gv[cc].y + dlog * swv[cc].y, sc * gv[cc].z + dlog * swv[cc].z,
                                     sc * gv[cc].w + dlog * swv[cc].w);
        acc[cc].x += dlog * xv[cc].x; acc[cc].y += dlog * xv[cc].y; acc[cc].z += dlog * xv[cc].z; acc[cc].w += dlog * xv[cc].w;
        *reinterpret_cast<float4*>(g_out + ((size_t)b * N + 1 + i) * D + 4 * ch) = o;
        uint2 pk;
        pk.x = pack_bf16x2(o.x, o.y);
        pk.y = pack_bf16x2(o.z, o.w);
        *reinterpret_cast<uint2*>(gb_out + ((size_t)b * N + 1 + i) * D + 4 * ch) = pk;
      }
    }
  }
  if (!weighted) return;
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) red[wave - 1][lane + 64 * c] = acc[c];
  }
  if (lane == 0) redb[wave] = accb;
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunks) {
        float4 a = acc[c];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float4 v = red[w][ch];
          a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        *reinterpret_cast<float4*>(part + ((size_t)b * gridDim.y + blockIdx.y) * (D + 4) + 4 * ch) = a;
      }
    }
    if (lane == 0) {
      float4 bs = make_float4((redb[0] + redb[1]) + (redb[2] + redb[3]), 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(part + ((size_t)b * gridDim.y + blockIdx.y) * (D + 4) + D) = bs;
    }
  }
}

// ---- ATS backward of the row sampling (ats.py:86,157): row t of the sampled tensors came from row ids[b,t] of the full ones.
// Valid rows (t == 0: CLS, or id != 0) are scattered -- ids are unique there --; padded rows replicate the CLS row in the forward
// but nothing downstream reads them (masked keys), so their gradient is zero and they are skipped.  The caller zero-fills the outputs.
template <int NCH>
__global__ __launch_bounds__(256) void ats_scatter_kernel(const float* __restrict__ g, const uint16_t* __restrict__ dao_s,
                                                          const int32_t* __restrict__ ids, float* __restrict__ g_full,
                                                          uint16_t* __restrict__ dao_full, int N, int Ks, int D, int M) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int b = row / Ks, t = row - b * Ks;
  const int id = ids[row];
  if (t > 0 && id == 0) return;
  const int nchunks = D >> 2;
  const size_t orow = (size_t)b * N + id;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
      *reinterpret_cast<float4*>(g_full + orow * D + 4 * ch) = *reinterpret_cast<const float4*>(g + (size_t)row * D + 4 * ch);
      *reinterpret_cast<uint2*>(dao_full + orow * D + 4 * ch) = *reinterpret_cast<const uint2*>(dao_s + (size_t)row * D + 4 * ch);
    }
  }
}

// DropPath (timm 0.4.12: x.div(keep_prob) * floor(keep_prob + rand), one draw per image; call sites topk.py:87,95): dst[b, r, :] =
// src[b, r, :] * scale[b] for bf16 rows; scale[b] is 0 or 1/keep_prob.  Applied to a branch's output before the residual add, and to
// the stream's gradient before it enters that branch's backward.  One 16-byte chunk per thread.
__global__ __launch_bounds__(256) void rowscale_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, const float* __restrict__ scale,
                                                       size_t nchunks, size_t chunks_per_image) {
  const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= nchunks) return;
  const float sc = scale[c / chunks_per_image];
  const uint4 u = *reinterpret_cast<const uint4*>(src + 8 * c);
  const unsigned int w[4] = {u.x, u.y, u.z, u.w};
  unsigned int o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(__uint_as_float(w[i] << 16) * sc, __uint_as_float(w[i] & 0xffff0000u) * sc);
  *reinterpret_cast<uint4*>(dst + 8 * c) = make_uint4(o[0], o[1], o[2], o[3]);
}

// fp32 rows -> bf16 rows (the initial gradient of the residual stream as a GEMM operand)
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t nchunks) {
  const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= nchunks) return;
  const float4 v = *reinterpret_cast<const float4*>(src + 4 * c);
  uint2 pk;
  pk.x = pack_bf16x2(v.x, v.y);
  pk.y = pack_bf16x2(v.z, v.w);
  *reinterpret_cast<uint2*>(dst + 4 * c) = pk;
}

inline int reduce_partials(const float* part, int S, size_t count, float* dst, int accumulate, hipStream_t st) {
  const unsigned nb = (unsigned)(((count + 3) / 4 + 63) / 64);
  hipLaunchKernelGGL(partial_reduce_kernel, dim3(nb), dim3(256), 0, st, part, S, count, dst, accumulate);
  return 0;
}

inline void reduce_partials2(const float* p0, size_t c0, float* d0, const float* p1, size_t c1, float* d1, int S, int accumulate, hipStream_t st) {
  if (S >= 64 && c0 <= 4096 && c1 <= 4096) {        // many partials of short vectors: spread the partials over the lanes
    const int nb0 = (int)((c0 / 4 + 15) / 16), nb1 = (int)((c1 / 4 + 15) / 16);
    hipLaunchKernelGGL(partial_reduce2_tall_kernel, dim3(nb0 + nb1), dim3(256), 0, st, p0, c0, d0, p1, c1, d1, S, nb0, accumulate);
    return;
  }
  const int nb0 = (int)((c0 / 4 + 63) / 64), nb1 = (int)((c1 / 4 + 63) / 64);
  hipLaunchKernelGGL(partial_reduce2_kernel, dim3(nb0 + nb1), dim3(256), 0, st, p0, c0, d0, p1, c1, d1, S, nb0, accumulate);
}

}  // namespace

// Token splits of a weight-gradient launch.  512 workgroups (2 per CU) are resident at a time, so tiles x S should fill whole rounds of
// 512; but every split writes an N x K fp32 partial that the reduce re-reads (DeiT-S: 37 splits of qkv's 1152 x 384 = 65 MB each way), so
// more rounds than necessary cost more than they balance.  Cost model (fitted on tools/lab/wgrad_target_ab.py: DeiT-S shapes prefer ONE
// round, the large DeiT-B matrices two): time ~ flops / (round efficiency x 500 TFLOP/s) + 8 S N K bytes / 4 TB/s.
static int wgrad_splits(int M, int N, int K) {
  const int tiles = ((N + WB - 1) / WB) * ((K + WB - 1) / WB);
  const int nslab = (M + WM - 1) / WM;
  int best = 1;
  double best_t = 1e30;
  for (int S = 1; S <= nslab && tiles * S <= 2048; ++S) {
    const int wg = tiles * S;
    const double eff = (double)wg / (double)(((wg + 511) / 512) * 512);
    const double t = 2.0 * M * N * K / (eff * 500e12) + 8.0 * S * N * K / 4e12;
    if (t < best_t) { best_t = t; best = S; }
  }
  return best;
}

// tr_wgrad_pc.hip: the producer/consumer weight-gradient kernel (192 x 192 tiles, LDS-DMA ring) for shapes whose N and K are multiples of 192
bool tr_wgrad_pc_fits(int M, int N, int K, long ldy, long ldx, int yskip);
int tr_wgrad_pc_splits(int M, int N, int K);
int tr_wgrad_pc_launch(const uint16_t* dY, long ldy, int yskip, const uint16_t* X, long ldx, float* part, float* bpart, int M, int N, int K, int S_max,
                       hipStream_t st);

extern "C" size_t tr_wgrad_workspace_floats(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int S = wgrad_splits(M, N, K);
  if (tr_wgrad_pc_fits(M, N, K, 8, 8, 0)) S = max(S, tr_wgrad_pc_splits(M, N, K));
  return (size_t)S * N * K + (size_t)S * N;         // weight partials, then the bias partials of tr_linear_bwd_params
}

extern "C" int tr_wgrad_bf16(const uint16_t* dY, long ldy, int yskip, const uint16_t* X, long ldx, float* dW, int accumulate, float* ws,
                             size_t ws_floats, int M, int N, int K, tr_stream_t s) {
  TR_REQUIRE(dY && X && dW && ws, TR_ERR_NULL, "tr_wgrad_bf16: null pointer");
  TR_REQUIRE(M > 0 && N >= 8 && K >= 8 && N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && ldy >= N && ldx >= K && yskip >= 0,
             TR_ERR_SHAPE, "tr_wgrad_bf16: need N,K,ldy,ldx multiples of 8 (M=%d N=%d K=%d ldy=%ld ldx=%ld)", M, N, K, ldy, ldx);
  TR_REQUIRE(tr_aligned16(dY) && tr_aligned16(X) && tr_aligned16(dW) && tr_aligned16(ws), TR_ERR_ALIGN, "tr_wgrad_bf16: pointers must be 16-byte aligned");
  const int nNt = (N + WB - 1) / WB, nKt = (K + WB - 1) / WB, tiles = nNt * nKt;
  const int nslab = (M + WM - 1) / WM;
  int S = wgrad_splits(M, N, K);
  const size_t fit = ws_floats / ((size_t)N * K);
  TR_REQUIRE(fit >= 1, TR_ERR_SHAPE, "tr_wgrad_bf16: workspace of %zu floats cannot hold one %d x %d partial", ws_floats, N, K);
  if ((size_t)S > fit) S = (int)fit;
  const int sps = (nslab + S - 1) / S;
  S = (nslab + sps - 1) / sps;                 // every split owns at least one slab
  hipStream_t st = static_cast<hipStream_t>(s);
  if (tr_wgrad_pc_fits(M, N, K, ldy, ldx, yskip)) {
    tr_prof_note("wgrad_pc_kernel", 2.0 * M * N * K, 2.0 * ((double)M * N + (double)M * K) + 4.0 * S * N * K);
    S = tr_wgrad_pc_launch(dY, ldy, yskip, X, ldx, ws, nullptr, M, N, K, (int)min(fit, (size_t)1 << 20), st);
  } else {
    tr_prof_note("wgrad_kernel", 2.0 * M * N * K, 2.0 * ((double)M * N + (double)M * K) + 4.0 * S * N * K);
    hipLaunchKernelGGL(wgrad_kernel<false>, dim3(tiles, S), dim3(256), 0, st, dY, ldy, yskip, X, ldx, ws, static_cast<float*>(nullptr), M, N, K, nNt, sps);
  }
  TR_CHECK_LAUNCH("tr_wgrad_bf16");
  reduce_partials(ws, S, (size_t)N * K, dW, accumulate, st);
  TR_CHECK_LAUNCH("tr_wgrad_bf16 (reduce)");
  return TR_OK;
}

// nn.Linear parameter gradients in one pass over dY: dW[N,K] (+)= dY^T X and db[N] (+)= column sums of dY (tr_wgrad_bf16 +
// tr_colsum_bf16 fused: the bias sums come out of the weight-gradient kernel's staging registers), one reduce launch for both.
extern "C" int tr_linear_bwd_params(const uint16_t* dY, long ldy, int yskip, const uint16_t* X, long ldx, float* dW, float* db, int accumulate,
                                    float* ws, size_t ws_floats, int M, int N, int K, tr_stream_t s) {
  TR_REQUIRE(dY && X && dW && db && ws, TR_ERR_NULL, "tr_linear_bwd_params: null pointer");
  TR_REQUIRE(M > 0 && N >= 8 && K >= 8 && N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && ldy >= N && ldx >= K && yskip >= 0,
             TR_ERR_SHAPE, "tr_linear_bwd_params: need N,K,ldy,ldx multiples of 8 (M=%d N=%d K=%d ldy=%ld ldx=%ld)", M, N, K, ldy, ldx);
  TR_REQUIRE(tr_aligned16(dY) && tr_aligned16(X) && tr_aligned16(dW) && tr_aligned16(db) && tr_aligned16(ws), TR_ERR_ALIGN,
             "tr_linear_bwd_params: pointers must be 16-byte aligned");
  const int nNt = (N + WB - 1) / WB, nKt = (K + WB - 1) / WB, tiles = nNt * nKt;
  const int nslab = (M + WM - 1) / WM;
  int S = wgrad_splits(M, N, K);
  const size_t per = (size_t)N * K + (size_t)N;
  const size_t fit = ws_floats / per;
  TR_REQUIRE(fit >= 1, TR_ERR_SHAPE, "tr_linear_bwd_params: workspace of %zu floats cannot hold one partial", ws_floats);
  if ((size_t)S > fit) S = (int)fit;
  const int sps = (nslab + S - 1) / S;
  S = (nslab + sps - 1) / sps;
  float* bpart = ws + (size_t)S * N * K;
  hipStream_t st = static_cast<hipStream_t>(s);
  if (tr_wgrad_pc_fits(M, N, K, ldy, ldx, yskip)) {
    // the bias partials sit behind the weight partials of the LARGEST split count the workspace admits (the launch may pick fewer)
    const int S_max = (int)min(fit, (size_t)1 << 20);
    bpart = ws + (size_t)S_max * N * K;
    tr_prof_note("wgrad_pc_kernel", 2.0 * M * N * K, 2.0 * ((double)M * N + (double)M * K) + 4.0 * S * N * K);
    S = tr_wgrad_pc_launch(dY, ldy, yskip, X, ldx, ws, bpart, M, N, K, S_max, st);
  } else {
    tr_prof_note("wgrad_kernel", 2.0 * M * N * K, 2.0 * ((double)M * N + (double)M * K) + 4.0 * S * N * K);
    hipLaunchKernelGGL(wgrad_kernel<true>, dim3(tiles, S), dim3(256), 0, st, dY, ldy, yskip, X, ldx, ws, bpart, M, N, K, nNt, sps);
  }
  TR_CHECK_LAUNCH("tr_linear_bwd_params");
  reduce_partials2(ws, (size_t)N * K, dW, bpart, (size_t)N, db, S, accumulate, st);
  TR_CHECK_LAUNCH("tr_linear_bwd_params (reduce)");
  return TR_OK;
}

void tr_wgrad_pc_group_splits(const int (*mnk)[3], int n, int* S);
bool tr_wgrad_pc_group_launch(const uint16_t* const* dY, const long* ldy, const uint16_t* const* X, const long* ldx, const int (*mnk)[3], int n,
                              float* ws, int* S, float** part, float** bpart, float* const* direct_w, float* const* direct_b, hipStream_t st);

// Parameter gradients of up to FOUR Linear layers in one weight-gradient launch and one reduce launch (the executor hands over fc2, fc1,
// proj and qkv of a block together, or pairs of them).  Each layer as in tr_linear_bwd_params (yskip = 0).  Groups the producer/consumer
// kernel does not take (an N or K that is not a multiple of 192, a workspace too small) run as separate tr_linear_bwd_params calls.
static bool group_fits(const tr_linear_grad* L, int n) {
  for (int i = 0; i < n; ++i)
    if (L[i].M <= 0 || L[i].N <= 0 || L[i].K <= 0 || !tr_wgrad_pc_fits(L[i].M, L[i].N, L[i].K, L[i].ldy, L[i].ldx, 0) || L[i].ldy < L[i].N ||
        L[i].ldx < L[i].K)
      return false;
  return true;
}

static size_t group_floats(const tr_linear_grad* L, int n, int* S) {
  int mnk[4][3];
  for (int i = 0; i < n; ++i) { mnk[i][0] = L[i].M; mnk[i][1] = L[i].N; mnk[i][2] = L[i].K; }
  tr_wgrad_pc_group_splits(mnk, n, S);
  size_t f = 0;
  for (int i = 0; i < n; ++i) f += (size_t)S[i] * ((size_t)L[i].N * L[i].K + L[i].N);
  return f;
}

extern "C" size_t tr_linear_bwd_group_workspace_floats(const tr_linear_grad* layers, int n) {
  if (!layers || n < 1 || n > 4) return 0;
  size_t single = 0;
  for (int i = 0; i < n; ++i) single = max(single, tr_wgrad_workspace_floats(layers[i].M, layers[i].N, layers[i].K));
  tr_linear_grad plain[4];
  for (int i = 0; i < n; ++i) { plain[i] = layers[i]; plain[i].ldy = layers[i].N; plain[i].ldx = layers[i].K; }
  if (!group_fits(plain, n)) return single;
  int S[4];
  return max(single, group_floats(plain, n, S));
}

extern "C" int tr_linear_bwd_group(const tr_linear_grad* layers, int n, int accumulate, float* ws, size_t ws_floats, tr_stream_t s) {
  TR_REQUIRE(layers && ws && n >= 1 && n <= 4, TR_ERR_NULL, "tr_linear_bwd_group: need 1..4 layers and a workspace");
  for (int i = 0; i < n; ++i)
    TR_REQUIRE(layers[i].dY && layers[i].X && layers[i].dW && layers[i].db, TR_ERR_NULL, "tr_linear_bwd_group: null pointer in layer %d", i);
  int S[4];
  bool grouped = n > 1 && group_fits(layers, n);
  if (grouped) grouped = group_floats(layers, n, S) <= ws_floats;
  if (!grouped) {
    for (int i = 0; i < n; ++i) {
      const tr_linear_grad& L = layers[i];
      const int rc = tr_linear_bwd_params(L.dY, L.ldy, 0, L.X, L.ldx, L.dW, L.db, accumulate, ws, ws_floats, L.M, L.N, L.K, s);
      if (rc != TR_OK) return rc;
    }
    return TR_OK;
  }
  const uint16_t* dY[4];
  const uint16_t* X[4];
  long ldy[4], ldx[4];
  int mnk[4][3];
  double flops = 0.0, bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    const tr_linear_grad& L = layers[i];
    TR_REQUIRE(tr_aligned16(L.dY) && tr_aligned16(L.X) && tr_aligned16(L.dW) && tr_aligned16(L.db), TR_ERR_ALIGN,
               "tr_linear_bwd_group: pointers of layer %d must be 16-byte aligned", i);
    dY[i] = L.dY; X[i] = L.X; ldy[i] = L.ldy; ldx[i] = L.ldx;
    mnk[i][0] = L.M; mnk[i][1] = L.N; mnk[i][2] = L.K;
    flops += 2.0 * L.M * L.N * L.K;
    bytes += 2.0 * ((double)L.M * L.N + (double)L.M * L.K);
  }
  TR_REQUIRE(tr_aligned16(ws), TR_ERR_ALIGN, "tr_linear_bwd_group: workspace must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  float* part[4];
  float* bpart[4];
  tr_prof_note("wgrad_pc_kernel", flops, bytes);
  float* dW[4];
  float* db[4];
  for (int i = 0; i < n; ++i) { dW[i] = layers[i].dW; db[i] = layers[i].db; }
  // overwriting callers with one token range per layer (the short late stages) get their results stored in place: no reduce launch
  const bool stored = tr_wgrad_pc_group_launch(dY, ldy, X, ldx, mnk, n, ws, S, part, bpart, accumulate ? nullptr : dW, accumulate ? nullptr : db, st);
  TR_CHECK_LAUNCH("tr_linear_bwd_group");
  if (stored) return TR_OK;
  RSegs segs;
  int nb = 0;
  for (int i = 0; i < 8; ++i) {
    const int l = i & 3, bias = i >> 2;        // segments: dW of layers 0..3, then db of layers 0..3
    RSeg& g = segs.s[i];
    if (l >= n) { g.part = nullptr; g.dst = nullptr; g.count = 0; g.S = 0; g.nb = 0; continue; }
    g.part = bias ? bpart[l] : part[l];
    g.dst = bias ? layers[l].db : layers[l].dW;
    g.count = bias ? (size_t)layers[l].N : (size_t)layers[l].N * layers[l].K;
    g.S = S[l];
    g.nb = (int)((g.count / 4 + 63) / 64);
    nb += g.nb;
  }
  hipLaunchKernelGGL(partial_reduce8_kernel, dim3(nb), dim3(256), 0, st, segs, accumulate);
  TR_CHECK_LAUNCH("tr_linear_bwd_group (reduce)");
  return TR_OK;
}

extern "C" size_t tr_linear_bwd_params2_workspace_floats(int M0, int N0, int K0, int M1, int N1, int K1) {
  const tr_linear_grad L[2] = {{nullptr, N0, nullptr, K0, nullptr, nullptr, M0, N0, K0}, {nullptr, N1, nullptr, K1, nullptr, nullptr, M1, N1, K1}};
  return tr_linear_bwd_group_workspace_floats(L, 2);
}

extern "C" int tr_linear_bwd_params2(const uint16_t* dY0, long ldy0, const uint16_t* X0, long ldx0, float* dW0, float* db0, int M0, int N0, int K0,
                                     const uint16_t* dY1, long ldy1, const uint16_t* X1, long ldx1, float* dW1, float* db1, int M1, int N1, int K1,
                                     int accumulate, float* ws, size_t ws_floats, tr_stream_t s) {
  const tr_linear_grad L[2] = {{dY0, ldy0, X0, ldx0, dW0, db0, M0, N0, K0}, {dY1, ldy1, X1, ldx1, dW1, db1, M1, N1, K1}};
  return tr_linear_bwd_group(L, 2, accumulate, ws, ws_floats, s);
}

// row ranges of a column sum: 32 rows at least (the head's 128 x 1000 gradient used to run as ONE range on two workgroups: 15 us)
static inline int colsum_splits(int M) {
  int S = (M + 31) / 32;
  return S > 256 ? 256 : S;
}

extern "C" size_t tr_colsum_workspace_floats(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (size_t)colsum_splits(M) * N;
}

extern "C" int tr_colsum_bf16(const uint16_t* dY, long ldy, int yskip, float* db, int accumulate, float* ws, size_t ws_floats, int M,
                              int N, tr_stream_t s) {
  TR_REQUIRE(dY && db && ws, TR_ERR_NULL, "tr_colsum_bf16: null pointer");
  TR_REQUIRE(M > 0 && N > 0 && N % 2 == 0 && ldy % 2 == 0 && ldy >= N && yskip >= 0, TR_ERR_SHAPE, "tr_colsum_bf16: need even N and ldy (M=%d N=%d)", M, N);
  int S = colsum_splits(M);
  if ((size_t)S * N > ws_floats) S = (int)(ws_floats / N);
  TR_REQUIRE(S >= 1, TR_ERR_SHAPE, "tr_colsum_bf16: workspace too small");
  const int rps = (M + S - 1) / S;
  S = (M + rps - 1) / rps;
  hipStream_t st = static_cast<hipStream_t>(s);
  tr_prof_note("colsum_kernel", 0.0, 2.0 * M * N);
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 511) / 512, S), dim3(256), 0, st, dY, ldy, yskip, ws, M, N, rps);
  TR_CHECK_LAUNCH("tr_colsum_bf16");
  reduce_partials(ws, S, (size_t)N, db, accumulate, st);
  TR_CHECK_LAUNCH("tr_colsum_bf16 (reduce)");
  return TR_OK;
}

extern "C" int tr_gelu_bf16(const uint16_t* pre, uint16_t* h, size_t n, tr_stream_t s) {
  TR_REQUIRE(pre && h, TR_ERR_NULL, "tr_gelu_bf16: null pointer");
  TR_REQUIRE(n > 0 && n % 8 == 0, TR_ERR_SHAPE, "tr_gelu_bf16: element count must be a positive multiple of 8");
  TR_REQUIRE(tr_aligned16(pre) && tr_aligned16(h), TR_ERR_ALIGN, "tr_gelu_bf16: pointers must be 16-byte aligned");
  const size_t nch = n / 8;
  tr_prof_note("gelu_fwd_kernel", 0.0, 4.0 * n);
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(s), pre, h, nch);
  TR_CHECK_LAUNCH("tr_gelu_bf16");
  return TR_OK;
}

extern "C" int tr_gelu_bwd_bf16(const uint16_t* pre, uint16_t* dh, size_t n, tr_stream_t s) {
  TR_REQUIRE(pre && dh, TR_ERR_NULL, "tr_gelu_bwd_bf16: null pointer");
  TR_REQUIRE(n > 0 && n % 8 == 0, TR_ERR_SHAPE, "tr_gelu_bwd_bf16: element count must be a positive multiple of 8");
  TR_REQUIRE(tr_aligned16(pre) && tr_aligned16(dh), TR_ERR_ALIGN, "tr_gelu_bwd_bf16: pointers must be 16-byte aligned");
  const size_t nch = n / 8;
  tr_prof_note("gelu_bwd_kernel", 0.0, 6.0 * n);
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(s), pre, dh, nch);
  TR_CHECK_LAUNCH("tr_gelu_bwd_bf16");
  return TR_OK;
}

static inline int ln_bwd_grid(int M) {
  int g = (M + 3) / 4;
  return g > 512 ? 512 : g;
}

extern "C" size_t tr_layernorm_bwd_workspace_floats(int M, int D) { return (size_t)ln_bwd_grid(M) * 2 * D; }

// ---- deferred reduction of the LayerNorm parameter gradients (see partial_reduce_tall_segs_kernel).  One context per host thread: the
// backward executor opens it around its block loop; calls outside a context, calls whose partials do not fit the region, and calls with
// few partials or long rows (which the tall layout does not serve) reduce at once as before.
namespace {
struct LnDefer {
  float* region = nullptr;
  size_t cap = 0, used = 0;
  TallSegs T;
  int n = 0;
  hipStream_t st = nullptr;
};
thread_local LnDefer* g_ln_defer = nullptr;
thread_local LnDefer g_ln_defer_store;
}  // namespace

int tr_ln_defer_flush() {
  LnDefer* c = g_ln_defer;
  if (c == nullptr || c->n == 0) return TR_OK;
  const int blocks = c->T.blk0[c->n];
  for (int i = c->n; i < LN_DEFER_SEGS; ++i) {          // unused entries never match a block
    c->T.part[i] = c->T.part[0]; c->T.dst[i] = c->T.dst[0]; c->T.count[i] = 0; c->T.S[i] = 0; c->T.acc[i] = 0;
    c->T.blk0[i] = 0x7fffffff;
  }
  c->T.blk0[LN_DEFER_SEGS] = blocks;
  hipLaunchKernelGGL(partial_reduce_tall_segs_kernel, dim3(blocks), dim3(256), 0, c->st, c->T);
  c->n = 0;
  c->used = 0;                 // the reduce is ordered before every later launch of this stream: the slices may be rewritten
  c->T.blk0[0] = 0;
  TR_CHECK_LAUNCH("tr_layernorm_bwd (deferred reduce)");
  return TR_OK;
}
void tr_ln_defer_begin(float* region, size_t floats, hipStream_t st) {
  LnDefer* c = &g_ln_defer_store;
  c->region = region; c->cap = floats; c->used = 0; c->n = 0; c->st = st;
  c->T.blk0[0] = 0;
  static const bool off = getenv("TR_LN_DEFER_OFF") != nullptr;       // lab switch: same-box A/B against the per-call reduce
  g_ln_defer = (region != nullptr && floats > 0 && !off) ? c : nullptr;
}
void tr_ln_defer_end() { g_ln_defer = nullptr; }           // drops what was not flushed (error paths)

#define TR_TRY_RC(call)             \
  do {                              \
    int rc__ = (call);              \
    if (rc__ != TR_OK) return rc__; \
  } while (0)
static int layernorm_bwd_impl(const uint16_t* dy, const float* x, long ldx, const float* gamma, const float* g_in, long ldgi,
                              float* g_out, long ldgo, uint16_t* gb_out, const int32_t* idx, int K, int n_in, int n_out,
                              float* g_fused, float* dgamma, float* dbeta, int accumulate, float* ws, size_t ws_floats, int M, int D,
                              float eps, bool scatter_add, tr_stream_t s) {
  TR_REQUIRE(dy && x && gamma && g_out && dgamma && dbeta && ws, TR_ERR_NULL, "tr_layernorm_bwd: null pointer");
  TR_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAX_CHUNKS && ldx % 4 == 0 && ldgo % 4 == 0 && (g_in == nullptr || ldgi % 4 == 0),
             TR_ERR_SHAPE, "tr_layernorm_bwd: need D %% 4 == 0, D <= 1024, strides %% 4 == 0 (M=%d D=%d)", M, D);
  if (idx != nullptr)
    TR_REQUIRE(K >= 1 && n_in >= K + 1 && n_in <= K + 2 && n_out >= K + 1 && M % n_in == 0 && (n_in == K + 1 || g_fused != nullptr) && ldgo == D,
               TR_ERR_SHAPE, "tr_layernorm_bwd: scatter needs n_in in {K+1, K+2}, M %% n_in == 0 (K=%d n_in=%d n_out=%d M=%d)", K, n_in, n_out, M);
  const int grid = ln_bwd_grid(M);
  TR_REQUIRE(ws_floats >= (size_t)grid * 2 * D, TR_ERR_SHAPE, "tr_layernorm_bwd: workspace too small (%zu < %zu floats)", ws_floats, (size_t)grid * 2 * D);
  TR_REQUIRE(tr_aligned16(dy) && tr_aligned16(x) && tr_aligned16(gamma) && tr_aligned16(g_in) && tr_aligned16(g_out) && tr_aligned16(gb_out) &&
                 tr_aligned16(ws) && tr_aligned16(g_fused),
             TR_ERR_ALIGN, "tr_layernorm_bwd: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  LnDefer* dc = g_ln_defer;
  const size_t need = (size_t)grid * 2 * D;
  const bool defer = dc != nullptr && dc->st == st && grid >= 64 && D <= 4096 && need <= dc->cap;
  if (defer) {
    if (dc->n + 2 > LN_DEFER_SEGS || dc->used + need > dc->cap) TR_TRY_RC(tr_ln_defer_flush());     // full: reduce what is there, start over
    ws = dc->region + dc->used;
    dc->used += need;
  }
  tr_prof_note("ln_bwd_kernel", 0.0, (double)M * D * (2.0 + 4.0 + (g_in ? 4.0 : 0.0) + 4.0 + (gb_out ? 2.0 : 0.0)));
  if (scatter_add)
    TR_DISPATCH_NCH(D, hipLaunchKernelGGL((ln_bwd_kernel<NCH, true>), dim3(grid), dim3(256), 0, st, dy, x, ldx, gamma, g_in, ldgi, g_out, ldgo,
                                          static_cast<uint16_t*>(nullptr), idx, K, n_in, n_out, g_fused, ws, M, D, eps));
  else
    TR_DISPATCH_NCH(D, hipLaunchKernelGGL((ln_bwd_kernel<NCH, false>), dim3(grid), dim3(256), 0, st, dy, x, ldx, gamma, g_in, ldgi, g_out, ldgo, gb_out,
                                          idx, K, n_in, n_out, g_fused, ws, M, D, eps));
  TR_CHECK_LAUNCH("tr_layernorm_bwd");
  if (defer) {
    const int nb = (D / 4 + 15) / 16;
    for (int k = 0; k < 2; ++k) {
      const int i = dc->n++;
      dc->T.part[i] = ws + (size_t)k * grid * D;
      dc->T.dst[i] = k ? dbeta : dgamma;
      dc->T.count[i] = D;
      dc->T.S[i] = grid;
      dc->T.acc[i] = accumulate;
      dc->T.blk0[i + 1] = dc->T.blk0[i] + nb;
    }
    return TR_OK;
  }
  reduce_partials2(ws, (size_t)D, dgamma, ws + (size_t)grid * D, (size_t)D, dbeta, grid, accumulate, st);
  TR_CHECK_LAUNCH("tr_layernorm_bwd (reduce)");
  return TR_OK;
}

extern "C" int tr_layernorm_bwd(const uint16_t* dy, const float* x, long ldx, const float* gamma, const float* g_in, long ldgi,
                                float* g_out, long ldgo, uint16_t* gb_out, const int32_t* idx, int K, int n_in, int n_out,
                                float* g_fused, float* dgamma, float* dbeta, int accumulate, float* ws, size_t ws_floats, int M, int D,
                                float eps, tr_stream_t s) {
  return layernorm_bwd_impl(dy, x, ldx, gamma, g_in, ldgi, g_out, ldgo, gb_out, idx, K, n_in, n_out, g_fused, dgamma, dbeta, accumulate, ws,
                            ws_floats, M, D, eps, false, s);
}

// As tr_layernorm_bwd with idx, for gathers whose ids may REPEAT (K-Medoids' medoid rows): rows are ADDED into the zero-filled g_out
// (float atomics); no bf16 copy is written.
extern "C" int tr_layernorm_bwd_scatter_add(const uint16_t* dy, const float* x, const float* gamma, const float* g_in, float* g_out,
                                            const int32_t* idx, int K, int n_out, float* dgamma, float* dbeta, int accumulate, float* ws,
                                            size_t ws_floats, int M, int D, float eps, tr_stream_t s) {
  TR_REQUIRE(idx != nullptr, TR_ERR_NULL, "tr_layernorm_bwd_scatter_add: idx is required");
  return layernorm_bwd_impl(dy, x, D, gamma, g_in, D, g_out, D, nullptr, idx, K, K + 1, n_out, nullptr, dgamma, dbeta, accumulate, ws, ws_floats,
                            M, D, eps, true, s);
}

// dl16: bf16 scratch [B, C]; ws: tr_wgrad_workspace_floats(B, C, D) floats
extern "C" int tr_head_bwd(const float* dlogits, const uint16_t* W, const uint16_t* xn, uint16_t* dxn, float* dW, float* db,
                           int accumulate, uint16_t* dl16, float* ws, size_t ws_floats, int B, int C, int D, tr_stream_t s) {
  TR_REQUIRE(dlogits && W && xn && dxn && dW && db && dl16 && ws, TR_ERR_NULL, "tr_head_bwd: null pointer");
  TR_REQUIRE(B > 0 && C > 0 && D > 0 && C % 8 == 0 && D % 8 == 0, TR_ERR_SHAPE, "tr_head_bwd: need classes, D multiples of 8 (B=%d C=%d D=%d)", B, C, D);
  hipStream_t st = static_cast<hipStream_t>(s);
  hipLaunchKernelGGL(head_dx_kernel, dim3((B + 3) / 4, (D + 63) / 64), dim3(256), 0, st, dlogits, W, dxn, B, C, D);
  TR_CHECK_LAUNCH("tr_head_bwd");
  int rc = tr_f32_to_bf16(dlogits, dl16, (size_t)B * C, s);
  if (rc != TR_OK) return rc;
  rc = tr_wgrad_bf16(dl16, C, 0, xn, D, dW, accumulate, ws, ws_floats, B, C, D, s);
  if (rc != TR_OK) return rc;
  return tr_colsum_bf16(dl16, C, 0, db, accumulate, ws, ws_floats, B, C, s);
}

extern "C" int tr_embed_bwd(const float* g, float* dpos, float* dcls, int accumulate, int B, int N, int D, tr_stream_t s) {
  TR_REQUIRE(g && dpos && dcls, TR_ERR_NULL, "tr_embed_bwd: null pointer");
  TR_REQUIRE(B > 0 && N > 0 && D > 0 && D % 4 == 0, TR_ERR_SHAPE, "tr_embed_bwd: bad shape B=%d N=%d D=%d", B, N, D);
  TR_REQUIRE(tr_aligned16(g) && tr_aligned16(dpos) && tr_aligned16(dcls), TR_ERR_ALIGN, "tr_embed_bwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((N * (D / 4) + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(s), g, dpos, dcls, B, N, D, accumulate);
  TR_CHECK_LAUNCH("tr_embed_bwd");
  return TR_OK;
}

extern "C" int tr_evit_fuse_bwd(const float* x, const uint16_t* delta, const int32_t* compl_idx, const float* scores, const float* g_fused,
                                float* g_out, uint16_t* gb_out, float* dscore, int B, int N, int K, int D, tr_stream_t s) {
  TR_REQUIRE(x && compl_idx && scores && g_fused && g_out && gb_out && dscore, TR_ERR_NULL, "tr_evit_fuse_bwd: null pointer");
  TR_REQUIRE(B > 0 && N > 1 && K >= 1 && K < N - 1 && D % 4 == 0 && D <= 256 * LN_MAX_CHUNKS, TR_ERR_SHAPE,
             "tr_evit_fuse_bwd: bad shape B=%d N=%d K=%d D=%d", B, N, K, D);
  hipStream_t st = static_cast<hipStream_t>(s);
  TR_DISPATCH_NCH(D, hipLaunchKernelGGL((evit_fuse_bwd_kernel<NCH>), dim3(B), dim3(256), 0, st, x, delta, compl_idx, scores, g_fused, g_out, gb_out,
                                        dscore, N, K, D));
  TR_CHECK_LAUNCH("tr_evit_fuse_bwd");
  return TR_OK;
}

extern "C" int tr_tome_merge_bwd(const float* g_merged, const float* size_in, const float* size_out, const int32_t* unm_idx,
                                 const int32_t* src_idx, const int32_t* dst_idx, int32_t* inv_map, float* g_out, uint16_t* gb_out, int B,
                                 int N, int r, int D, tr_stream_t s) {
  TR_REQUIRE(g_merged && size_out && unm_idx && src_idx && dst_idx && inv_map && g_out && gb_out, TR_ERR_NULL, "tr_tome_merge_bwd: null pointer");
  TR_REQUIRE(B > 0 && N >= 3 && r >= 1 && r <= (N - 1) / 2 && D % 4 == 0 && D <= 256 * LN_MAX_CHUNKS, TR_ERR_SHAPE,
             "tr_tome_merge_bwd: bad shape B=%d N=%d r=%d D=%d", B, N, r, D);
  hipStream_t st = static_cast<hipStream_t>(s);
  hipLaunchKernelGGL(tome_invmap_kernel, dim3(B), dim3(256), 0, st, unm_idx, src_idx, dst_idx, inv_map, N, r);
  const int M = B * N;
  TR_DISPATCH_NCH(D, hipLaunchKernelGGL((tome_merge_bwd_kernel<NCH>), dim3((M + 3) / 4), dim3(256), 0, st, g_merged, size_in, size_out, inv_map, g_out,
                                        gb_out, N, N - r, D, M));
  TR_CHECK_LAUNCH("tr_tome_merge_bwd");
  return TR_OK;
}

extern "C" int tr_f32_to_bf16(const float* src, uint16_t* dst, size_t n, tr_stream_t s) {
  TR_REQUIRE(src && dst, TR_ERR_NULL, "tr_f32_to_bf16: null pointer");
  TR_REQUIRE(n > 0 && n % 4 == 0, TR_ERR_SHAPE, "tr_f32_to_bf16: element count must be a positive multiple of 4");
  const size_t nch = n / 4;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(s), src, dst, nch);
  TR_CHECK_LAUNCH("tr_f32_to_bf16");
  return TR_OK;
}

// DPC-KNN CTM backward.  g_in fp32 [B,K+1,D] (gradient wrt the merged stream, CLS row first), x0 fp32 [B,N,D] (stream before the
// merge), x1 fp32 [B,K+1,D] (merged stream), wtok fp32 [B,N-1] (token weights of the forward; ignored when score_w == NULL =
// equal_weight), assign int32 [B,N-1].  g_out fp32 / gb_out bf16 [B,N,D].  d_sw [D], d_sb [1] (+)=.  ws: B*(D+4) floats.
extern "C" int tr_cluster_merge_bwd(const float* g_in, const float* x0, const float* x1, const float* wtok, const int32_t* assign,
                                    const float* score_w, float* g_out, uint16_t* gb_out, float* d_sw, float* d_sb, int accumulate,
                                    float* ws, size_t ws_floats, int B, int N, int K, int D, tr_stream_t s) {
  TR_REQUIRE(g_in && x0 && x1 && assign && g_out && gb_out && ws, TR_ERR_NULL, "tr_cluster_merge_bwd: null pointer");
  TR_REQUIRE(score_w == nullptr || (wtok && d_sw && d_sb), TR_ERR_NULL, "tr_cluster_merge_bwd: weighted merge needs wtok, d_sw, d_sb");
  TR_REQUIRE(B > 0 && N > 1 && K >= 1 && K <= 640 && K <= N - 1 && D % 4 == 0 && D <= 256 * LN_MAX_CHUNKS, TR_ERR_SHAPE,
             "tr_cluster_merge_bwd: bad shape B=%d N=%d K=%d D=%d (K <= 640)", B, N, K, D);
  TR_REQUIRE(ws_floats >= (size_t)B * (D + 4), TR_ERR_SHAPE, "tr_cluster_merge_bwd: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(s);
  // workgroups per image: as many as the workspace holds partials for (8 with (8 B + 1)(D + 4) floats), while B x split stays near the chip
  int split = 8;
  while (split > 1 && (ws_floats < ((size_t)B * split + 1) * (D + 4) || B * split > 2048)) split >>= 1;
  TR_DISPATCH_NCH(D, hipLaunchKernelGGL((cluster_merge_bwd_kernel<NCH>), dim3(B, split), dim3(256), 0, st, g_in, x0, x1, wtok, assign, score_w, g_out,
                                        gb_out, ws, N, K, D));
  TR_CHECK_LAUNCH("tr_cluster_merge_bwd");
  if (score_w != nullptr) {
    // partial rows are [d_sw (D) | d_sb, 0, 0, 0]: reduce them as one vector of D+4 into a scratch tail, then split
    float* tail = ws + (size_t)B * split * (D + 4);
    TR_REQUIRE(ws_floats >= ((size_t)B * split + 1) * (D + 4), TR_ERR_SHAPE, "tr_cluster_merge_bwd: workspace too small");
    reduce_partials(ws, B * split, (size_t)D + 4, tail, 0, st);
    reduce_partials(tail, 1, (size_t)D, d_sw, accumulate, st);
    reduce_partials(tail + D, 1, (size_t)1, d_sb, accumulate, st);
    TR_CHECK_LAUNCH("tr_cluster_merge_bwd (reduce)");
  }
  return TR_OK;
}

extern "C" int tr_ats_scatter(const float* g, const uint16_t* dao_s, const int32_t* ids, float* g_full, uint16_t* dao_full, int B, int N,
                              int Ks, int D, tr_stream_t s) {
  TR_REQUIRE(g && dao_s && ids && g_full && dao_full, TR_ERR_NULL, "tr_ats_scatter: null pointer");
  TR_REQUIRE(B > 0 && N > 1 && Ks >= 1 && Ks <= N && D % 4 == 0 && D <= 256 * LN_MAX_CHUNKS, TR_ERR_SHAPE, "tr_ats_scatter: bad shape");
  const int M = B * Ks;
  hipStream_t st = static_cast<hipStream_t>(s);
  TR_DISPATCH_NCH(D, hipLaunchKernelGGL((ats_scatter_kernel<NCH>), dim3((M + 3) / 4), dim3(256), 0, st, g, dao_s, ids, g_full, dao_full, N, Ks, D, M));
  TR_CHECK_LAUNCH("tr_ats_scatter");
  return TR_OK;
}

// dst[e] = (accumulate ? dst[e] : 0) + sum_{s < S} part[s][e]: the deterministic second stage of the two-stage reductions
extern "C" int tr_reduce_partials_f32(const float* part, int S, size_t count, float* dst, int accumulate, tr_stream_t s) {
  TR_REQUIRE(part && dst && S >= 1 && count >= 1, TR_ERR_NULL, "tr_reduce_partials_f32: bad arguments");
  reduce_partials(part, S, count, dst, accumulate, static_cast<hipStream_t>(s));
  TR_CHECK_LAUNCH("tr_reduce_partials_f32");
  return TR_OK;
}

// ---- fused weight repack: fp32 master parameters -> the bf16 operand copies of the GEMMs (row-major, and transposed for the data-gradient
// GEMMs), ALL matrices of a model in ONE launch.  An optimizer step changes every parameter, so every training step re-makes ~100 copies;
// as one torch kernel per copy that was ~100 launches of 2-5 us plus their boundaries (0.3-0.4 ms of a 9.6 ms step).  Item i: src fp32
// [rows, cols] contiguous; dst bf16 [rows, cols] (nullable); dst_t bf16 [cols, rows] (nullable).  One workgroup per 64 x 64 tile; `first`
// [n + 1] = prefix sums of the items' tile counts.  The destinations are persistent buffers: their addresses stay valid across repacks
// (captured graphs and workspaces survive an optimizer step).
__global__ __launch_bounds__(256) void cast_pack_kernel(const tr_cast_item* __restrict__ items, const int* __restrict__ first, int n_items) {
  __shared__ unsigned short tile[64][66];
  int it = 0;
  {     // the item this tile belongs to: binary search over the prefix sums (n_items <= a few hundred)
    int lo = 0, hi = n_items;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
    it = lo;
  }
  const tr_cast_item I = items[it];
  const int tiles_c = (I.cols + 63) >> 6;
  const int t = blockIdx.x - first[it], r0 = (t / tiles_c) << 6, c0 = (t % tiles_c) << 6;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;          // 16 x 16 threads, 4 columns each, 4 row passes
  const float* src = static_cast<const float*>(I.src);
  uint16_t* dst = static_cast<uint16_t*>(I.dst);
  uint16_t* dst_t = static_cast<uint16_t*>(I.dst_t);
  // 16-byte loads / 8-byte stores only where every row starts on one: whole vectors per row and aligned bases (item-uniform)
  const bool vec = (I.cols & 3) == 0 && (reinterpret_cast<uintptr_t>(I.src) & 15u) == 0 && (reinterpret_cast<uintptr_t>(I.dst) & 7u) == 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = r0 + ty + 16 * p, c = c0 + 4 * tx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec && r < I.rows && c + 3 < I.cols) v = *reinterpret_cast<const float4*>(src + (size_t)r * I.cols + c);
    else if (r < I.rows) {
      if (c < I.cols) v.x = src[(size_t)r * I.cols + c];
      if (c + 1 < I.cols) v.y = src[(size_t)r * I.cols + c + 1];
      if (c + 2 < I.cols) v.z = src[(size_t)r * I.cols + c + 2];
      if (c + 3 < I.cols) v.w = src[(size_t)r * I.cols + c + 3];
    }
    const unsigned lo = pack_bf16x2(v.x, v.y), hi = pack_bf16x2(v.z, v.w);
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
      tile[ty + 16 * p][4 * tx] = (unsigned short)(lo & 0xffffu);
      tile[ty + 16 * p][4 * tx + 1] = (unsigned short)(lo >> 16);
      tile[ty + 16 * p][4 * tx + 2] = (unsigned short)(hi & 0xffffu);
      tile[ty + 16 * p][4 * tx + 3] = (unsigned short)(hi >> 16);
    }
  }
  if (dst_t == nullptr) return;
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = c0 + ty + 16 * p, r = r0 + 4 * tx;               // transposed: row c of dst_t, columns r .. r + 3
    if (c >= I.cols) continue;
    const unsigned short a = tile[4 * tx][ty + 16 * p], b = tile[4 * tx + 1][ty + 16 * p], cc = tile[4 * tx + 2][ty + 16 * p],
                         d = tile[4 * tx + 3][ty + 16 * p];
    uint16_t* o = dst_t + (size_t)c * I.rows + r;
    if (r + 3 < I.rows && (I.rows & 3) == 0 && (reinterpret_cast<uintptr_t>(dst_t) & 7u) == 0) *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)a | ((unsigned)b << 16), (unsigned)cc | ((unsigned)d << 16));
    else {
      if (r < I.rows) o[0] = a;
      if (r + 1 < I.rows) o[1] = b;
      if (r + 2 < I.rows) o[2] = cc;
      if (r + 3 < I.rows) o[3] = d;
    }
  }
}

// nn.Dropout with the caller's keep mask (1 byte per element, non-zero = keep): dst = keep ? src * mul : 0, mul = 1 / (1 - p).  Eight
// elements per thread: 16 bytes of bf16 (or 2 x 16 of fp32) and 8 mask bytes.  The product is rounded to bf16 once.
__global__ __launch_bounds__(256) void dropout_bf16_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, const uint8_t* __restrict__ keep,
                                                           float mul, size_t n8) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const uint4 v = reinterpret_cast<const uint4*>(src)[i];
  const uint2 k = reinterpret_cast<const uint2*>(keep)[i];
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  const unsigned long long kb = ((unsigned long long)k.y << 32) | k.x;
  unsigned o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float lo = ((kb >> (16 * j)) & 0xffull) ? __builtin_bit_cast(float, w[j] << 16) * mul : 0.f;
    const float hi = ((kb >> (16 * j + 8)) & 0xffull) ? __builtin_bit_cast(float, w[j] & 0xffff0000u) * mul : 0.f;
    o[j] = pack_bf16x2(lo, hi);
  }
  reinterpret_cast<uint4*>(dst)[i] = make_uint4(o[0], o[1], o[2], o[3]);
}
__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, const uint8_t* __restrict__ keep, float mul,
                                                          size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(src)[i];
  const unsigned k = reinterpret_cast<const unsigned*>(keep)[i];
  reinterpret_cast<float4*>(dst)[i] = make_float4((k & 0xffu) ? v.x * mul : 0.f, (k & 0xff00u) ? v.y * mul : 0.f, (k & 0xff0000u) ? v.z * mul : 0.f,
                                                  (k & 0xff000000u) ? v.w * mul : 0.f);
}
// dst[b, r, :] = src[b, r, :] * scale[b]: bf16 [B, rows, D] (dst may be src).  DropPath's per-image scaling (see the kernel).
extern "C" int tr_rowscale_bf16(const uint16_t* src, uint16_t* dst, const float* scale, int B, int rows, int D, tr_stream_t s) {
  TR_REQUIRE(src && dst && scale, TR_ERR_NULL, "tr_rowscale_bf16: null pointer");
  TR_REQUIRE(B > 0 && rows > 0 && D > 0 && D % 8 == 0, TR_ERR_SHAPE, "tr_rowscale_bf16: bad shape B=%d rows=%d D=%d", B, rows, D);
  const size_t cpi = (size_t)rows * D / 8, n = cpi * B;
  hipLaunchKernelGGL(rowscale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(s), src, dst, scale, n, cpi);
  TR_CHECK_LAUNCH("tr_rowscale_bf16");
  return TR_OK;
}

// items / first: DEVICE arrays (n_items entries / n_items + 1 prefix sums of ceil(rows/64) * ceil(cols/64)); total_tiles = first[n_items].
extern "C" int tr_cast_pack_bf16(const tr_cast_item* items, const int* first, int n_items, int total_tiles, tr_stream_t s) {
  TR_REQUIRE(items && first, TR_ERR_NULL, "tr_cast_pack_bf16: null pointer");
  TR_REQUIRE(n_items > 0 && total_tiles > 0, TR_ERR_SHAPE, "tr_cast_pack_bf16: nothing to pack (n_items=%d tiles=%d)", n_items, total_tiles);
  hipLaunchKernelGGL(cast_pack_kernel, dim3((unsigned)total_tiles), dim3(256), 0, static_cast<hipStream_t>(s), items, first, n_items);
  TR_CHECK_LAUNCH("tr_cast_pack_bf16");
  return TR_OK;
}

// Dropout of the training path (timm's drop_rate: pos_drop, proj_drop, the Mlp's two nn.Dropout -- topk.py:186, :53) with the CALLER's
// keep mask, forward and backward alike: dst = keep ? src * mul : 0 (dst may be src).  n elements, n % 8 == 0 (bf16) / n % 4 == 0 (fp32).
extern "C" int tr_dropout_bf16(const uint16_t* src, uint16_t* dst, const uint8_t* keep, float mul, size_t n, tr_stream_t s) {
  TR_REQUIRE(src && dst && keep, TR_ERR_NULL, "tr_dropout_bf16: null pointer");
  TR_REQUIRE(n > 0 && n % 8 == 0, TR_ERR_SHAPE, "tr_dropout_bf16: element count must be a positive multiple of 8 (got %zu)", n);
  TR_REQUIRE(tr_aligned16(src) && tr_aligned16(dst) && (reinterpret_cast<uintptr_t>(keep) & 7u) == 0, TR_ERR_ALIGN,
             "tr_dropout_bf16: data must be 16-byte, the mask 8-byte aligned");
  hipLaunchKernelGGL(dropout_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(s), src, dst, keep, mul, n / 8);
  TR_CHECK_LAUNCH("tr_dropout_bf16");
  return TR_OK;
}
extern "C" int tr_dropout_f32(const float* src, float* dst, const uint8_t* keep, float mul, size_t n, tr_stream_t s) {
  TR_REQUIRE(src && dst && keep, TR_ERR_NULL, "tr_dropout_f32: null pointer");
  TR_REQUIRE(n > 0 && n % 4 == 0, TR_ERR_SHAPE, "tr_dropout_f32: element count must be a positive multiple of 4 (got %zu)", n);
  TR_REQUIRE(tr_aligned16(src) && tr_aligned16(dst) && (reinterpret_cast<uintptr_t>(keep) & 3u) == 0, TR_ERR_ALIGN,
             "tr_dropout_f32: data must be 16-byte, the mask 4-byte aligned");
  hipLaunchKernelGGL(dropout_f32_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(s), src, dst, keep, mul, n / 4);
  TR_CHECK_LAUNCH("tr_dropout_f32");
  return TR_OK;
}
