// DPC-KNN token clustering (models/dpcknn.py) on gfx950.  Everything here is fp32 on the fp32 residual stream, in both
// executor precisions: the outputs are INTEGER decisions (cluster centres, assignments) taken on distance differences.
//
//   tr_dpcknn_cluster          cluster_dpc_knn dpcknn.py:44-100: pairwise distances (matmul form, like torch.cdist's mm path),
//                              k-nearest-neighbour density (+ caller-supplied noise, dpcknn.py:71-72), distance to the nearest
//                              denser token, score = distance * density, top-K centres (tr_cls_topk), nearest-centre assignment.
//   tr_cluster_merge_layernorm merge_tokens dpcknn.py:103-132 + CTM's token weight exp(Linear(D,1)) dpcknn.py:155-157, fused
//                              with the next block's norm1: one wave per output cluster, members added in token order (= the
//                              order torch's CPU index_add_ applies them).
// HBM: the [B,P,P] distance matrix is written once and read three times (density, parent distance, assignment) --
// 4*P*P*4 B per image (0.6 MB at P=196), against 2*P*P*D flops of the Gram product.
#include "tr_common.h"
#include "tr_rowops.h"

extern "C" int tr_cls_topk(const float* cls_rows, int32_t* idx, int32_t* compl_idx, float* scores, int B, int H, int N, int K,
                           tr_stream_t s);

namespace {

constexpr int CT = 64, CK = 16;   // 64x64 distance tile, 16-deep slabs of D

// one wave per token: squared norm
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ x, float* __restrict__ nrm, int B, int N, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int P = N - 1;
  if (row >= B * P) return;
  const int b = row / P, p = row - b * P;
  const float* xr = x + ((size_t)b * N + 1 + p) * D;
  float acc = 0.f;
  if ((D & 3) == 0 && D <= 1024) {                                       // 16-byte loads, the whole row in one batch
    const int nch = D >> 2;
    float4 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const float4*>(xr + 4 * min(lane + 64 * c, nch - 1));
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (lane + 64 * c < nch) { acc = fmaf(v[c].x, v[c].x, acc); acc = fmaf(v[c].y, v[c].y, acc); acc = fmaf(v[c].z, v[c].z, acc); acc = fmaf(v[c].w, v[c].w, acc); }
  } else {
    for (int d = lane; d < D; d += 64) acc = fmaf(xr[d], xr[d], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) nrm[row] = acc;
}

// The same norms in the summation order of the one-launch clustering kernel (tr_cluster_fused.hip), which accumulates them while it
// streams the 32-float slabs of the Gram product: two threads per row, thread h chains fma over floats 16h .. 16h+15 of every slab, slab
// after slab; |x|^2 = chain 0 + chain 1.  Used whenever D % 32 == 0, so that the staged launches and the fused kernel see the same
// norms bit for bit.
__global__ __launch_bounds__(256) void sqnorm_pair_kernel(const float* __restrict__ x, float* __restrict__ nrm, int B, int N, int D) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int P = N - 1;
  const int row = min(t >> 1, B * P - 1), h = t & 1;
  const int b = row / P, p = row - b * P;
  const float* xr = x + ((size_t)b * N + 1 + p) * D + 16 * h;
  float a = 0.f;
  for (int k0 = 0; k0 < D; k0 += 32) {
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(xr + k0 + 4 * q);
#pragma unroll
    for (int q = 0; q < 4; ++q) { a = fmaf(v[q].x, v[q].x, a); a = fmaf(v[q].y, v[q].y, a); a = fmaf(v[q].z, v[q].z, a); a = fmaf(v[q].w, v[q].w, a); }
  }
  const float other = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));   // lane ^ 1
  if (h == 0 && (t >> 1) < B * P) nrm[row] = a + other;
}

// dist[b][i][j] = sqrt(max(|x_i|^2 + |x_j|^2 - 2 x_i.x_j, 1e-30)) / sqrt(D);  grid (tiles, B).
// DIRECT: sqrt(sum (x_i - x_j)^2) / sqrt(D) -- torch.cdist only takes the matmul form when P > 25
// (use_mm_for_euclid_dist_if_necessary); small late stages (P <= 25) get exact zeros on the diagonal like the reference.
template <bool DIRECT>
__global__ __launch_bounds__(256) void dist_kernel(const float* __restrict__ x, const float* __restrict__ nrm,
                                                   float* __restrict__ dist, int N, int D, float sqrt_d) {
  __shared__ float sA[CK][CT + 4];
  __shared__ float sB[CK][CT + 4];
  const int P = N - 1;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int nt = (P + CT - 1) / CT;
  const int i0 = (blockIdx.x / nt) * CT, j0 = (blockIdx.x % nt) * CT;
  const int b = blockIdx.y;
  const float* xb = x + ((size_t)b * N + 1) * D;
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  const float* ap = xb + (size_t)min(i0 + lr, P - 1) * D + lk;
  const float* bp = xb + (size_t)min(j0 + lr, P - 1) * D + lk;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < D; k0 += CK) {
    const float4 a = *reinterpret_cast<const float4*>(ap + k0);
    const float4 w = *reinterpret_cast<const float4*>(bp + k0);
    __syncthreads();
    sA[lk + 0][lr] = a.x; sA[lk + 1][lr] = a.y; sA[lk + 2][lr] = a.z; sA[lk + 3][lr] = a.w;
    sB[lk + 0][lr] = w.x; sB[lk + 1][lr] = w.y; sB[lk + 2][lr] = w.z; sB[lk + 3][lr] = w.w;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CK; ++k) {
      const float4 av = *reinterpret_cast<const float4*>(&sA[k][ty * 4]);
      const float4 wv = *reinterpret_cast<const float4*>(&sB[k][tx * 4]);
      const float a4[4] = {av.x, av.y, av.z, av.w}, w4[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (DIRECT) {
            const float df = a4[i] - w4[j];
            acc[i][j] = fmaf(df, df, acc[i][j]);
          } else {
            acc[i][j] = fmaf(a4[i], w4[j], acc[i][j]);
          }
        }
    }
  }
  const float* nb = nrm + (size_t)b * P;
  float* db = dist + (size_t)b * P * P;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gi = i0 + ty * 4 + i;
    if (gi >= P) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gj = j0 + tx * 4 + j;
      if (gj >= P) continue;
      const float d2 = DIRECT ? acc[i][j] : fmaxf((nb[gi] + nb[gj]) - 2.0f * acc[i][j], 1e-30f);
      db[(size_t)gi * P + gj] = sqrtf(d2) / sqrt_d;
    }
  }
}

// ---- fast-path distances: Gram product on MFMA with every fp32 operand split into bf16 hi + bf16 lo
// (x.y ~= hi.hi' + hi.lo' + lo.hi': relative error ~2^-16, against 2^-9 for a plain bf16 product).  Used by the bf16
// executor, whose token features already carry bf16-level noise from the GEMMs upstream; the fp32 validation executor keeps
// the exact VALU kernel above.  128x128 tile per workgroup, 64x64 per wave, v_mfma_f32_16x16x32_bf16, D walked in 32-wide
// slabs with the next slab's global loads in flight under the current slab's MFMAs.
constexpr int GT = 256, GK = 32, GLD = 40;     // tile, slab depth, LDS row stride in bf16 (80 B: conflict-free 16-B fragment reads)

__device__ __forceinline__ void split_store(const float4 v, unsigned short* hi, unsigned short* lo) {
  const float f[4] = {v.x, v.y, v.z, v.w};
  unsigned int h2[2], l2[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const unsigned int hp = pack_bf16x2(f[2 * q], f[2 * q + 1]);
    const float r0 = f[2 * q] - __uint_as_float(hp << 16), r1 = f[2 * q + 1] - __uint_as_float(hp & 0xffff0000u);
    h2[q] = hp;
    l2[q] = pack_bf16x2(r0, r1);
  }
  *reinterpret_cast<uint2*>(hi) = make_uint2(h2[0], h2[1]);
  *reinterpret_cast<uint2*>(lo) = make_uint2(l2[0], l2[1]);
}

// 512 threads = 4x2 waves of 64x64: a 256 x 128 tile per workgroup (two workgroups cover an image at 224^2 inputs, so a token
// row is read ~1.5 times per launch instead of 4 times with 128 x 128 tiles -- the kernel is bound by those fp32 reads).
constexpr int GTN = 128;
__global__ __launch_bounds__(512) void dist_mfma_kernel(const float* __restrict__ x, const float* __restrict__ nrm,
                                                        float* __restrict__ dist, int N, int D, float sqrt_d) {
  __shared__ __attribute__((aligned(16))) unsigned short sAh[GT * GLD], sAl[GT * GLD], sBh[GTN * GLD], sBl[GTN * GLD];
  const int P = N - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  const int ntj = (P + GTN - 1) / GTN, nt = ((P + GT - 1) / GT) * ntj;
  // 1-D grid, XCD-aware: the tiles of one image read the same token rows (15 tiles at P = 576 read them 10 x), so an image's tiles go to
  // ONE XCD and its L2 serves the re-reads (r04a at 384^2: 299 MB per launch against 80 MB algorithmic with the tiles dealt round-robin)
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = lid / nt, tile = lid - b * nt;
  const int i0 = (tile / ntj) * GT, j0 = (tile % ntj) * GTN;
  // The matrix is symmetric: a tile that lies wholly BELOW the diagonal (its last column before its first row) is not computed -- the
  // tiles above the diagonal also store their elements transposed wherever the mirror image falls into such a tile (round 4: 6 of the 15
  // tiles at P = 576).  Those pairs are then exactly symmetric; pairs inside the tiles that straddle the diagonal are still computed twice.
  if (j0 + GTN <= i0) return;
  const float* xb = x + ((size_t)b * N + 1) * D;
  const int lr = tid >> 1, lc = (tid & 1) * 16;                      // A staging: row lr (0..255), 16 consecutive floats
  const int br = tid >> 2, bc = (tid & 3) * 8;                       // B staging: row br (0..127), 8 consecutive floats
  const float* ap = xb + (size_t)min(i0 + lr, P - 1) * D + lc;
  const float* bp = xb + (size_t)min(j0 + br, P - 1) * D + bc;
  float4 ra[4], rb[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const float4*>(ap + 4 * q);
#pragma unroll
  for (int q = 0; q < 2; ++q) rb[q] = *reinterpret_cast<const float4*>(bp + 4 * q);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool active = i0 + wm * 64 < P && j0 + wn * 64 < P;          // else this wave's 64x64 block lies outside the matrix
  for (int k0 = 0; k0 < D; k0 += GK) {
    lds_barrier();                                                   // previous slab's fragment reads are done (LDS only: the prefetch stays in flight)
#pragma unroll
    for (int q = 0; q < 4; ++q) split_store(ra[q], sAh + lr * GLD + lc + 4 * q, sAl + lr * GLD + lc + 4 * q);
#pragma unroll
    for (int q = 0; q < 2; ++q) split_store(rb[q], sBh + br * GLD + bc + 4 * q, sBl + br * GLD + bc + 4 * q);
    if (k0 + GK < D) {
#pragma unroll
      for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const float4*>(ap + k0 + GK + 4 * q);
#pragma unroll
      for (int q = 0; q < 2; ++q) rb[q] = *reinterpret_cast<const float4*>(bp + k0 + GK + 4 * q);
    }
    lds_barrier();
    if (!active) continue;
    bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[t] = *reinterpret_cast<const bf16x8*>(sAh + (wm * 64 + t * 16 + frow) * GLD + fq * 8);
      al[t] = *reinterpret_cast<const bf16x8*>(sAl + (wm * 64 + t * 16 + frow) * GLD + fq * 8);
      bh[t] = *reinterpret_cast<const bf16x8*>(sBh + (wn * 64 + t * 16 + frow) * GLD + fq * 8);
      bl[t] = *reinterpret_cast<const bf16x8*>(sBl + (wn * 64 + t * 16 + frow) * GLD + fq * 8);
    }
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        // small terms first, then the dominant one
        acc[bb][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[bb], ah[a], acc[bb][a], 0, 0, 0);
        acc[bb][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[bb], al[a], acc[bb][a], 0, 0, 0);
        acc[bb][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[bb], ah[a], acc[bb][a], 0, 0, 0);
      }
  }
  // acc[bb][a][e] = x_i . x_j with i = i0 + wm*64 + a*16 + frow, j = j0 + wn*64 + bb*16 + 4*fq + e
  const float* nb = nrm + (size_t)b * P;
  float* db = dist + (size_t)b * P * P;
  // the 4 row norms and 16 column norms of this lane are fetched up front (clamped, branch-free) instead of one dependent load
  // inside every output element
  float ni4[4], nj[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) ni4[a] = nb[min(i0 + wm * 64 + a * 16 + frow, P - 1)];
#pragma unroll
  for (int bb = 0; bb < 4; ++bb)
#pragma unroll
    for (int e = 0; e < 4; ++e) nj[bb][e] = nb[min(j0 + wn * 64 + bb * 16 + 4 * fq + e, P - 1)];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int gi = i0 + wm * 64 + a * 16 + frow;
    if (gi >= P) continue;
    const float ni = ni4[a];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const int gj0 = j0 + wn * 64 + bb * 16 + 4 * fq;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = sqrtf(fmaxf((ni + nj[bb][e]) - 2.0f * acc[bb][a][e], 1e-30f)) / sqrt_d;
      // a lane's four values are four consecutive columns of one row: ONE 16-byte store when the row pitch allows (P % 4 == 0), i.e. 16
      // rows x 64 contiguous bytes per instruction instead of 64 scattered 4-byte writes (round 4)
      if ((P & 3) == 0 && gj0 + 3 < P) {
        *reinterpret_cast<float4*>(db + (size_t)gi * P + gj0) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (gj0 + e < P) db[(size_t)gi * P + gj0 + e] = o[e];
      }
      // mirror image (row gj, column gi): owned by the tile (gj / GT, gi / GTN); written here when that tile was skipped.  gj0 .. gj0 + 3
      // share their row tile (gj0 is a multiple of 4, GT of 256); 16 lanes with consecutive gi write 64 contiguous bytes of a row.
      if ((gi / GTN) * GTN + GTN <= (gj0 / GT) * GT) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (gj0 + e < P) db[(size_t)(gj0 + e) * P + gi] = o[e];
      }
    }
  }
}

constexpr int KNN_MAX = 8;
constexpr int MAX_PER_LANE = 16;  // P <= 1024

// (value, index) minimum over the wave, ties -> smallest index; result in every lane.  Full-rate VALU: four DPP steps inside the 16-lane
// rows, then v_permlane16_swap / v_permlane32_swap across them (round 4: the ds_bpermute butterfly this replaces -- 12 LDS-pipe round
// trips per reduction, 48 per step of kmed_iterate_kernel -- was most of that kernel's 214 us at 576 tokens).
__device__ __forceinline__ void wave_min_pair(float& v, int& i) {
#define TR_MINPAIR_STEP(OV, OI)                                   \
  do {                                                            \
    const float ov__ = (OV);                                      \
    const int oi__ = (OI);                                        \
    if (ov__ < v || (ov__ == v && oi__ < i)) { v = ov__; i = oi__; } \
  } while (0)
  TR_MINPAIR_STEP(__builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)), __builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true));
  TR_MINPAIR_STEP(__builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)), __builtin_amdgcn_mov_dpp(i, 0x4E, 0xF, 0xF, true));
  TR_MINPAIR_STEP(__builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)), __builtin_amdgcn_mov_dpp(i, 0x141, 0xF, 0xF, true));
  TR_MINPAIR_STEP(__builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)), __builtin_amdgcn_mov_dpp(i, 0x140, 0xF, 0xF, true));
  {
    const auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap((unsigned)i, (unsigned)i, false, false);
    const unsigned a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
    v = __builtin_bit_cast(float, a0); i = (int)b0;                // one of the two is this lane's own pair, the other its partner row's
    TR_MINPAIR_STEP(__builtin_bit_cast(float, a1), (int)b1);
  }
  {
    const auto a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap((unsigned)i, (unsigned)i, false, false);
    const unsigned a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
    v = __builtin_bit_cast(float, a0); i = (int)b0;
    TR_MINPAIR_STEP(__builtin_bit_cast(float, a1), (int)b1);
  }
#undef TR_MINPAIR_STEP
}

// one wave per row: density = exp(-mean(k smallest d^2)) + noise*1e-6; per-image max distance via atomicMax on the bits.
// PER_LANE = row elements held per lane (4: P <= 256, the 224^2 case; 16: P <= 1024)
template <int PER_LANE>
__global__ __launch_bounds__(256) void density_kernel(const float* __restrict__ dist, const float* __restrict__ noise,
                                                      float* __restrict__ density, float* __restrict__ rowmax, int B, int P,
                                                      int k) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * P) return;
  const float* dr = dist + (size_t)row * P;
  float v[PER_LANE];
  float mx = 0.f;
#pragma unroll
  for (int c = 0; c < PER_LANE; ++c) v[c] = dr[min(c * 64 + lane, P - 1)];     // branch-free: the row's loads go out in one batch
#pragma unroll
  for (int c = 0; c < PER_LANE; ++c) {
    const int j = c * 64 + lane;
    if (j < P) mx = fmaxf(mx, v[c]);
    else v[c] = INFINITY;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) rowmax[row] = mx;                                       // reduced per image by parent_score_kernel (no atomics)
  float ss = 0.f;
  for (int t = 0; t < k; ++t) {                                          // ascending extraction = torch.topk(largest=False) order
    float best = INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < PER_LANE; ++c)
      if (v[c] < best) { best = v[c]; bi = c * 64 + lane; }
    wave_min_pair(best, bi);
    ss += best * best;
#pragma unroll
    for (int c = 0; c < PER_LANE; ++c)
      if (c * 64 + lane == bi) v[c] = INFINITY;
  }
  if (lane == 0) density[row] = expf(-(ss / (float)k)) + (noise ? noise[row] * 1e-6f : 0.f);
}

// one wave per row: distance to the nearest token of higher density (else the image's max distance); score = that * density,
// written in [B,N] layout (column 0 = CLS slot, unused) for tr_cls_topk
template <int PER_LANE>
__global__ __launch_bounds__(256) void parent_score_kernel(const float* __restrict__ dist, const float* __restrict__ density,
                                                           const float* __restrict__ rowmax, float* __restrict__ score_rows, int B,
                                                           int P) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * P) return;
  const int b = row / P, i = row - b * P;
  const float* dr = dist + (size_t)row * P;
  const float* db = density + (size_t)b * P;
  // all three row reads (row maxima, densities, distances) are issued branch-free before the first reduction
  float rm[PER_LANE], dn[PER_LANE], dd[PER_LANE];
#pragma unroll
  for (int c = 0; c < PER_LANE; ++c) {
    const int j = min(c * 64 + lane, P - 1);
    rm[c] = rowmax[(size_t)b * P + j];
    dn[c] = db[j];
    dd[c] = dr[j];
  }
  const float di = db[i];
  float dmax = 0.f;                                                      // dist_matrix.flatten(1).max(): max of the row maxima
#pragma unroll
  for (int c = 0; c < PER_LANE; ++c) dmax = fmaxf(dmax, rm[c]);             // clamped duplicates do not change a maximum
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64));
  float mn = INFINITY;
#pragma unroll
  for (int c = 0; c < PER_LANE; ++c) mn = fminf(mn, dn[c] > di ? dd[c] : dmax);   // nor a minimum
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 64));
  if (lane == 0) {
    score_rows[(size_t)b * (P + 1) + 1 + i] = mn * di;
    if (i == 0) score_rows[(size_t)b * (P + 1)] = 0.f;
  }
}

// one workgroup per image: nearest centre per token (first on ties), then centres to themselves
__global__ __launch_bounds__(256) void assign_kernel(const float* __restrict__ dist, const int32_t* __restrict__ centers,
                                                     int32_t* __restrict__ idx_cluster, int P, int K) {
  extern __shared__ int s_c[];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < K; k += 256) s_c[k] = centers[(size_t)b * K + k];
  __syncthreads();
  const float* db = dist + (size_t)b * P * P;
  for (int p = tid; p < P; p += 256) {
    float best = INFINITY;
    int arg = 0;
    for (int k0 = 0; k0 < K; k0 += 8) {               // eight centre rows requested per step (independent addresses), compared in order
      float d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d[u] = db[(size_t)s_c[min(k0 + u, K - 1)] * P + p];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u < K && d[u] < best) { best = d[u]; arg = k0 + u; }
    }
    idx_cluster[(size_t)b * P + p] = arg;
  }
  __syncthreads();
  for (int k = tid; k < K; k += 256) idx_cluster[(size_t)b * P + s_c[k]] = k;
}

// one wave per token: w = exp(x . ws + bs)   (CTM.score, dpcknn.py:155-157)
__global__ __launch_bounds__(256) void token_weight_kernel(const float* __restrict__ x, const float* __restrict__ ws,
                                                           const float* __restrict__ bs, float* __restrict__ w, int B, int N,
                                                           int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int P = N - 1;
  if (row >= B * P) return;
  const int b = row / P, p = row - b * P;
  const float* xr = x + ((size_t)b * N + 1 + p) * D;
  float acc = 0.f;
  if ((D & 3) == 0 && D <= 1024) {
    const int nch = D >> 2;
    float4 v[4], u[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ch = min(lane + 64 * c, nch - 1);
      v[c] = *reinterpret_cast<const float4*>(xr + 4 * ch);
      u[c] = *reinterpret_cast<const float4*>(ws + 4 * ch);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (lane + 64 * c < nch) { acc = fmaf(v[c].x, u[c].x, acc); acc = fmaf(v[c].y, u[c].y, acc); acc = fmaf(v[c].z, u[c].z, acc); acc = fmaf(v[c].w, u[c].w, acc); }
  } else {
    for (int d = lane; d < D; d += 64) acc = fmaf(xr[d], ws[d], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) w[row] = expf(acc + bs[0]);
}


// one wave per output row (row 0 = CLS copy, row 1+c = cluster c), fused with the following LayerNorm
template <bool F32, int NCH>
__global__ __launch_bounds__(256) void cluster_merge_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                      const int32_t* __restrict__ idx_cluster,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      float* __restrict__ x_out, void* __restrict__ y, int N, int K,
                                                                      int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int rblocks = (K + 1 + 3) >> 2;
  const int b = blockIdx.x / rblocks;
  const int r = (blockIdx.x % rblocks) * 4 + (threadIdx.x >> 6);
  if (r > K) return;
  const int P = N - 1, nchunks = D >> 2;
  const float* xb = x + (size_t)b * N * D;
  float4 v[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r == 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) v[c] = *reinterpret_cast<const float4*>(xb + 4 * min(lane + 64 * c, nchunks - 1));
  } else {
    const int cl = r - 1;
    const int32_t* ic = idx_cluster + (size_t)b * P;
    const float* wb = w ? w + (size_t)b * P : nullptr;
    // all_weight = sum of member weights in token order + 1e-6 (index_add_ then + 1e-6, dpcknn.py:123-126)
    // lane l holds cluster id and weight of token p0 + l; members come out of a ballot in token order and their weights by
    // v_readlane, so the only dependent memory accesses left are the member rows themselves (fetched branch-free, whole row at once)
    float aw = 0.f;
    for (int p0 = 0; p0 < P; p0 += 64) {
      const int p = p0 + lane;
      const bool in = p < P;
      const int cid = in ? ic[p] : -1;
      const float wl = in ? (wb ? wb[p] : 1.0f) : 0.f;
      unsigned long long mask = __ballot(cid == cl);
      while (mask) {
        const int q = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        aw += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wl), q));
      }
    }
    aw += 1e-6f;
    for (int p0 = 0; p0 < P; p0 += 64) {
      const int p = p0 + lane;
      const bool in = p < P;
      const int cid = in ? ic[p] : -1;
      const float wl = in ? (wb ? wb[p] : 1.0f) : 0.f;
      unsigned long long mask = __ballot(cid == cl);
      while (mask) {
        const int q = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const float nw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wl), q)) / aw;   // norm_weight = token_weight / all_weight[idx]
        const float* xr = xb + (size_t)(1 + p0 + q) * D;
        float4 a[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) a[c] = *reinterpret_cast<const float4*>(xr + 4 * min(lane + 64 * c, nchunks - 1));
#pragma unroll
        for (int c = 0; c < NCH; ++c) { v[c].x += a[c].x * nw; v[c].y += a[c].y * nw; v[c].z += a[c].z * nw; v[c].w += a[c].w * nw; }
      }
    }
  }
  const size_t orow = (size_t)b * (K + 1) + r;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (lane + 64 * c < nchunks) *reinterpret_cast<float4*>(x_out + orow * D + 4 * (lane + 64 * c)) = v[c];
  ln_row_store<F32, NCH>(v, nchunks, lane, D, eps, gamma, beta,
                    F32 ? (void*)(reinterpret_cast<float*>(y) + orow * D) : (void*)(reinterpret_cast<uint16_t*>(y) + orow * D));
}

// ---- K-Medoids (models/kmedoids.py) ---------------------------------------------------------------------------------------
// w[b][n] = sum_h sum_wave colsum_part[b][h][wave][n], fixed order; [B,N] layout (column 0 = CLS, ignored by the top-K)
__global__ __launch_bounds__(256) void kmed_weight_kernel(const float* __restrict__ part, float* __restrict__ w, int B, int N, int H) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * N) return;
  const int b = e / N, n = e - b * N;
  float acc = 0.f;
  for (int h = 0; h < H; ++h)
    for (int wv = 0; wv < 4; ++wv) acc += part[(((size_t)b * H + h) * 4 + wv) * N + n];
  w[e] = acc;
}

// one wave per row: t_i = sum_j dist[i][j] * w_i   (row sums of weighted_dist_matrix, kmedoids.py:65,77)
__global__ __launch_bounds__(256) void kmed_rowcost_kernel(const float* __restrict__ dist, const float* __restrict__ w,
                                                           float* __restrict__ t, int B, int P) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * P) return;
  const int b = row / P, i = row - b * P;
  const float wi = w[(size_t)b * (P + 1) + 1 + i];
  const float* dr = dist + (size_t)row * P;
  float acc = 0.f;
  for (int j = lane; j < P; j += 64) acc += dr[j] * wi;
  acc = wave_sum(acc);
  if (lane == 0) t[row] = acc;
}

__device__ __forceinline__ unsigned long long cost_key(float v, int i) {          // v >= 0: bit order == value order
  return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned int)i;
}

// one workgroup (16 waves) per image: iterate {assignment, medoid update}, final assignment.
// Assignment (round 4): a token's distances to the K medoids are read from the medoids' ROWS (dist[c_k][p], 64 tokens of a row per
// wave-load: full 256-byte segments) instead of gathered from the token's own row (dist[p][c_k]: 64 lanes in 64 different sectors of a
// 2.3-KB row -- 83 k sector accesses per pass at P = 576, K = 144, which was the kernel: 44 us per pass).  The matrix is symmetric up to
// the last bit of the split-operand Gram product (|x_i|^2 + |x_j|^2 is the same sum, the cross terms are accumulated in the other
// order), so an arg-min can differ from the column form only between medoids whose distances to the token agree to ~1e-7 relative: the
// tolerance the medoid comparisons against the oracle already carry.  Work unit = (64 tokens, 16 medoids): 16 row loads in flight, the
// lane's first minimum over ascending k, then ds_min_u64 on (distance bits, k) per token -- ties -> smallest k, torch.argmin's rule.
constexpr int KMT = 1024;
__global__ __launch_bounds__(KMT) void kmed_iterate_kernel(const float* __restrict__ dist, const float* __restrict__ t,
                                                           int32_t* __restrict__ centers, int32_t* __restrict__ assign, int P, int K,
                                                           int iters) {
  extern __shared__ unsigned long long s_best[];      // [K] packed (cost, index); then [P] packed (distance, medoid); then int s_c[K]
  unsigned long long* s_tok = s_best + K;
  int* s_c = reinterpret_cast<int*>(s_tok + P);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* db = dist + (size_t)b * P * P;
  const float* tb = t + (size_t)b * P;
  for (int k = tid; k < K; k += KMT) s_c[k] = centers[(size_t)b * K + k];
  __syncthreads();
  const float masked = (float)P * 1000000.0f;         // a row outside cluster k sums to P * 1e6 (kmedoids.py:76-77)
  const int ng = (P + 63) >> 6, nc = (K + 15) >> 4;
  for (int it = 0; it <= iters; ++it) {
    for (int k = tid; k < K; k += KMT) s_best[k] = cost_key(masked, 0);
    for (int p = tid; p < P; p += KMT) s_tok[p] = ~0ull;
    __syncthreads();
    for (int u = wave; u < ng * nc; u += KMT / 64) {
      const int g = u / nc, k0 = (u - g * nc) * 16;
      const int p = min(g * 64 + lane, P - 1);
      float d[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) d[q] = db[(size_t)s_c[min(k0 + q, K - 1)] * P + p];
      float best = INFINITY;
      int arg = 0x7fffffff;
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (k0 + q < K && d[q] < best) { best = d[q]; arg = k0 + q; }       // ascending k, strict <: the first minimum
      // (all-NaN distances leave arg unset: the token keeps ~0 below and falls to medoid 0, like torch.argmin never returns out of range)
      if (g * 64 + lane < P && arg != 0x7fffffff) atomicMin(&s_tok[p], ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)arg);
    }
    __syncthreads();
    for (int p = tid; p < P; p += KMT) {
      const unsigned long long kb = s_tok[p];
      const int arg = kb == ~0ull ? 0 : (int)(kb & 0xffffffffull);
      if (it == iters) assign[(size_t)b * P + p] = arg;
      else atomicMin(&s_best[arg], cost_key(tb[p], p));   // smallest cost, ties -> smallest index; empty cluster -> index 0
    }
    __syncthreads();
    if (it < iters)
      for (int k = tid; k < K; k += KMT) s_c[k] = (int)(s_best[k] & 0xffffffffull);
    __syncthreads();
  }
  for (int k = tid; k < K; k += KMT) centers[(size_t)b * K + k] = s_c[k];
}

}  // namespace

namespace {
// fused_order: the summation order of the one-launch DPC-KNN kernel (only where that kernel exists: the staged DPC-KNN launches then see its
// norms bit for bit); everything else takes the row-per-wave kernel, which is twice as fast at D = 768 (11 vs 22 us at B = 64, P = 576)
void launch_sqnorm(const float* x, float* nrm, int B, int N, int D, hipStream_t st, bool fused_order) {
  const int rows = B * (N - 1);
  if (fused_order && D % 32 == 0) hipLaunchKernelGGL(sqnorm_pair_kernel, dim3((2 * rows + 255) / 256), dim3(256), 0, st, x, nrm, B, N, D);
  else hipLaunchKernelGGL(sqnorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, nrm, B, N, D);
}
}  // namespace

extern "C" size_t tr_dpcknn_workspace_floats(int B, int N) {
  const size_t P = N > 1 ? N - 1 : 0;
  return (size_t)B * P * P + (size_t)B * (3 * P + N) + 64 + (size_t)B;
}

static void launch_dist(bool fast, const float* x, const float* nrm, float* dist, int B, int N, int D, float sqrt_d, hipStream_t st) {
  const int P = N - 1;
  if (P <= 25) {                       // torch.cdist's exact form for tiny inputs
    const int nt = (P + CT - 1) / CT;
    hipLaunchKernelGGL(dist_kernel<true>, dim3(nt * nt, B), dim3(256), 0, st, x, nrm, dist, N, D, sqrt_d);
  } else if (fast && D % GK == 0) {
    const int nti = (P + GT - 1) / GT, ntj = (P + GTN - 1) / GTN;
    hipLaunchKernelGGL(dist_mfma_kernel, dim3(nti * ntj * B), dim3(512), 0, st, x, nrm, dist, N, D, sqrt_d);
  } else {
    const int nt = (P + CT - 1) / CT;
    hipLaunchKernelGGL(dist_kernel<false>, dim3(nt * nt, B), dim3(256), 0, st, x, nrm, dist, N, D, sqrt_d);
  }
}

extern "C" int tr_dpcknn_fused_supported(int N, int D, int k);
extern "C" int tr_dpcknn_cluster_fused(const float* x, const float* noise, int32_t* centers, int32_t* idx_cluster, float* scores, int B, int N,
                                       int D, int K, int k, tr_stream_t s);

extern "C" int tr_dpcknn_cluster(const float* x, const float* noise, float* ws, int32_t* centers, int32_t* idx_cluster,
                                 float* scores, int B, int N, int D, int K, int k, int fast_dist, tr_stream_t s) {
  TR_REQUIRE(x && ws && centers && idx_cluster && scores, TR_ERR_NULL, "tr_dpcknn_cluster: null pointer");
  const int P = N - 1;
  // fast_dist 1: one launch with the distance matrix in LDS wherever that kernel applies (tr_cluster_fused.hip); 2: the staged launches
  if (fast_dist == 1 && B > 0 && K >= 1 && K <= P && tr_dpcknn_fused_supported(N, D, k) && tr_aligned16(x))
    return tr_dpcknn_cluster_fused(x, noise, centers, idx_cluster, scores, B, N, D, K, k, s);
  TR_REQUIRE(B > 0 && P >= 2 && P <= 64 * MAX_PER_LANE && D > 0 && D % CK == 0, TR_ERR_SHAPE,
             "tr_dpcknn_cluster: need 2 <= P <= %d and D %% %d == 0 (N=%d D=%d)", 64 * MAX_PER_LANE, CK, N, D);
  TR_REQUIRE(K >= 1 && K <= P && k >= 1 && k <= KNN_MAX && k <= P, TR_ERR_SHAPE, "tr_dpcknn_cluster: bad K=%d / k=%d for P=%d", K, k, P);
  TR_REQUIRE(tr_aligned16(x) && tr_aligned16(ws), TR_ERR_ALIGN, "tr_dpcknn_cluster: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  float* dist = ws;
  float* nrm = dist + (size_t)B * P * P;
  float* density = nrm + (size_t)B * P;
  float* score_rows = density + (size_t)B * P;                  // [B,N]
  float* rowmax = nrm;                                            // the norms are dead once the distances exist
  const int rows = B * P, rb = (rows + 3) / 4;
  launch_sqnorm(x, nrm, B, N, D, st, tr_dpcknn_fused_supported(N, D, k) != 0);
  launch_dist(fast_dist != 0, x, nrm, dist, B, N, D, (float)sqrt((double)D), st);
  if (P <= 256) hipLaunchKernelGGL(density_kernel<4>, dim3(rb), dim3(256), 0, st, dist, noise, density, rowmax, B, P, k);
  else hipLaunchKernelGGL(density_kernel<16>, dim3(rb), dim3(256), 0, st, dist, noise, density, rowmax, B, P, k);
  if (P <= 256) hipLaunchKernelGGL(parent_score_kernel<4>, dim3(rb), dim3(256), 0, st, dist, density, rowmax, score_rows, B, P);
  else hipLaunchKernelGGL(parent_score_kernel<16>, dim3(rb), dim3(256), 0, st, dist, density, rowmax, score_rows, B, P);
  TR_CHECK_LAUNCH("tr_dpcknn_cluster");
  int rc = tr_cls_topk(score_rows, centers, nullptr, scores, B, 1, N, K, s);      // topk(score, K), sorted descending
  if (rc != TR_OK) return rc;
  hipLaunchKernelGGL(assign_kernel, dim3(B), dim3(256), sizeof(int) * K, st, dist, centers, idx_cluster, P, K);
  TR_CHECK_LAUNCH("tr_dpcknn_cluster");
  return TR_OK;
}

extern "C" int tr_cluster_merge_layernorm(const float* x, const float* score_w, const float* score_b, float* w_ws,
                                          const int32_t* idx_cluster, const float* gamma, const float* beta, float* x_out, void* y,
                                          int y_is_f32, int B, int N, int K, int D, float eps, tr_stream_t s) {
  TR_REQUIRE(x && idx_cluster && gamma && beta && x_out && y, TR_ERR_NULL, "tr_cluster_merge_layernorm: null pointer");
  TR_REQUIRE((score_w == nullptr) == (score_b == nullptr) && (score_w == nullptr || w_ws != nullptr), TR_ERR_NULL,
             "tr_cluster_merge_layernorm: score weight, bias and the [B,P] weight scratch go together");
  TR_REQUIRE(B > 0 && N >= 2 && K >= 1 && K <= N - 1 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAX_CHUNKS, TR_ERR_SHAPE,
             "tr_cluster_merge_layernorm: bad shape B=%d N=%d K=%d D=%d", B, N, K, D);
  TR_REQUIRE(x_out != x, TR_ERR_SHAPE, "tr_cluster_merge_layernorm: needs a distinct x_out");
  TR_REQUIRE(tr_aligned16(x) && tr_aligned16(x_out) && tr_aligned16(y) && tr_aligned16(gamma) && tr_aligned16(beta), TR_ERR_ALIGN,
             "tr_cluster_merge_layernorm: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  if (score_w) hipLaunchKernelGGL(token_weight_kernel, dim3((B * (N - 1) + 3) / 4), dim3(256), 0, st, x, score_w, score_b, w_ws, B, N, D);
  const int rblocks = (K + 1 + 3) / 4;
  if (y_is_f32)
    TR_DISPATCH_NCH(D, hipLaunchKernelGGL((cluster_merge_layernorm_kernel<true, NCH>), dim3(B * rblocks), dim3(256), 0, st, x,
                                          score_w ? w_ws : nullptr, idx_cluster, gamma, beta, x_out, y, N, K, D, eps));
  else
    TR_DISPATCH_NCH(D, hipLaunchKernelGGL((cluster_merge_layernorm_kernel<false, NCH>), dim3(B * rblocks), dim3(256), 0, st, x,
                                          score_w ? w_ws : nullptr, idx_cluster, gamma, beta, x_out, y, N, K, D, eps));
  TR_CHECK_LAUNCH("tr_cluster_merge_layernorm");
  return TR_OK;
}

// equal_weight branch of k_medoids_fit (kmedoids.py:43-58): the first medoid is one random token id shared by the whole batch
// (np.random.choice on the host: an INPUT here), then K-1 times the token with the largest "max distance to the chosen medoids"
// joins (rows of already chosen tokens are zeroed, kmedoids.py:52-54; torch.max -> first index on ties).  One workgroup per image
// over the [P,P] distance matrix; s_max[i] is kept incrementally (a new medoid only adds one column to the max).
__global__ __launch_bounds__(256) void kmed_init_equal_kernel(const float* __restrict__ dist, int32_t* __restrict__ centers, int P, int K,
                                                              int init) {
  __shared__ float s_max[64 * MAX_PER_LANE];
  __shared__ unsigned char s_ch[64 * MAX_PER_LANE];
  __shared__ unsigned long long s_red[4];
  __shared__ int s_new;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* db = dist + (size_t)b * P * P;
  int32_t* cb = centers + (size_t)b * K;
  if (tid == 0) cb[0] = init;
  for (int i = tid; i < P; i += 256) {
    s_ch[i] = i == init;
    s_max[i] = i == init ? 0.f : db[(size_t)i * P + init];
  }
  for (int k = 1; k < K; ++k) {
    __syncthreads();
    unsigned long long best = 0ull;
    for (int i = tid; i < P; i += 256) {
      const unsigned long long key = ((unsigned long long)__float_as_uint(fmaxf(s_max[i], 0.f)) << 32) | (unsigned int)(0xffffffffu - (unsigned int)i);
      best = key > best ? key : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(best, o, 64);
      best = other > best ? other : best;
    }
    if (lane == 0) s_red[wave] = best;
    __syncthreads();
    if (tid == 0) {
      unsigned long long m = s_red[0];
      for (int w = 1; w < 4; ++w) m = s_red[w] > m ? s_red[w] : m;
      const int nw = (int)(0xffffffffu - (unsigned int)(m & 0xffffffffull));
      s_new = nw;
      cb[k] = nw;
    }
    __syncthreads();
    const int nw = s_new;
    for (int i = tid; i < P; i += 256) {
      if (i == nw) { s_ch[i] = 1; s_max[i] = 0.f; }
      else if (!s_ch[i]) s_max[i] = fmaxf(s_max[i], db[(size_t)i * P + nw]);
    }
  }
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, float v, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

// k_medoids_fit with token_weight = None (args.equal_weight, kmedoids.py:43-58): init_idx = the host's np.random.choice draw.
extern "C" int tr_kmedoids_equal(const float* x, int init_idx, float* ws, int32_t* centers, int32_t* assign, int B, int N, int D, int K,
                                 int iters, int fast_dist, tr_stream_t s) {
  TR_REQUIRE(x && ws && centers && assign, TR_ERR_NULL, "tr_kmedoids_equal: null pointer");
  const int P = N - 1;
  TR_REQUIRE(B > 0 && P >= 2 && P <= 64 * MAX_PER_LANE && D > 0 && D % CK == 0, TR_ERR_SHAPE,
             "tr_kmedoids_equal: need 2 <= P <= %d and D %% %d == 0 (N=%d D=%d)", 64 * MAX_PER_LANE, CK, N, D);
  TR_REQUIRE(K >= 1 && K <= P && iters >= 0 && init_idx >= 0 && init_idx < P, TR_ERR_SHAPE, "tr_kmedoids_equal: bad K=%d / iters=%d / init=%d for P=%d",
             K, iters, init_idx, P);
  TR_REQUIRE(tr_aligned16(x) && tr_aligned16(ws), TR_ERR_ALIGN, "tr_kmedoids_equal: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  float* dist = ws;
  float* nrm = dist + (size_t)B * P * P;
  float* t = nrm + (size_t)B * P;
  float* wrow = t + (size_t)B * P;                              // [B,N] of ones
  const int rows = B * P, rb = (rows + 3) / 4;
  hipLaunchKernelGGL(fill_kernel, dim3((B * N + 255) / 256), dim3(256), 0, st, wrow, 1.0f, (size_t)B * N);
  launch_sqnorm(x, nrm, B, N, D, st, false);
  launch_dist(fast_dist != 0, x, nrm, dist, B, N, D, 1.0f, st);
  hipLaunchKernelGGL(kmed_init_equal_kernel, dim3(B), dim3(256), 0, st, dist, centers, P, K, init_idx);
  hipLaunchKernelGGL(kmed_rowcost_kernel, dim3(rb), dim3(256), 0, st, dist, wrow, t, B, P);
  hipLaunchKernelGGL(kmed_iterate_kernel, dim3(B), dim3(KMT), (size_t)K * 12 + (size_t)P * 8, st, dist, t, centers, assign, P, K, iters);
  TR_CHECK_LAUNCH("tr_kmedoids_equal");
  return TR_OK;
}

extern "C" int tr_kmedoids(const float* x, const float* colsum_part, float* ws, int32_t* centers, int32_t* assign, int B, int N, int D,
                           int H, int K, int iters, int fast_dist, tr_stream_t s) {
  TR_REQUIRE(x && colsum_part && ws && centers && assign, TR_ERR_NULL, "tr_kmedoids: null pointer");
  const int P = N - 1;
  TR_REQUIRE(B > 0 && H > 0 && P >= 2 && P <= 64 * MAX_PER_LANE && D > 0 && D % CK == 0, TR_ERR_SHAPE,
             "tr_kmedoids: need 2 <= P <= %d and D %% %d == 0 (N=%d D=%d)", 64 * MAX_PER_LANE, CK, N, D);
  TR_REQUIRE(K >= 1 && K <= P && iters >= 0, TR_ERR_SHAPE, "tr_kmedoids: bad K=%d / iters=%d for P=%d", K, iters, P);
  TR_REQUIRE(tr_aligned16(x) && tr_aligned16(ws), TR_ERR_ALIGN, "tr_kmedoids: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  float* dist = ws;
  float* nrm = dist + (size_t)B * P * P;
  float* t = nrm + (size_t)B * P;
  float* wrow = t + (size_t)B * P;                              // [B,N]
  const int rows = B * P, rb = (rows + 3) / 4;
  hipLaunchKernelGGL(kmed_weight_kernel, dim3((B * N + 255) / 256), dim3(256), 0, st, colsum_part, wrow, B, N, H);
  TR_CHECK_LAUNCH("tr_kmedoids");
  // cluster_idx = topk(token_weight, K) (kmedoids.py:59); the scores output lands in t and is overwritten below
  int rc = tr_cls_topk(wrow, centers, nullptr, t, B, 1, N, K, s);
  if (rc != TR_OK) return rc;
  launch_sqnorm(x, nrm, B, N, D, st, false);
  launch_dist(fast_dist != 0, x, nrm, dist, B, N, D, 1.0f, st);                               // torch.cdist(x, x)
  hipLaunchKernelGGL(kmed_rowcost_kernel, dim3(rb), dim3(256), 0, st, dist, wrow, t, B, P);
  hipLaunchKernelGGL(kmed_iterate_kernel, dim3(B), dim3(KMT), (size_t)K * 12 + (size_t)P * 8, st, dist, t, centers, assign, P, K, iters);
  TR_CHECK_LAUNCH("tr_kmedoids");
  return TR_OK;
}
