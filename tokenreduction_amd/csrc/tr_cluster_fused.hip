// DPC-KNN clustering (cluster_dpc_knn, models/dpcknn.py:44-100) in ONE launch, one workgroup per image, for up to 208 patch tokens
// (224 x 224 inputs: 196, and every later stage).  SURVEY 8d: the P x P distance matrix "must not" go to HBM -- 196^2 fp32 is 150 KiB.
//
// What tr_dpcknn_cluster used to launch per stage (profiles/r04a_dpcknn_small_*): sqnorm 13 us + dist_mfma 56 us (writes B x P x P
// fp32 = 39 MB) + density 33 us + parent_score 11 us + cls_topk 7 us + assign 10 us = 130 us, the matrix read back three times.
// Here it lives in LDS, as the UPPER TRIANGLE in 16 x 16 tiles (91 tiles x 1088 B = 97 KiB; the matrix is symmetric), next to two
// 16-KiB operand slabs; only centres, assignments and scores leave the chip:
//   0./1. squared norms ride along with the Gram product's slab loop (two fma chains per row, sqnorm_pair_kernel's order: x is read once);
//   1. Gram product on the matrix cores with every fp32 operand split into bf16 hi + lo (the arithmetic of dist_mfma_kernel: per
//      32-deep slab lo.hi' + hi.lo' + hi.hi'), upper-triangle tiles only, tile rows w and 15-w on wave w (13,12,11,11,... tiles);
//      distance = sqrt(max(|x_i|^2 + |x_j|^2 - 2 g_ij, 1e-30)) / sqrt(D) written into the tile;
//   2. density = exp(-mean of the k smallest d^2) (+ noise): FOUR rows per wave, 16 lanes per row -- each lane keeps the five smallest
//      of its 13 elements sorted (insertion by v_med3), four DPP steps merge the lists (min(a_i, b_4-i), then a 9-comparator sort): no
//      index bookkeeping and no cross-lane LDS traffic (the old kernel spent ~60 ds_bpermute per row); the row's elements stay in
//      registers for step 3;
//   3. distance to the nearest denser token (else the image's maximum), score = distance x density;
//   4. top-K centres by rank counting on order keys (ties: lower index first -- tr_cls_topk's rule);
//   5. nearest centre per token: one wave per centre row, (distance bits, centre number) packed into 64 bits and ds_min_u64'ed per token
//      -- the minimum IS "smallest distance, first centre on ties".
// The one deviation from the old pipeline: element (i, j) below the diagonal is read from (j, i).  dist_mfma_kernel computed both, and its
// two small product terms enter in the opposite order there, so its matrix was symmetric only up to the last bit; decisions taken on a
// last-bit difference are ties for every practical purpose (the fixture tests compare all centres and assignments).
#include "tr_common.h"

namespace {

constexpr int FT_NT = 13;                   // tiles per side: P <= 208
constexpr int FT_PMAX = 16 * FT_NT;
constexpr int FT_TROW = 17;                 // floats per tile row (padded: conflict-free row AND column reads)
constexpr int FT_TILE = 16 * FT_TROW;       // 272 floats
constexpr int FT_GLD = 40;                  // operand slab row stride in bf16 (80 B)
constexpr int FT_GK = 32;
constexpr int FT_KSEL = 5;                  // neighbours kept per row (k <= 5)

__device__ __forceinline__ int ft_rowstart(int a, int nt) { return a * nt - ((a * (a - 1)) >> 1); }   // index of tile (a, a)

template <int CTRL>
__device__ __forceinline__ float ft_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: four involutions that pair disjoint subsets of a 16-lane row
#define FT_ROW16_REDUCE(EXPR_STEP)  EXPR_STEP(0xB1) EXPR_STEP(0x4E) EXPR_STEP(0x141) EXPR_STEP(0x140)

__device__ __forceinline__ void ft_cmpx(float& a, float& b) {
  const float lo = fminf(a, b), hi = fmaxf(a, b);
  a = lo;
  b = hi;
}

__device__ __forceinline__ void ft_split_store(const float4 v, unsigned short* hi, unsigned short* lo) {
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

#ifdef TR_FUSED_STAMPS
__device__ unsigned long long ft_stamps[8];        // lab only: s_memtime at the phase boundaries of workgroup 17
#define FT_STAMP(k) do { if (blockIdx.x == 17 && threadIdx.x == 0) ft_stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FT_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(512) void dpcknn_fused_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                           int32_t* __restrict__ centers, int32_t* __restrict__ idx_cluster,
                                                           float* __restrict__ scores, int N, int D, int K, int k, float sqrt_d) {
  __shared__ __attribute__((aligned(16))) float T[91 * FT_TILE];                      // 99,008 B
  __shared__ __attribute__((aligned(16))) unsigned short sH[FT_PMAX * FT_GLD], sL[FT_PMAX * FT_GLD];   // 2 x 16,640 B
  __shared__ float s_nrm[FT_PMAX], s_den[FT_PMAX], s_rmax[FT_PMAX], s_noise[FT_PMAX];
  __shared__ unsigned int s_key[FT_PMAX];
  __shared__ int s_cen[FT_PMAX];
  __shared__ unsigned long long s_best[FT_PMAX];
  __shared__ float s_dmax;
  const int P = N - 1, nt = (P + 15) >> 4;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* xb = x + ((size_t)b * N + 1) * D;

  FT_STAMP(0);
  // the density noise of this image goes to LDS up front: a global load inside the density loop stalls its wave for a microsecond per row group
  if (tid < P) s_noise[tid] = noise ? noise[(size_t)b * P + tid] : 0.f;
  FT_STAMP(1);
  // ---- 1. Gram product, upper-triangle tiles: wave w owns tile rows w (13 - w tiles at nt = 13) and 15 - w
  const int frow = lane & 15, fq = lane >> 4;
  const int ra = wave, rb2 = 15 - wave;
  const int na = ra < nt ? nt - ra : 0, nb = rb2 < nt ? nt - rb2 : 0;
  f32x4 accA[FT_NT], accB[5];
#pragma unroll
  for (int s = 0; s < FT_NT; ++s) accA[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 5; ++s) accB[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const int lr = tid >> 1, lc = (tid & 1) * 16;                      // staging: row lr (0..255), 16 consecutive floats
    const bool stg = lr < FT_PMAX;
    const float* ap = xb + (size_t)min(lr, P - 1) * D + lc;
    // two slabs of prefetch: the loads of slab s+2 are issued before slab s is split (the per-slab matrix work, ~1.2k cycles, is shorter
    // than a first-touch load; one slab of prefetch left ~2k cycles of every 3.1k-cycle slab step exposed)
    float4 rv[4], rw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const float4*>(ap + 4 * q);
#pragma unroll
    for (int q = 0; q < 4; ++q) rw[q] = *reinterpret_cast<const float4*>(ap + min(FT_GK, D - FT_GK) + 4 * q);
    // |x_row|^2 rides along: each of a row's two threads chains fma over its 16 floats of every slab, slab after slab; the row's norm is
    // the sum of the two chains (sqnorm_kernel adds in the same order, so the staged launches see the same norms bit for bit)
    float nacc = 0.f;
    for (int k0 = 0; k0 < D; k0 += FT_GK) {
      lds_barrier();                                                   // the slab before has been read (LDS only: the prefetch stays in flight)
      if (stg) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ft_split_store(rv[q], sH + lr * FT_GLD + lc + 4 * q, sL + lr * FT_GLD + lc + 4 * q);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        nacc = fmaf(rv[q].x, rv[q].x, nacc); nacc = fmaf(rv[q].y, rv[q].y, nacc);
        nacc = fmaf(rv[q].z, rv[q].z, nacc); nacc = fmaf(rv[q].w, rv[q].w, nacc);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) rv[q] = rw[q];
      if (k0 + 2 * FT_GK < D) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rw[q] = *reinterpret_cast<const float4*>(ap + k0 + 2 * FT_GK + 4 * q);
      }
      lds_barrier();
      // operand fragments: the A side is the tile ROW (i), the B side the tile COLUMN (j); acc[e] = x_i . x_j, i = frow, j = 4 fq + e
      if (na > 0) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sH + (ra * 16 + frow) * FT_GLD + fq * 8);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(sL + (ra * 16 + frow) * FT_GLD + fq * 8);
#pragma unroll
        for (int s = 0; s < FT_NT; ++s) {
          if (s < na) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sH + ((ra + s) * 16 + frow) * FT_GLD + fq * 8);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(sL + ((ra + s) * 16 + frow) * FT_GLD + fq * 8);
            // small terms first, then the dominant one (dist_mfma_kernel's order)
            accA[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, accA[s], 0, 0, 0);
            accA[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, accA[s], 0, 0, 0);
            accA[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, accA[s], 0, 0, 0);
          }
        }
      }
      if (nb > 0) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sH + (rb2 * 16 + frow) * FT_GLD + fq * 8);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(sL + (rb2 * 16 + frow) * FT_GLD + fq * 8);
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          if (s < nb) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sH + ((rb2 + s) * 16 + frow) * FT_GLD + fq * 8);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(sL + ((rb2 + s) * 16 + frow) * FT_GLD + fq * 8);
            accB[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, accB[s], 0, 0, 0);
            accB[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, accB[s], 0, 0, 0);
            accB[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, accB[s], 0, 0, 0);
          }
        }
      }
    }
    {
      const float other = ft_dpp<0xB1>(nacc);          // the row's other thread (tid ^ 1)
      if ((tid & 1) == 0 && lr < P) s_nrm[lr] = nacc + other;
    }
  }
  FT_STAMP(2);
  __syncthreads();                                   // s_nrm complete
  {
    auto put = [&](const f32x4 g, int ti, int tj) __attribute__((always_inline)) {
      const int i = ti * 16 + frow;
      const float ni = s_nrm[min(i, P - 1)];
      float* t = T + (ft_rowstart(ti, nt) + (tj - ti)) * FT_TILE + frow * FT_TROW + 4 * fq;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = tj * 16 + 4 * fq + e;
        const float nj = s_nrm[min(j, P - 1)];
        t[e] = sqrtf(fmaxf((ni + nj) - 2.0f * g[e], 1e-30f)) / sqrt_d;
      }
    };
#pragma unroll
    for (int s = 0; s < FT_NT; ++s)
      if (s < na) put(accA[s], ra, ra + s);
#pragma unroll
    for (int s = 0; s < 5; ++s)
      if (s < nb) put(accB[s], rb2, rb2 + s);
  }
  __syncthreads();

  FT_STAMP(3);
  // Element (i, j) of the symmetric matrix for a lane that walks row i with j = m * 16 + sl (tile column m, lane column sl):
  //   m >= ti: tile (ti, m), row r, column sl      -> baseU + m * 272          baseU = (rowstart(ti) - ti) * 272 + r * 17 + sl
  //   m <  ti: tile (m, ti), row sl, column r      -> baseL + m * stepL - cL   baseL = ti * 272 + sl * 17 + r, stepL = (nt - 1) * 272,
  //                                                                             cL = m (m - 1) / 2 * 272  (compile time)
  // Two adds, a compare and a select per element.  A lane keeps the 13 elements of each of its (up to 7) rows in registers from the
  // density pass to the parent-distance pass: the matrix is read once.
  const int grp = lane >> 4, sl = lane & 15;
  const int stepL = (nt - 1) * FT_TILE;
  constexpr int NPASS = (FT_PMAX + 31) / 32;           // 7
  float dv[NPASS][FT_NT];
  float dj[FT_NT];                                     // densities of this lane's columns (phase 3)

  // ---- 2. density and row maximum: four rows per wave and pass, 16 lanes per row
  // (two rows per lane group at a time: the sorted insertion is one long dependency chain per row, a second row doubles the work in flight)
  auto density_pass = [&](const int ps, float (&bsm)[FT_KSEL], float& mx) __attribute__((always_inline)) {
    const int i = min(ps * 32 + wave * 4 + grp, P - 1);
    const int ti = i >> 4, r = i & 15;
    const int baseU = (ft_rowstart(ti, nt) - ti) * FT_TILE + r * FT_TROW + sl;
    int offL = ti * FT_TILE + sl * FT_TROW + r;          // running: + stepL - m * 272 per tile column
#pragma unroll
    for (int t = 0; t < FT_KSEL; ++t) bsm[t] = INFINITY;
    mx = 0.f;
    // branch-free: every lane loads (a clamped, in-range address where its element does not exist) and selects -- with `if`s around the
    // load hipcc built an exec-mask branch per element and spilled scalar registers into VGPR lanes (8.5k instructions for this phase)
#pragma unroll
    for (int m = 0; m < FT_NT; ++m) {
      const int j = m * 16 + sl;
      const int mc = min(m, nt - 1);
      const int offU = baseU + mc * FT_TILE;
      const int off = (m >= ti) ? offU : offL;
      const float d = T[off];
      const bool ok = (m < nt) && (j < P);
      float v = ok ? d : INFINITY;
      mx = fmaxf(mx, ok ? d : 0.f);
      offL += stepL - m * FT_TILE;
      dv[ps][m] = v;
      // insert into the sorted five: new[t] = median(old[t-1], old[t], v) -- five INDEPENDENT full-rate ops (a compare-exchange chain is
      // ten dependent ones)
      {
        float nb_[FT_KSEL];
        nb_[0] = fminf(bsm[0], v);
#pragma unroll
        for (int t = 1; t < FT_KSEL; ++t) nb_[t] = __builtin_amdgcn_fmed3f(bsm[t - 1], bsm[t], v);
#pragma unroll
        for (int t = 0; t < FT_KSEL; ++t) bsm[t] = nb_[t];
      }
    }
  };
  auto density_finish = [&](const int ps, float (&bsm)[FT_KSEL], float mx) __attribute__((always_inline)) {
#define FT_MERGE(CTRL)                                                                                          \
    {                                                                                                           \
      float o[FT_KSEL];                                                                                         \
      _Pragma("unroll") for (int t = 0; t < FT_KSEL; ++t) o[t] = ft_dpp<CTRL>(bsm[t]);                         \
      _Pragma("unroll") for (int t = 0; t < FT_KSEL; ++t) bsm[t] = fminf(bsm[t], o[FT_KSEL - 1 - t]);          \
      ft_cmpx(bsm[0], bsm[1]); ft_cmpx(bsm[3], bsm[4]); ft_cmpx(bsm[2], bsm[4]); ft_cmpx(bsm[2], bsm[3]);       \
      ft_cmpx(bsm[0], bsm[3]); ft_cmpx(bsm[0], bsm[2]); ft_cmpx(bsm[1], bsm[4]); ft_cmpx(bsm[1], bsm[3]);       \
      ft_cmpx(bsm[1], bsm[2]);                                                                                  \
      mx = fmaxf(mx, ft_dpp<CTRL>(mx));                                                                         \
    }
    FT_ROW16_REDUCE(FT_MERGE)
#undef FT_MERGE
    const int i = ps * 32 + wave * 4 + grp;
    if (sl == 0 && i < P) {
      float ss = 0.f;
#pragma unroll
      for (int t = 0; t < FT_KSEL; ++t)
        if (t < k) ss += bsm[t] * bsm[t];                                // ascending = torch.topk(largest=False) order
      s_den[i] = expf(-(ss / (float)k)) + (noise ? s_noise[i] * 1e-6f : 0.f);
      s_rmax[i] = mx;
    }
  };
#pragma unroll
  for (int ps = 0; ps < NPASS; ps += 2) {
    if (ps * 32 + wave * 4 < P) {                        // (the second row of the pair may lie beyond P: clamped, computed, not written)
      float b0[FT_KSEL], b1[FT_KSEL], m0, m1;
      density_pass(ps, b0, m0);
      if (ps + 1 < NPASS) density_pass(ps + 1, b1, m1);
      density_finish(ps, b0, m0);
      if (ps + 1 < NPASS) density_finish(ps + 1, b1, m1);
    }
  }
  __syncthreads();
  FT_STAMP(4);
  if (wave == 0) {                                   // dist_matrix.flatten(1).max(): max of the row maxima
    float m = 0.f;
    for (int j = lane; j < P; j += 64) m = fmaxf(m, s_rmax[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) s_dmax = m;
  }
  for (int p = tid; p < P; p += 512) s_best[p] = ~0ull;
#pragma unroll
  for (int m = 0; m < FT_NT; ++m) dj[m] = s_den[min(m * 16 + sl, P - 1)];
  __syncthreads();

  // ---- 3. distance to the nearest denser token, score, order key (the rows' elements are still in registers)
  const float dmax = s_dmax;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int i0 = ps * 32 + wave * 4;
    if (i0 < P) {
      const int i = min(i0 + grp, P - 1);
      const float di = s_den[i];
      float mn = INFINITY;
#pragma unroll
      for (int m = 0; m < FT_NT; ++m) {
        const int j = m * 16 + sl;
        if (m < nt && j < P) mn = fminf(mn, dj[m] > di ? dv[ps][m] : dmax);
      }
#define FT_MIN(CTRL) mn = fminf(mn, ft_dpp<CTRL>(mn));
      FT_ROW16_REDUCE(FT_MIN)
#undef FT_MIN
      if (sl == 0 && i0 + grp < P) {
        const float sc = mn * di;
        scores[(size_t)b * P + i] = sc;
        unsigned int u = __float_as_uint(sc + 0.0f);                       // tr_cls_topk's order key: -0 -> +0, NaN above +inf
        u = (sc != sc) ? 0xffffffffu : (u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u));
        s_key[i] = u;
      }
    }
  }
  __syncthreads();

  FT_STAMP(5);
  // ---- 4. the K highest scores, in descending order (ties: lower index first): two threads per token, half of the scores each
  {
    const int i = tid >> 1, hf = tid & 1;
    const int ic = min(i, P - 1);
    const unsigned int si = s_key[ic];
    const int half = (P + 1) >> 1;
    const int j0 = hf * half, j1 = min(P, j0 + half);
    int rank = 0;
    for (int j = j0; j < j1; ++j) {
      const unsigned int sj = s_key[j];
      rank += (sj > si) || (sj == si && j < ic);
    }
    rank += __builtin_amdgcn_mov_dpp(rank, 0xB1, 0xF, 0xF, true);          // the neighbour lane's half
    if (hf == 0 && i < P && rank < K) {
      s_cen[rank] = i;
      centers[(size_t)b * K + rank] = i;
    }
  }
  __syncthreads();

  FT_STAMP(6);
  // ---- 5. nearest centre per token: centre c on wave c % 8, its row against every token, (distance, centre) minimum per token in LDS
  {
    int pu[4], pl[4], pm[4];                              // per lane token p = q * 64 + lane: the lane parts of the two address forms
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = q * 64 + lane, m = p >> 4, psl = p & 15;
      pm[q] = p < P ? m : -1;                              // -1: no such token (never >= ti, and skipped below)
      pu[q] = m * FT_TILE + psl;
      pl[q] = (ft_rowstart(min(m, nt - 1), nt) - m) * FT_TILE + psl * FT_TROW;
    }
    for (int c = wave; c < K; c += 8) {
      const int ci = s_cen[c];                             // wave-uniform
      const int ti = ci >> 4, r = ci & 15;
      const int uU = (ft_rowstart(ti, nt) - ti) * FT_TILE + r * FT_TROW, uL = ti * FT_TILE + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (pm[q] >= 0) {
          const int off = (pm[q] >= ti) ? uU + pu[q] : uL + pl[q];
          const unsigned long long key = ((unsigned long long)__float_as_uint(T[off]) << 32) | (unsigned int)c;    // distances are > 0
          atomicMin(&s_best[q * 64 + lane], key);
        }
      }
    }
  }
  __syncthreads();
  for (int p = tid; p < P; p += 512) idx_cluster[(size_t)b * P + p] = (int)(unsigned int)(s_best[p] & 0xffffffffu);
  __syncthreads();                                   // the centres' own entries must land after the pass above (same addresses)
  for (int c = tid; c < K; c += 512) idx_cluster[(size_t)b * P + s_cen[c]] = c;
  FT_STAMP(7);
}

}  // namespace

#ifdef TR_FUSED_STAMPS
extern "C" void ftdbg_read(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ft_stamps), sizeof(unsigned long long) * 8); }
#endif

// Does the one-launch kernel take this problem?  (fast distances only: the fp32 validation executor keeps the exact VALU pipeline)
extern "C" int tr_dpcknn_fused_supported(int N, int D, int k) {
  const int P = N - 1;
  return P > 25 && P <= FT_PMAX && D % FT_GK == 0 && D % 4 == 0 && D <= 1024 && k >= 1 && k <= FT_KSEL && k <= P;
}

extern "C" int tr_dpcknn_cluster_fused(const float* x, const float* noise, int32_t* centers, int32_t* idx_cluster, float* scores, int B, int N,
                                       int D, int K, int k, tr_stream_t s) {
  TR_REQUIRE(x && centers && idx_cluster && scores, TR_ERR_NULL, "tr_dpcknn_cluster_fused: null pointer");
  TR_REQUIRE(B > 0 && tr_dpcknn_fused_supported(N, D, k), TR_ERR_SHAPE,
             "tr_dpcknn_cluster_fused: need 26 <= P <= %d, D %% 32 == 0, D <= 1024, k <= %d (N=%d D=%d k=%d)", FT_PMAX, FT_KSEL, N, D, k);
  TR_REQUIRE(K >= 1 && K <= N - 1, TR_ERR_SHAPE, "tr_dpcknn_cluster_fused: bad K=%d for P=%d", K, N - 1);
  TR_REQUIRE(tr_aligned16(x), TR_ERR_ALIGN, "tr_dpcknn_cluster_fused: x must be 16-byte aligned");
  // algorithmic traffic: the tokens in, centres + assignments + scores out
  tr_prof_note("dpcknn_fused_kernel", 3.0 * 2.0 * (N - 1) * (double)(N - 1) / 2 * D * B, ((double)(N - 1) * D * 4 + 4.0 * K + 8.0 * (N - 1)) * B);
  hipLaunchKernelGGL(dpcknn_fused_kernel, dim3(B), dim3(512), 0, static_cast<hipStream_t>(s), x, noise, centers, idx_cluster, scores, N, D, K, k,
                     (float)sqrt((double)D));
  TR_CHECK_LAUNCH("tr_dpcknn_cluster_fused");
  return TR_OK;
}
