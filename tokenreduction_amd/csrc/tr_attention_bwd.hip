// Backward of multi-head self-attention (head_dim 64, N <= 224 tokens) on gfx950 MFMA.
//
// Gradient of  attn = softmax(q k^T * dh^-0.5 [+ log size]); out = attn @ v   (topk.py:44-51, tome.py:41-58) with respect to
// q, k, v, given d out -- what torch.autograd derives for the reference's eager ops -- plus EViT's extra path: the fused token
// multiplies by cls_attn = attn[:, :, 0, 1:].mean(1) (evit.py:117-120), so d cls_attn / H is added to dP of query 0.
//
// Flash-style: nothing of the N x N matrix is kept by the forward; P is recomputed here from q, k (the whole score row of a
// query fits the register file at N <= 224, so the row maximum and normaliser are recomputed exactly, no saved LSE).
// One workgroup (4 waves, one per SIMD) per (image, head):
//   K, V of all keys stay in LDS (row-major images, read by rows for S and dP and TRANSPOSED -- ds_read_b64_tr_b16 -- for dQ);
//   queries go in blocks of 64 (wave w owns rows 16w .. 16w+15 of a block):
//     S^T = K Q^T, dP^T = V dO^T       key on the accumulator ROW: a lane holds 4 consecutive keys of one query, so the
//                                       softmax is lane-local + two shuffles, and P / dS go to LDS as 8-byte writes
//     p = softmax, delta = sum_j p dp, ds = p (dp - delta) * dh^-0.5
//     dQ^T = K^T dS^T                  (each wave on its own 16 queries; 8-byte stores of 4 consecutive d)
//     dK^T += Q^T dS, dV^T += dO^T P   accumulated in registers over the query blocks, key tiles split over the waves
// LDS images use one swizzle each that is conflict-free for both the 16-byte row reads and the transposed reads
// (searched with tools/lds_sim.py): 128-B rows: chunk ^ (row bit1 << 1 | row bit3 << 2); 512-B rows (P, dS): chunk ^ f16(row).
#include "tr_common.h"
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ bf16x8 lds_tr_pair(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
  const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}

// reductions over the four lanes (lane >> 4 = 0..3) that share a query: two VALU row / half swaps (v_permlane16_swap, v_permlane32_swap)
// instead of two ds_bpermute round trips each -- see tr_attention.hip (the builtin's result elements are copied to scalars first:
// __builtin_bit_cast of `a[1]` directly read element 0, hipcc ROCm 7.2)
__device__ __forceinline__ float bq_rows_max(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned a0 = a[0], a1 = a[1];
  v = fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1));
  const auto c = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned c0 = c[0], c1 = c[1];
  return fmaxf(__builtin_bit_cast(float, c0), __builtin_bit_cast(float, c1));
}
__device__ __forceinline__ float bq_rows_sum(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned a0 = a[0], a1 = a[1];
  v = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
  const auto c = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned c0 = c[0], c1 = c[1];
  return __builtin_bit_cast(float, c0) + __builtin_bit_cast(float, c1);
}

// [rows][64 bf16] image, 128-B rows
__device__ __forceinline__ int qswz(int row, int ch) { return row * 128 + ((ch ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2))) << 4); }
// [64 rows][256 bf16] image, 512-B rows
__device__ __forceinline__ int pswz(int row, int ch) { return row * 512 + ((ch ^ ((row & 15) ^ ((0 - (row & 1)) & 14))) << 4); }

// POLICY (DyViT training, Policy_Attention.softmax_with_policy dyvit.py:39-51): `size` carries the keep policy [B,N] of 1/0,
//   e = exp(s - max), a = e * pi (pi[q][k] = policy[k], 1 on the diagonal), p = (a + eps/N) / (sum a + eps), eps = 1e-6;
//   with w = dp - sum_k p_k dp_k:  ds = w * a / (sum a + eps),  d policy[k] += w * e / (sum a + eps) over all queries q != k and all
//   heads (the straight-through Gumbel sample upstream makes the policy differentiable, dyvit.py:223-224).  The per-key sums are
//   accumulated in registers and combined in a fixed order, written per (image, head) to dpol_part [B,H,N].
template <int NKB, bool BIAS, bool POLICY>
__global__ __launch_bounds__(256, 1) void attention_bwd_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dO,
                                                               const float* __restrict__ size, const float* __restrict__ dcls,
                                                               uint16_t* __restrict__ dqkv, float* __restrict__ dpol_part, int N, int H) {
  constexpr int NP = NKB * 32;       // padded key count
  constexpr int NT = NKB * 2;        // 16-key tiles
  constexpr int KT_W = (NT + 3) / 4; // key tiles a wave accumulates dK / dV for
  __shared__ __attribute__((aligned(16))) unsigned char sK[NP * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sV[NP * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sQ[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sDO[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sP[64 * 512];
  __shared__ __attribute__((aligned(16))) unsigned char sDS[64 * 512];
  __shared__ __attribute__((aligned(16))) float sLB[NP];     // log2(size[key]) (ToMe / key masks), 0 without sizes
  __shared__ __attribute__((aligned(16))) float sDC[NP];     // d cls_attn[key] / H (EViT), 0 otherwise
  // POLICY: d policy[key], one row per (query block, wave): the wave's 16 query columns are reduced over the lanes (DPP) and stored by one
  // lane; the rows are added at the end in a fixed order.  As LDS float atomics straight from every lane -- sixteen lanes of a quarter wave
  // on one address, N^2 atomics per head -- this sum made the policy variant 3x the plain kernel (601 vs 196 us at N = 197); as a
  // read-modify-write per entry it still waited out 56 LDS round trips per block (346 us).
  __shared__ __attribute__((aligned(16))) float sPol[POLICY ? 16 * NP : 1];      // [query block (<= 4)][wave][key]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int ldq = 3 * H * 64, ldo = H * 64;
  const uint16_t* base = qkv + (size_t)b * N * ldq;
  const uint16_t* dobase = dO + (size_t)b * N * ldo + h * 64;
  uint16_t* dbase = dqkv + (size_t)b * N * ldq;
  const int qcol = h * 64, kcol = H * 64 + h * 64, vcol = 2 * H * 64 + h * 64;

  // ---- stage K and V (keys >= N: zero rows); all loads first, then the LDS writes
  {
    uint4 kreg[NKB], vreg[NKB];
#pragma unroll
    for (int it = 0; it < NKB; ++it) {
      const int c = tid + 256 * it, key = c >> 3, ch = c & 7;
      kreg[it] = *reinterpret_cast<const uint4*>(base + (size_t)min(key, N - 1) * ldq + kcol + ch * 8);
      vreg[it] = *reinterpret_cast<const uint4*>(base + (size_t)min(key, N - 1) * ldq + vcol + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < NKB; ++it) {
      const int c = tid + 256 * it, key = c >> 3, ch = c & 7;
      if (key >= N) {
        kreg[it] = make_uint4(0u, 0u, 0u, 0u);
        vreg[it] = make_uint4(0u, 0u, 0u, 0u);
      }
      *reinterpret_cast<uint4*>(sK + qswz(key, ch)) = kreg[it];
      *reinterpret_cast<uint4*>(sV + qswz(key, ch)) = vreg[it];
    }
    for (int key = tid; key < NP; key += 256) {
      if (POLICY) {
        sLB[key] = key < N ? size[(size_t)b * N + key] : 0.f;
        sDC[key] = 0.f;
      } else {
        sLB[key] = (BIAS && key < N) ? __builtin_amdgcn_logf(size[(size_t)b * N + key]) : 0.f;     // v_log_f32 = log2
        sDC[key] = (dcls != nullptr && key < N) ? dcls[(size_t)b * N + key] / (float)H : 0.f;
      }
    }
  }

  const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
  const float c_exp = 0.125f * 1.44269504088896340736f;    // dh^-0.5 * log2(e)

  f32x4 dk[KT_W][4], dv[KT_W][4];
#pragma unroll
  for (int t = 0; t < KT_W; ++t)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      dk[t][d] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[t][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  const int nqb = (N + 63) >> 6;
  // Q and dO rows of a query block travel global -> registers -> LDS.  The loads of block qb + 1 are issued right after block qb's
  // registers have been written to LDS, so they fly under block qb's phases (the barriers inside a block wait for LDS only).
  uint4 qreg[2], oreg[2];
  auto load_block = [&](int qb) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + 256 * it, r = c >> 3, ch = c & 7;
      const int i = qb * 64 + r;
      qreg[it] = *reinterpret_cast<const uint4*>(base + (size_t)min(i, N - 1) * ldq + qcol + ch * 8);
      oreg[it] = *reinterpret_cast<const uint4*>(dobase + (size_t)min(i, N - 1) * ldo + ch * 8);
    }
  };
  load_block(0);
  for (int qb = 0; qb < nqb; ++qb) {
    // ---- stage this block's Q and dO rows (rows >= N: zero)
    {
      __syncthreads();                // every wave is done with the previous block's Q, dO, P, dS images (and the loads have landed)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int c = tid + 256 * it, r = c >> 3, ch = c & 7;
        if (qb * 64 + r >= N) {
          qreg[it] = make_uint4(0u, 0u, 0u, 0u);
          oreg[it] = make_uint4(0u, 0u, 0u, 0u);
        }
        *reinterpret_cast<uint4*>(sQ + qswz(r, ch)) = qreg[it];
        *reinterpret_cast<uint4*>(sDO + qswz(r, ch)) = oreg[it];
      }
      if (qb + 1 < nqb) load_block(qb + 1);
      lds_barrier();
    }

    // ---- phase 1: this wave's 16 queries (rows 16*wave + li of the block) against all keys
    const int il = 16 * wave + li;                      // query row inside the block (this lane's accumulator column)
    const int iq = qb * 64 + il;                        // global query index
    bf16x8 qf[2], of[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(sQ + qswz(il, 4 * ks + g));
      of[ks] = *reinterpret_cast<const bf16x8*>(sDO + qswz(il, 4 * ks + g));
    }
    f32x4 sacc[NT], dpacc[NT];
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};     // C = 0 as the inline constant of a chain's first MFMA (no v_mov per accumulator register)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + qswz(16 * jt + li, 4 * ks + g));
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + qswz(16 * jt + li, 4 * ks + g));
        sacc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], ks == 0 ? z4 : sacc[jt], 0, 0, 0);     // S^T[key 4g+r][query li]
        dpacc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, of[ks], ks == 0 ? z4 : dpacc[jt], 0, 0, 0);   // dP^T
      }
    }
    // softmax over the keys of query li: registers (jt, r) of this lane and the lanes li + 16, 32, 48
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const float4 lb = *reinterpret_cast<const float4*>(&sLB[16 * jt + 4 * g]);
      const float lbv[4] = {lb.x, lb.y, lb.z, lb.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = POLICY ? sacc[jt][r] * c_exp : sacc[jt][r] * c_exp + lbv[r];
        if (16 * jt + 4 * g + r >= N) t = -INFINITY;
        sacc[jt][r] = t;
        mx = fmaxf(mx, t);
      }
    }
    mx = bq_rows_max(mx);
    if (mx == -INFINITY) mx = 0.f;                      // every key masked: all weights 0 (as the forward)
    float l = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const float4 lb = *reinterpret_cast<const float4*>(&sLB[16 * jt + 4 * g]);
      const float lbv[4] = {lb.x, lb.y, lb.z, lb.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pe = __builtin_amdgcn_exp2f(sacc[jt][r] - mx);
        sacc[jt][r] = pe;                                 // POLICY: e (the policy is applied below, e itself is needed for d policy)
        l += POLICY ? pe * ((16 * jt + 4 * g + r == iq) ? 1.0f : lbv[r]) : pe;
      }
    }
    l = bq_rows_sum(l);
    const float inv = POLICY ? 1.0f / (l + 1e-6f) : (l > 0.f ? 1.0f / l : 0.f);
    const float padd = POLICY ? 1e-6f / (float)N : 0.f;
    float dl = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const float4 dc = *reinterpret_cast<const float4*>(&sDC[16 * jt + 4 * g]);
      const float dcv[4] = {dc.x, dc.y, dc.z, dc.w};
      const float4 lb = *reinterpret_cast<const float4*>(&sLB[16 * jt + 4 * g]);
      const float lbv[4] = {lb.x, lb.y, lb.z, lb.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * jt + 4 * g + r;
        float pn;
        if (POLICY) {
          const float pi = key == iq ? 1.0f : lbv[r];
          pn = key < N ? (sacc[jt][r] * pi + padd) * inv : 0.f;
        } else {
          pn = sacc[jt][r] * inv;
          sacc[jt][r] = pn;
          if (iq == 0) dpacc[jt][r] += dcv[r];           // EViT: d cls_attn reaches the CLS query's row (key 0 carries 0)
        }
        dl += pn * dpacc[jt][r];
      }
    }
    dl = bq_rows_sum(dl);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const float4 lb = *reinterpret_cast<const float4*>(&sLB[16 * jt + 4 * g]);
      const float lbv[4] = {lb.x, lb.y, lb.z, lb.w};
      float pv[4], dsv[4], dpv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * jt + 4 * g + r;
        const float w = dpacc[jt][r] - dl;
        if (POLICY) {
          const float pi = key == iq ? 1.0f : lbv[r];
          const float en = sacc[jt][r] * inv;             // e / (sum a + eps)
          pv[r] = key < N ? en * pi + padd * inv : 0.f;
          dsv[r] = w * en * pi * 0.125f;
          dpv[r] = row16_sum((key != iq && key < N && iq < N) ? w * en : 0.f);      // summed over this wave's 16 query columns (lanes li)
        } else {
          pv[r] = sacc[jt][r];
          dsv[r] = sacc[jt][r] * w * 0.125f;
        }
      }
      if (POLICY && li == 0)       // this (query block, wave)'s own row: a plain 16-byte store, nothing to wait for
        *reinterpret_cast<float4*>(&sPol[(qb * 4 + wave) * NP + 16 * jt + 4 * g]) = make_float4(dpv[0], dpv[1], dpv[2], dpv[3]);
      uint2 pp, ds;
      pp.x = pack_bf16x2(pv[0], pv[1]);
      pp.y = pack_bf16x2(pv[2], pv[3]);
      ds.x = pack_bf16x2(dsv[0], dsv[1]);
      ds.y = pack_bf16x2(dsv[2], dsv[3]);
      const int off = pswz(il, 2 * jt + (g >> 1)) + 8 * (g & 1);     // keys 16jt + 4g .. +3 of row il
      *reinterpret_cast<uint2*>(sP + off) = pp;
      *reinterpret_cast<uint2*>(sDS + off) = ds;
    }
    // ---- dQ^T[d][query] = sum_key K[key][d] dS[query][key]: own rows only (LDS operations of one wave are ordered)
    {
      f32x4 dq[4];
#pragma unroll
      for (int ks = 0; ks < NKB; ++ks) {
        const bf16x8 dsf = *reinterpret_cast<const bf16x8*>(sDS + pswz(il, 4 * ks + g));          // B[k = key][col = query]
        const int r0 = 32 * ks + 8 * g + q4;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int ch = 2 * d + (p4 >> 1);
          const bf16x8 ktf = lds_tr_pair(sK + qswz(r0, ch) + 8 * (p4 & 1), sK + qswz(r0 + 4, ch) + 8 * (p4 & 1));   // A[row = d][k = key]
          dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, ks == 0 ? z4 : dq[d], 0, 0, 0);
        }
      }
      if (iq < N) {
        uint16_t* qrow = dbase + (size_t)iq * ldq + qcol + 4 * g;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          uint2 pk;
          pk.x = pack_bf16x2(dq[d][0], dq[d][1]);
          pk.y = pack_bf16x2(dq[d][2], dq[d][3]);
          *reinterpret_cast<uint2*>(qrow + 16 * d) = pk;
        }
      }
    }
    lds_barrier();                    // P and dS rows of all four waves are in LDS (LDS-only wait: the next block's loads stay in flight)
    // ---- phase 2: dK^T[d][key] += sum_query Q[query][d] dS[query][key], dV^T[d][key] += sum_query dO[query][d] P[query][key]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = 32 * ks + 8 * g + q4;
      bf16x8 qt[4], ot[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int ch = 2 * d + (p4 >> 1);
        qt[d] = lds_tr_pair(sQ + qswz(r0, ch) + 8 * (p4 & 1), sQ + qswz(r0 + 4, ch) + 8 * (p4 & 1));       // A[row = d][k = query]
        ot[d] = lds_tr_pair(sDO + qswz(r0, ch) + 8 * (p4 & 1), sDO + qswz(r0 + 4, ch) + 8 * (p4 & 1));
      }
#pragma unroll
      for (int t = 0; t < KT_W; ++t) {
        const int jt = wave + 4 * t;
        if (jt < NT) {                  // wave-uniform
          const int ch = 2 * jt + (p4 >> 1);
          const bf16x8 dsf = lds_tr_pair(sDS + pswz(r0, ch) + 8 * (p4 & 1), sDS + pswz(r0 + 4, ch) + 8 * (p4 & 1));   // B[k = query][col = key]
          const bf16x8 pf = lds_tr_pair(sP + pswz(r0, ch) + 8 * (p4 & 1), sP + pswz(r0 + 4, ch) + 8 * (p4 & 1));
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            dk[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt[d], dsf, dk[t][d], 0, 0, 0);
            dv[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot[d], pf, dv[t][d], 0, 0, 0);
          }
        }
      }
    }
  }
  if (POLICY) {
    // the rows of every (query block, wave), added in that order
    __syncthreads();
    for (int key = tid; key < N; key += 256) {
      float a = 0.f;
      for (int rw = 0; rw < 4 * nqb; ++rw) a += sPol[rw * NP + key];
      dpol_part[((size_t)b * H + h) * N + key] = a;
    }
  }
  // ---- dK, dV rows: accumulator (t, d): rows d-index 16d + 4g + r, column key 16*(wave + 4t) + li
#pragma unroll
  for (int t = 0; t < KT_W; ++t) {
    const int key = 16 * (wave + 4 * t) + li;
    if (wave + 4 * t < NT && key < N) {
      uint16_t* krow = dbase + (size_t)key * ldq + kcol + 4 * g;
      uint16_t* vrow = dbase + (size_t)key * ldq + vcol + 4 * g;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint2 pk;
        pk.x = pack_bf16x2(dk[t][d][0], dk[t][d][1]);
        pk.y = pack_bf16x2(dk[t][d][2], dk[t][d][3]);
        *reinterpret_cast<uint2*>(krow + 16 * d) = pk;
        pk.x = pack_bf16x2(dv[t][d][0], dv[t][d][1]);
        pk.y = pack_bf16x2(dv[t][d][2], dv[t][d][3]);
        *reinterpret_cast<uint2*>(vrow + 16 * d) = pk;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3: the same algorithm on EIGHT waves (two per SIMD).  The four-wave kernel above runs one wave per SIMD through five dependent
// phases per query block (S/dP MFMAs -> softmax -> LDS writes -> dQ MFMAs -> barrier -> dK/dV MFMAs): the matrix pipe was busy 12 % of
// the launch (profiles/r02_train_pmc_sq.json), nothing overlapped.  Here a PAIR of waves (w, w + 4) shares the 16 queries of wave w:
//   * phase 1: each takes HALF the key tiles for S^T / dP^T (NKB instead of 2 NKB tiles: half the MFMAs, half the registers); the row
//     maximum, the normaliser and delta = sum_k p dp go through three small LDS arrays (two exchanges, each behind a barrier);
//   * dQ^T: each computes two of the four 16-wide d-blocks over ALL keys (dS rows come from LDS anyway);
//   * dK^T / dV^T: the key tiles are dealt over eight waves.
// Per wave and block 74 MFMAs instead of 148, and while one wave of a SIMD sits in its softmax or waits for LDS the other issues MFMAs.
// Same LDS images and swizzles, same arithmetic per element; the sums over keys are split in two halves (different fp32 summation
// order: results agree with the four-wave kernel to rounding, not bitwise).
template <int NKB, bool BIAS, bool POLICY>
__global__ __launch_bounds__(512, 2) void attention_bwd8_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dO,
                                                                const float* __restrict__ size, const float* __restrict__ dcls,
                                                                uint16_t* __restrict__ dqkv, float* __restrict__ dpol_part, int N, int H) {
  constexpr int NP = NKB * 32;       // padded key count
  constexpr int NT = NKB * 2;        // 16-key tiles
  constexpr int KT_W = (NT + 7) / 8; // key tiles a wave accumulates dK / dV for
  __shared__ __attribute__((aligned(16))) unsigned char sK[NP * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sV[NP * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sQ[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sDO[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sP[64 * 512];
  __shared__ __attribute__((aligned(16))) unsigned char sDS[64 * 512];
  __shared__ __attribute__((aligned(16))) float sLB[NP];     // log2(size[key]) (ToMe / key masks), 0 without sizes
  __shared__ __attribute__((aligned(16))) float sDC[NP];     // d cls_attn[key] / H (EViT), 0 otherwise
  __shared__ float sMax[2][64], sSum[2][64], sSdp[2][64];    // per key half and query of the block: row max, sum e, sum e dp
  __shared__ __attribute__((aligned(16))) float sPol[POLICY ? 16 * NP : 1];      // [query block (<= 4)][query group][key]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qg = wave & 3, half = wave >> 2;            // query group of the block (16 queries), key half
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int ldq = 3 * H * 64, ldo = H * 64;
  const uint16_t* base = qkv + (size_t)b * N * ldq;
  const uint16_t* dobase = dO + (size_t)b * N * ldo + h * 64;
  uint16_t* dbase = dqkv + (size_t)b * N * ldq;
  const int qcol = h * 64, kcol = H * 64 + h * 64, vcol = 2 * H * 64 + h * 64;

  // ---- stage K and V (keys >= N: zero rows); all loads first, then the LDS writes
  {
    constexpr int NIT = (NKB + 1) / 2;
    uint4 kreg[NIT], vreg[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + 512 * it, key = c >> 3, ch = c & 7;
      kreg[it] = *reinterpret_cast<const uint4*>(base + (size_t)min(key, N - 1) * ldq + kcol + ch * 8);
      vreg[it] = *reinterpret_cast<const uint4*>(base + (size_t)min(key, N - 1) * ldq + vcol + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + 512 * it, key = c >> 3, ch = c & 7;
      if (key < NP) {
        if (key >= N) {
          kreg[it] = make_uint4(0u, 0u, 0u, 0u);
          vreg[it] = make_uint4(0u, 0u, 0u, 0u);
        }
        *reinterpret_cast<uint4*>(sK + qswz(key, ch)) = kreg[it];
        *reinterpret_cast<uint4*>(sV + qswz(key, ch)) = vreg[it];
      }
    }
    for (int key = tid; key < NP; key += 512) {
      if (POLICY) {
        sLB[key] = key < N ? size[(size_t)b * N + key] : 0.f;
        sDC[key] = 0.f;
      } else {
        sLB[key] = (BIAS && key < N) ? __builtin_amdgcn_logf(size[(size_t)b * N + key]) : 0.f;     // v_log_f32 = log2
        sDC[key] = (dcls != nullptr && key < N) ? dcls[(size_t)b * N + key] / (float)H : 0.f;
      }
    }
  }

  const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
  const float c_exp = 0.125f * 1.44269504088896340736f;    // dh^-0.5 * log2(e)
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 dk[KT_W][4], dv[KT_W][4];
#pragma unroll
  for (int t = 0; t < KT_W; ++t)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      dk[t][d] = z4;
      dv[t][d] = z4;
    }

  const int nqb = (N + 63) >> 6;
  // Q and dO rows of a query block travel global -> registers -> LDS; the loads of block qb + 1 fly under block qb's phases
  uint4 qreg, oreg;
  auto load_block = [&](int qb) __attribute__((always_inline)) {
    const int r = tid >> 3, ch = tid & 7;
    const int i = qb * 64 + r;
    qreg = *reinterpret_cast<const uint4*>(base + (size_t)min(i, N - 1) * ldq + qcol + ch * 8);
    oreg = *reinterpret_cast<const uint4*>(dobase + (size_t)min(i, N - 1) * ldo + ch * 8);
  };
  load_block(0);
  for (int qb = 0; qb < nqb; ++qb) {
    // ---- stage this block's Q and dO rows (rows >= N: zero)
    {
      __syncthreads();                // every wave is done with the previous block's Q, dO, P, dS images (and the loads have landed)
      const int r = tid >> 3, ch = tid & 7;
      if (qb * 64 + r >= N) {
        qreg = make_uint4(0u, 0u, 0u, 0u);
        oreg = make_uint4(0u, 0u, 0u, 0u);
      }
      *reinterpret_cast<uint4*>(sQ + qswz(r, ch)) = qreg;
      *reinterpret_cast<uint4*>(sDO + qswz(r, ch)) = oreg;
      if (qb + 1 < nqb) load_block(qb + 1);
      lds_barrier();
    }

    // ---- phase 1: this wave's 16 queries against ITS HALF of the keys
    const int il = 16 * qg + li;                        // query row inside the block (this lane's accumulator column)
    const int iq = qb * 64 + il;                        // global query index
    bf16x8 qf[2], of[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(sQ + qswz(il, 4 * ks + g));
      of[ks] = *reinterpret_cast<const bf16x8*>(sDO + qswz(il, 4 * ks + g));
    }
    f32x4 sacc[NKB], dpacc[NKB];
#pragma unroll
    for (int t = 0; t < NKB; ++t) {
      const int jt = half * NKB + t;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + qswz(16 * jt + li, 4 * ks + g));
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + qswz(16 * jt + li, 4 * ks + g));
        sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], ks == 0 ? z4 : sacc[t], 0, 0, 0);     // S^T[key 4g+r][query li]
        dpacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, of[ks], ks == 0 ? z4 : dpacc[t], 0, 0, 0);   // dP^T
      }
    }
    // softmax over the keys of query li: this wave's half first, the partner's through LDS
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKB; ++t) {
      const int jt = half * NKB + t;
      const float4 lb = *reinterpret_cast<const float4*>(&sLB[16 * jt + 4 * g]);
      const float lbv[4] = {lb.x, lb.y, lb.z, lb.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float tt = POLICY ? sacc[t][r] * c_exp : sacc[t][r] * c_exp + lbv[r];
        if (16 * jt + 4 * g + r >= N) tt = -INFINITY;
        sacc[t][r] = tt;
        mx = fmaxf(mx, tt);
      }
    }
    mx = bq_rows_max(mx);
#ifndef TR_LAB_BWD_NOEXCH
    if (g == 0) sMax[half][il] = mx;
    lds_barrier();
    mx = fmaxf(sMax[0][il], sMax[1][il]);
#endif
    if (mx == -INFINITY) mx = 0.f;                      // every key masked: all weights 0 (as the forward)
    const float padd = POLICY ? 1e-6f / (float)N : 0.f;
    float l = 0.f, sdp = 0.f;
#pragma unroll
    for (int t = 0; t < NKB; ++t) {
      const int jt = half * NKB + t;
      const float4 lb = *reinterpret_cast<const float4*>(&sLB[16 * jt + 4 * g]);
      const float lbv[4] = {lb.x, lb.y, lb.z, lb.w};
      const float4 dc = *reinterpret_cast<const float4*>(&sDC[16 * jt + 4 * g]);
      const float dcv[4] = {dc.x, dc.y, dc.z, dc.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * jt + 4 * g + r;
        const float pe = __builtin_amdgcn_exp2f(sacc[t][r] - mx);
        sacc[t][r] = pe;                                  // POLICY: e (the policy is applied below, e itself is needed for d policy)
        if (POLICY) {
          const float pi = key == iq ? 1.0f : lbv[r];
          l += pe * pi;
          sdp += key < N ? (pe * pi + padd) * dpacc[t][r] : 0.f;       // times inv below: sum_k p_k dp_k
        } else {
          if (iq == 0) dpacc[t][r] += dcv[r];             // EViT: d cls_attn reaches the CLS query's row (key 0 carries 0)
          l += pe;
          sdp += pe * dpacc[t][r];
        }
      }
    }
    l = bq_rows_sum(l);
    sdp = bq_rows_sum(sdp);
#ifndef TR_LAB_BWD_NOEXCH
    if (g == 0) { sSum[half][il] = l; sSdp[half][il] = sdp; }
    lds_barrier();
    l = sSum[0][il] + sSum[1][il];
    const float inv = POLICY ? 1.0f / (l + 1e-6f) : (l > 0.f ? 1.0f / l : 0.f);
    const float dl = (sSdp[0][il] + sSdp[1][il]) * inv;
#else
    const float inv = POLICY ? 1.0f / (l + 1e-6f) : (l > 0.f ? 1.0f / l : 0.f);
    const float dl = sdp * inv;
#endif
#pragma unroll
    for (int t = 0; t < NKB; ++t) {
      const int jt = half * NKB + t;
      const float4 lb = *reinterpret_cast<const float4*>(&sLB[16 * jt + 4 * g]);
      const float lbv[4] = {lb.x, lb.y, lb.z, lb.w};
      float pv[4], dsv[4], dpv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * jt + 4 * g + r;
        const float w = dpacc[t][r] - dl;
        if (POLICY) {
          const float pi = key == iq ? 1.0f : lbv[r];
          const float en = sacc[t][r] * inv;              // e / (sum a + eps)
          pv[r] = key < N ? en * pi + padd * inv : 0.f;
          dsv[r] = w * en * pi * 0.125f;
          dpv[r] = row16_sum((key != iq && key < N && iq < N) ? w * en : 0.f);      // summed over this wave's 16 query columns (lanes li)
        } else {
          pv[r] = sacc[t][r] * inv;
          dsv[r] = pv[r] * w * 0.125f;
        }
      }
      if (POLICY && li == 0)       // this (query block, query group)'s own row; the two key halves fill disjoint columns
        *reinterpret_cast<float4*>(&sPol[(qb * 4 + qg) * NP + 16 * jt + 4 * g]) = make_float4(dpv[0], dpv[1], dpv[2], dpv[3]);
      uint2 pp, ds;
      pp.x = pack_bf16x2(pv[0], pv[1]);
      pp.y = pack_bf16x2(pv[2], pv[3]);
      ds.x = pack_bf16x2(dsv[0], dsv[1]);
      ds.y = pack_bf16x2(dsv[2], dsv[3]);
      // keys 16jt + 4g .. +3 of row il.  Rows 8..15 of a 16-row group keep the two 8-byte halves of a chunk swapped: with the half fixed,
      // the 16 lanes of a write group cover 8 of the 16 (chunk mod 8, half) positions twice -- a 2-way conflict on every P / dS write
      // (10 % of the kernel's LDS cycles in r03_a); the readers below undo the swap (tools/lds_sim.py: 8 -> 4 cycles, reads unchanged)
      const int off = pswz(il, 2 * jt + (g >> 1)) + 8 * ((g & 1) ^ ((li >> 3) & 1));
      *reinterpret_cast<uint2*>(sP + off) = pp;
      *reinterpret_cast<uint2*>(sDS + off) = ds;
    }
    lds_barrier();                    // P and dS rows of all eight waves are in LDS (LDS-only wait: the next block's loads stay in flight)
    // ---- dQ^T[d][query] = sum_key K[key][d] dS[query][key]: this wave's two d-blocks (2 half, 2 half + 1) over ALL keys
    {
      f32x4 dq[2];
#pragma unroll
      for (int ks = 0; ks < NKB; ++ks) {
        bf16x8 dsf = *reinterpret_cast<const bf16x8*>(sDS + pswz(il, 4 * ks + g));          // B[k = key][col = query]
        if ((li >> 3) & 1) dsf = __builtin_shufflevector(dsf, dsf, 4, 5, 6, 7, 0, 1, 2, 3);  // this row's halves are stored swapped
        const int r0 = 32 * ks + 8 * g + q4;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
          const int ch = 2 * (2 * half + dd) + (p4 >> 1);
          const bf16x8 ktf = lds_tr_pair(sK + qswz(r0, ch) + 8 * (p4 & 1), sK + qswz(r0 + 4, ch) + 8 * (p4 & 1));   // A[row = d][k = key]
          dq[dd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, ks == 0 ? z4 : dq[dd], 0, 0, 0);
        }
      }
      if (iq < N) {
        uint16_t* qrow = dbase + (size_t)iq * ldq + qcol + 4 * g + 32 * half;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
          uint2 pk;
          pk.x = pack_bf16x2(dq[dd][0], dq[dd][1]);
          pk.y = pack_bf16x2(dq[dd][2], dq[dd][3]);
          *reinterpret_cast<uint2*>(qrow + 16 * dd) = pk;
        }
      }
    }
    // ---- phase 2: dK^T[d][key] += sum_query Q[query][d] dS[query][key], dV^T[d][key] += sum_query dO[query][d] P[query][key]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = 32 * ks + 8 * g + q4;
      bf16x8 qt[4], ot[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int ch = 2 * d + (p4 >> 1);
        qt[d] = lds_tr_pair(sQ + qswz(r0, ch) + 8 * (p4 & 1), sQ + qswz(r0 + 4, ch) + 8 * (p4 & 1));       // A[row = d][k = query]
        ot[d] = lds_tr_pair(sDO + qswz(r0, ch) + 8 * (p4 & 1), sDO + qswz(r0 + 4, ch) + 8 * (p4 & 1));
      }
#pragma unroll
      for (int t = 0; t < KT_W; ++t) {
        const int jt = wave + 8 * t;
        if (jt < NT) {                  // wave-uniform
          const int ch = 2 * jt + (p4 >> 1);
          const int hb = 8 * ((p4 & 1) ^ (g & 1));      // rows r0 and r0 + 4 lie in the same half of their 16-row group: bit 3 = g & 1
          const bf16x8 dsf = lds_tr_pair(sDS + pswz(r0, ch) + hb, sDS + pswz(r0 + 4, ch) + hb);   // B[k = query][col = key]
          const bf16x8 pf = lds_tr_pair(sP + pswz(r0, ch) + hb, sP + pswz(r0 + 4, ch) + hb);
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            dk[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt[d], dsf, dk[t][d], 0, 0, 0);
            dv[t][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot[d], pf, dv[t][d], 0, 0, 0);
          }
        }
      }
    }
  }
  if (POLICY) {
    // the rows of every (query block, query group), added in that order
    __syncthreads();
    for (int key = tid; key < N; key += 512) {
      float a = 0.f;
      for (int rw = 0; rw < 4 * nqb; ++rw) a += sPol[rw * NP + key];
      dpol_part[((size_t)b * H + h) * N + key] = a;
    }
  }
  // ---- dK, dV rows: accumulator (t, d): rows d-index 16d + 4g + r, column key 16*(wave + 8t) + li
#pragma unroll
  for (int t = 0; t < KT_W; ++t) {
    const int key = 16 * (wave + 8 * t) + li;
    if (wave + 8 * t < NT && key < N) {
      uint16_t* krow = dbase + (size_t)key * ldq + kcol + 4 * g;
      uint16_t* vrow = dbase + (size_t)key * ldq + vcol + 4 * g;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint2 pk;
        pk.x = pack_bf16x2(dk[t][d][0], dk[t][d][1]);
        pk.y = pack_bf16x2(dk[t][d][2], dk[t][d][3]);
        *reinterpret_cast<uint2*>(krow + 16 * d) = pk;
        pk.x = pack_bf16x2(dv[t][d][0], dv[t][d][1]);
        pk.y = pack_bf16x2(dv[t][d][2], dv[t][d][3]);
        *reinterpret_cast<uint2*>(vrow + 16 * d) = pk;
      }
    }
  }
}

template <int NKB>
int launch_bwd(const uint16_t* qkv, const uint16_t* dO, const float* size, const float* dcls, uint16_t* dqkv, float* dpol_part, int B, int N,
               int H, hipStream_t st) {
  static const bool use4 = [] { const char* e = getenv("TR_ATT_BWD4"); return e && atoi(e) != 0; }();   // lab: the four-wave kernel
  if (!use4) {
    if (dpol_part != nullptr)
      hipLaunchKernelGGL((attention_bwd8_kernel<NKB, false, true>), dim3(B * H), dim3(512), 0, st, qkv, dO, size, dcls, dqkv, dpol_part, N, H);
    else if (size != nullptr)
      hipLaunchKernelGGL((attention_bwd8_kernel<NKB, true, false>), dim3(B * H), dim3(512), 0, st, qkv, dO, size, dcls, dqkv, dpol_part, N, H);
    else
      hipLaunchKernelGGL((attention_bwd8_kernel<NKB, false, false>), dim3(B * H), dim3(512), 0, st, qkv, dO, size, dcls, dqkv, dpol_part, N, H);
    return 0;
  }
  if (dpol_part != nullptr)
    hipLaunchKernelGGL((attention_bwd_kernel<NKB, false, true>), dim3(B * H), dim3(256), 0, st, qkv, dO, size, dcls, dqkv, dpol_part, N, H);
  else if (size != nullptr)
    hipLaunchKernelGGL((attention_bwd_kernel<NKB, true, false>), dim3(B * H), dim3(256), 0, st, qkv, dO, size, dcls, dqkv, dpol_part, N, H);
  else
    hipLaunchKernelGGL((attention_bwd_kernel<NKB, false, false>), dim3(B * H), dim3(256), 0, st, qkv, dO, size, dcls, dqkv, dpol_part, N, H);
  return 0;
}

int dispatch_bwd(const uint16_t* qkv, const uint16_t* dO, const float* size, const float* dcls, uint16_t* dqkv, float* dpol_part, int B, int N,
                 int H, hipStream_t st) {
  switch ((N + 31) / 32) {
    case 1: return launch_bwd<1>(qkv, dO, size, dcls, dqkv, dpol_part, B, N, H, st);
    case 2: return launch_bwd<2>(qkv, dO, size, dcls, dqkv, dpol_part, B, N, H, st);
    case 3: return launch_bwd<3>(qkv, dO, size, dcls, dqkv, dpol_part, B, N, H, st);
    case 4: return launch_bwd<4>(qkv, dO, size, dcls, dqkv, dpol_part, B, N, H, st);
    case 5: return launch_bwd<5>(qkv, dO, size, dcls, dqkv, dpol_part, B, N, H, st);
    case 6: return launch_bwd<6>(qkv, dO, size, dcls, dqkv, dpol_part, B, N, H, st);
    default: return launch_bwd<7>(qkv, dO, size, dcls, dqkv, dpol_part, B, N, H, st);
  }
}

}  // namespace

// d qkv [B*N, 3*H*64] (bf16) from d out [B*N, H*64] (bf16) and the forward's qkv.  size (nullable) fp32 [B,N]: the key bias the
// forward used (log size, or a 1/0 key mask).  dcls (nullable) fp32 [B,N]: gradient wrt the head-MEAN of the CLS query's softmax
// row (evit.py:117-120; entry 0 = the CLS key must be 0) -- added / H to every head's dP row 0.  N <= 224.
extern "C" int tr_attention_bwd_bf16(const uint16_t* qkv, const uint16_t* dout, const float* size, const float* dcls, uint16_t* dqkv,
                                     int B, int N, int H, tr_stream_t s) {
  TR_REQUIRE(qkv && dout && dqkv, TR_ERR_NULL, "tr_attention_bwd_bf16: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 1, TR_ERR_SHAPE, "tr_attention_bwd_bf16: bad shape B=%d N=%d H=%d", B, N, H);
  TR_REQUIRE(N <= 224, TR_ERR_SHAPE, "tr_attention_bwd_bf16: N=%d > 224 (training at 384^2 inputs is not built)", N);
  TR_REQUIRE(tr_aligned16(qkv) && tr_aligned16(dout) && tr_aligned16(dqkv), TR_ERR_ALIGN, "tr_attention_bwd_bf16: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  tr_prof_note("attention_bwd_kernel", 10.0 * B * H * (double)N * N * 64, 2.0 * B * N * 8.0 * H * 64);
  dispatch_bwd(qkv, dout, size, dcls, dqkv, nullptr, B, N, H, st);
  TR_CHECK_LAUNCH("tr_attention_bwd_bf16");
  return TR_OK;
}

// DyViT training: backward of tr_attention_policy_bf16.  policy fp32 [B,N] of 1/0 (the forward's); dpol_part fp32 [B,H,N]: per head,
// the gradient wrt policy[key] summed over the queries (sum the heads with tr_head_sum).
extern "C" int tr_attention_policy_bwd_bf16(const uint16_t* qkv, const uint16_t* dout, const float* policy, uint16_t* dqkv, float* dpol_part,
                                            int B, int N, int H, tr_stream_t s) {
  TR_REQUIRE(qkv && dout && policy && dqkv && dpol_part, TR_ERR_NULL, "tr_attention_policy_bwd_bf16: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 1 && N <= 224, TR_ERR_SHAPE, "tr_attention_policy_bwd_bf16: need 1 <= N <= 224 (N=%d)", N);
  TR_REQUIRE(tr_aligned16(qkv) && tr_aligned16(dout) && tr_aligned16(dqkv), TR_ERR_ALIGN, "tr_attention_policy_bwd_bf16: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  tr_prof_note("attention_bwd_kernel<policy>", 10.0 * B * H * (double)N * N * 64, 2.0 * B * N * 8.0 * H * 64);
  dispatch_bwd(qkv, dout, policy, nullptr, dqkv, dpol_part, B, N, H, st);
  TR_CHECK_LAUNCH("tr_attention_policy_bwd_bf16");
  return TR_OK;
}
