// Backward of multi-head self-attention for LONG sequences (N > 224: the 384 x 384 inputs, 577 tokens) on gfx950 MFMA.
//
// tr_attention_bwd.hip keeps K, V of a head in LDS and a whole score row in registers; neither fits beyond 224 tokens.  Here the
// keys are walked in blocks of 64 (flash-attention style) and the work is cut into three deterministic launches per (image, head):
//   1. stats  (one workgroup per 64-query block): online max / sum over all keys -> LSE[q] = log2 sum_k 2^t[q,k], and
//             delta[q] = sum_k p[q,k] dP[q,k] accumulated with the same running maximum (no attention OUTPUT needed: ATS keeps only
//             the sampled rows of it)
//   2. dQ     (one workgroup per 64-query block): p = 2^(t - LSE), dS = p (dP - delta) dh^-0.5, dQ = dS K over the key blocks
//   3. dK, dV (one workgroup per 64-key block):  the same p, dS recomputed over the query blocks, dV = P^T dO, dK = dS^T Q
// 9 matrix products of N x N x 64 per head against the 5 of the short kernel: the price of no float atomics (dQ is not
// accumulated across workgroups) and of not saving anything N x N or per-row in the forward.  Same arithmetic as the short kernel
// otherwise (bf16 operands, fp32 accumulate and softmax, P and dS rounded to bf16 for the MFMA products; log-size / key-mask bias
// of ToMe / ATS / Heuristic; EViT's d cls_attn added to query 0's dP).  t = s * dh^-0.5 * log2 e + log2 size[key].
//
// POLICY (DyViT training at 384^2: Policy_Attention.softmax_with_policy dyvit.py:39-51, the backward of tr_attention_policy_bf16; same
// expressions as the short kernel's POLICY variant, tr_attention_bwd.hip): `size` carries the keep policy [B,N] of 1/0;
//   e = 2^(t - max), a = e pi (pi[q][k] = policy[k], 1 on the diagonal), L = sum_k a_k + eps, p = (a + eps/N) / L, eps = 1e-6;
//   delta = sum_k p_k dP_k;  dS = (dP - delta) a / L;  d policy[k] += (dP - delta) e / L over the queries q != k (per head: dpol_part).
//   stats carries three rows per query: max + log2 L (so that 2^(t - .) = e / L), delta, 1 / L.
#include "tr_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ bf16x8 lds_tr_pair(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
  const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}

// [rows][64 bf16] image, 128-B rows: conflict-free for 16-byte row reads AND transposed reads (tools/lds_sim.py; tr_attention_bwd.hip)
__device__ __forceinline__ int qswz(int row, int ch) { return row * 128 + ((ch ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2))) << 4); }

// The P / dS images ([64 queries][64 keys]) are WRITTEN 8 bytes per lane by 16 lanes that hold the same keys of 16 consecutive queries;
// under qswz those 16 rows share 4 chunk positions and one 8-byte half: a 4-way conflict (16 instead of 4 cycles per write).  This
// swizzle spreads the 16 rows over all 16 (chunk, half) positions of a 128-byte row: chunk ^= row bits {0, 1, 3}, the two halves of a
// chunk swapped on rows with bit 2 set; 16-byte row reads and the transposed 8-byte reads stay conflict-free (search + check:
// tools/lds_sim.py).  A 16-byte row read of a swapped row swaps its register halves back.
__device__ __forceinline__ int dswz(int row, int ch) { return row * 128 + ((ch ^ ((row & 3) | (((row >> 3) & 1) << 2))) << 4); }
__device__ __forceinline__ int dswz8(int row, int ch, int half) { return dswz(row, ch) + 8 * (half ^ ((row >> 2) & 1)); }
__device__ __forceinline__ bf16x8 dswz_row_read(const unsigned char* img, int row, int ch) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(img + dswz(row, ch));
  return ((row >> 2) & 1) ? __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3) : v;
}

constexpr int LB = 64;                                        // keys / queries per block
constexpr float C_EXP = 0.125f * 1.44269504088896340736f;     // dh^-0.5 * log2(e)

struct HeadPtrs {
  const uint16_t* base;      // qkv rows of the image
  const uint16_t* dobase;    // d out rows of the image, this head's columns
  int ldq, ldo, qcol, kcol, vcol;
};
__device__ __forceinline__ HeadPtrs head_ptrs(const uint16_t* qkv, const uint16_t* dO, int b, int h, int N, int H) {
  HeadPtrs p;
  p.ldq = 3 * H * 64;
  p.ldo = H * 64;
  p.base = qkv + (size_t)b * N * p.ldq;
  p.dobase = dO + (size_t)b * N * p.ldo + h * 64;
  p.qcol = h * 64;
  p.kcol = H * 64 + h * 64;
  p.vcol = 2 * H * 64 + h * 64;
  return p;
}

// stage 64 rows x 64 columns (bf16) starting at row r0 of a row-major tensor into a qswz image; rows >= N are zero
__device__ __forceinline__ void load_rows(const uint16_t* src, size_t ld, int col, int r0, int N, int tid, uint4 (&reg)[2]) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c = tid + 256 * it, r = c >> 3, ch = c & 7;
    reg[it] = *reinterpret_cast<const uint4*>(src + (size_t)min(r0 + r, N - 1) * ld + col + ch * 8);
  }
}
__device__ __forceinline__ void put_rows(unsigned char* img, int r0, int N, int tid, uint4 (&reg)[2]) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c = tid + 256 * it, r = c >> 3, ch = c & 7;
    if (r0 + r >= N) reg[it] = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(img + qswz(r, ch)) = reg[it];
  }
}

// S^T and dP^T of this wave's 16 queries (fragments qf / of) against the 64 keys staged in sK / sV: accumulator (jt, r) = key
// 16 jt + 4 g + r of query li.  t = s * C_EXP + bias, -inf beyond N.
template <bool BIAS>
__device__ __forceinline__ void scores_block(const unsigned char* sK, const unsigned char* sV, const float* sLB, const float* sDC, const bf16x8 (&qf)[2],
                                             const bf16x8 (&of)[2], int k0, int N, bool cls_query, int li, int g, float (&t)[4][4], float (&dp)[4][4]) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, d = s;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + qswz(16 * jt + li, 4 * ks + g));
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + qswz(16 * jt + li, 4 * ks + g));
      s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, of[ks], d, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kl = 16 * jt + 4 * g + r;
      float v = s[r] * C_EXP + (BIAS ? sLB[kl] : 0.f);
      if (k0 + kl >= N) v = -INFINITY;
      t[jt][r] = v;
      dp[jt][r] = d[r] + (cls_query ? sDC[kl] : 0.f);       // EViT: d cls_attn reaches the CLS query's row
    }
  }
}

__device__ __forceinline__ void stage_key_meta(float* sLB, float* sDC, const float* size, const float* dcls, int b, int k0, int N, int H, int tid,
                                               bool bias, bool policy = false) {
  if (tid < LB) {
    const int key = k0 + tid;
    if (policy) sLB[tid] = key < N ? size[(size_t)b * N + key] : 0.f;                             // the keep policy itself
    else sLB[tid] = (bias && key < N) ? __builtin_amdgcn_logf(size[(size_t)b * N + key]) : 0.f;      // v_log_f32 = log2; log2(0) = -inf masks the key
    sDC[tid] = (dcls != nullptr && key < N) ? dcls[(size_t)b * N + key] / (float)H : 0.f;
  }
}

// ---- 1. per-query statistics: stats[(b*H + h)*N + q] = LSE (log2 domain), stats[B*H*N + ...] = delta
template <bool BIAS, bool POLICY>
__global__ __launch_bounds__(256) void attn_bwd_stats_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dO,
                                                             const float* __restrict__ size, const float* __restrict__ dcls,
                                                             float* __restrict__ stats, int N, int H, int BH) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[LB * 128], sV[LB * 128];
  __shared__ float sLB[LB], sDC[LB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const HeadPtrs P = head_ptrs(qkv, dO, b, h, N, H);
  const int iq = blockIdx.x * LB + 16 * wave + li;
  bf16x8 qf[2], of[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(P.base + (size_t)min(iq, N - 1) * P.ldq + P.qcol + 32 * ks + 8 * g);
    of[ks] = *reinterpret_cast<const bf16x8*>(P.dobase + (size_t)min(iq, N - 1) * P.ldo + 32 * ks + 8 * g);
  }
  float m = -INFINITY, l = 0.f, ds = 0.f;            // running maximum (common to the four lanes of a query), this lane's partial sums
  float sdp = 0.f;                                   // POLICY: sum_k dP_k (the eps/N term of delta)
  uint4 kreg[2], vreg[2];
  for (int k0 = 0; k0 < N; k0 += LB) {
    load_rows(P.base, P.ldq, P.kcol, k0, N, tid, kreg);
    load_rows(P.base, P.ldq, P.vcol, k0, N, tid, vreg);
    __syncthreads();                                  // the previous block's fragments have been read
    put_rows(sK, k0, N, tid, kreg);
    put_rows(sV, k0, N, tid, vreg);
    stage_key_meta(sLB, sDC, size, dcls, b, k0, N, H, tid, BIAS, POLICY);
    __syncthreads();
    float t[4][4], dp[4][4];
    scores_block<BIAS && !POLICY>(sK, sV, sLB, sDC, qf, of, k0, N, iq == 0, li, g, t, dp);
    float mx = m;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, t[jt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (mx > -INFINITY) {
      const float resc = __builtin_amdgcn_exp2f(m - mx);     // 0 on the first block with a live key (m = -inf)
      l *= resc;
      ds *= resc;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float e = __builtin_amdgcn_exp2f(t[jt][r] - mx);
          if (POLICY) {
            const int kl = 16 * jt + 4 * g + r;
            e *= (k0 + kl == iq) ? 1.0f : sLB[kl];
            sdp += dp[jt][r];                               // keys past N: zero V rows, dP = 0
          }
          l += e;
          ds += e * dp[jt][r];
        }
      m = mx;
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  ds += __shfl_xor(ds, 16, 64);
  ds += __shfl_xor(ds, 32, 64);
  if (POLICY) {
    sdp += __shfl_xor(sdp, 16, 64);
    sdp += __shfl_xor(sdp, 32, 64);
    if (g == 0 && iq < N) {
      const float L = l + 1e-6f, invL = 1.0f / L;
      stats[(size_t)bh * N + iq] = m + __builtin_amdgcn_logf(L);
      stats[(size_t)BH * N + (size_t)bh * N + iq] = (ds + (1e-6f / (float)N) * sdp) * invL;
      stats[(size_t)2 * BH * N + (size_t)bh * N + iq] = invL;
    }
    return;
  }
  if (g == 0 && iq < N) {
    const bool live = l > 0.f;
    stats[(size_t)bh * N + iq] = live ? m + __builtin_amdgcn_logf(l) : 0.f;      // every key masked: p = 2^(-inf - 0) = 0 downstream
    stats[(size_t)BH * N + (size_t)bh * N + iq] = live ? ds / l : 0.f;
  }
}

// p and dS of this wave's queries for the staged key block, written to its own 16 rows of sP / sDS ([64 queries][64 keys], qswz)
// POLICY: sLB = the staged keys' policy, kdiag = the local key index of this query's own key (or -1), padd = eps/N / L (0 for rows past N),
// dpol (dK/dV kernel only) += (dP - delta) e / L off the diagonal
template <bool POLICY = false, bool DPOL = false>
__device__ __forceinline__ void write_p_ds(unsigned char* sP, unsigned char* sDS, const float (&t)[4][4], const float (&dp)[4][4], float lse,
                                           float delta, int il, int g, const float* sLB = nullptr, int kdiag = -1, float padd = 0.f,
                                           int kvalid = LB, float (*dpol)[4] = nullptr) {
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    float pv[4], dsv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pv[r] = __builtin_amdgcn_exp2f(t[jt][r] - lse);
      if (POLICY) {
        const int kl = 16 * jt + 4 * g + r;
        const float w = pv[r] * (dp[jt][r] - delta);             // (dP - delta) e / L
        const bool diag = kl == kdiag;
        if (DPOL) dpol[jt][r] += diag ? 0.f : w;
        const float pi = diag ? 1.0f : sLB[kl];
        dsv[r] = w * pi * 0.125f;
        pv[r] = pv[r] * pi + (kl < kvalid ? padd : 0.f);
        continue;
      }
      dsv[r] = pv[r] * (dp[jt][r] - delta) * 0.125f;
    }
    uint2 pp, dd;
    pp.x = pack_bf16x2(pv[0], pv[1]);
    pp.y = pack_bf16x2(pv[2], pv[3]);
    dd.x = pack_bf16x2(dsv[0], dsv[1]);
    dd.y = pack_bf16x2(dsv[2], dsv[3]);
    const int off = dswz8(il, 2 * jt + (g >> 1), g & 1);        // keys 16 jt + 4 g .. + 3 of row il
    if (sP != nullptr) *reinterpret_cast<uint2*>(sP + off) = pp;
    *reinterpret_cast<uint2*>(sDS + off) = dd;
  }
}

// ---- 2. dQ: one workgroup per (64-query block, image, head)
template <bool BIAS, bool POLICY>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dO,
                                                          const float* __restrict__ size, const float* __restrict__ dcls,
                                                          const float* __restrict__ stats, uint16_t* __restrict__ dqkv, int N, int H, int BH) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[LB * 128], sV[LB * 128], sDS[LB * 128];
  __shared__ float sLB[LB], sDC[LB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4, q4 = li >> 2, p4 = li & 3;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const HeadPtrs P = head_ptrs(qkv, dO, b, h, N, H);
  const int il = 16 * wave + li, iq = blockIdx.x * LB + il;
  bf16x8 qf[2], of[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(P.base + (size_t)min(iq, N - 1) * P.ldq + P.qcol + 32 * ks + 8 * g);
    of[ks] = *reinterpret_cast<const bf16x8*>(P.dobase + (size_t)min(iq, N - 1) * P.ldo + 32 * ks + 8 * g);
  }
  const float lse = stats[(size_t)bh * N + min(iq, N - 1)], delta = stats[(size_t)BH * N + (size_t)bh * N + min(iq, N - 1)];
  f32x4 dq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 kreg[2], vreg[2];
  for (int k0 = 0; k0 < N; k0 += LB) {
    load_rows(P.base, P.ldq, P.kcol, k0, N, tid, kreg);
    load_rows(P.base, P.ldq, P.vcol, k0, N, tid, vreg);
    __syncthreads();
    put_rows(sK, k0, N, tid, kreg);
    put_rows(sV, k0, N, tid, vreg);
    stage_key_meta(sLB, sDC, size, dcls, b, k0, N, H, tid, BIAS, POLICY);
    __syncthreads();
    float t[4][4], dp[4][4];
    scores_block<BIAS && !POLICY>(sK, sV, sLB, sDC, qf, of, k0, N, iq == 0, li, g, t, dp);
    if (POLICY) write_p_ds<true, false>(nullptr, sDS, t, dp, lse, delta, il, g, sLB, iq - k0);
    else write_p_ds(nullptr, sDS, t, dp, lse, delta, il, g);
    // dQ^T[d][query] += sum_key K[key][d] dS[query][key]: own rows only (LDS operations of one wave are ordered)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 dsf = dswz_row_read(sDS, il, 4 * ks + g);          // B[k = key][col = query]
      const int r0 = 32 * ks + 8 * g + q4;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int ch = 2 * d + (p4 >> 1);
        const bf16x8 ktf = lds_tr_pair(sK + qswz(r0, ch) + 8 * (p4 & 1), sK + qswz(r0 + 4, ch) + 8 * (p4 & 1));   // A[row = d][k = key]
        dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, dq[d], 0, 0, 0);
      }
    }
  }
  if (iq < N) {
    uint16_t* qrow = dqkv + (size_t)b * N * P.ldq + (size_t)iq * P.ldq + P.qcol + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      uint2 pk;
      pk.x = pack_bf16x2(dq[d][0], dq[d][1]);
      pk.y = pack_bf16x2(dq[d][2], dq[d][3]);
      *reinterpret_cast<uint2*>(qrow + 16 * d) = pk;
    }
  }
}

// ---- 3. dK, dV: one workgroup per (64-key block, image, head); wave w accumulates the 16 keys 16 w .. 16 w + 15 of the block
template <bool BIAS, bool POLICY>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dO,
                                                           const float* __restrict__ size, const float* __restrict__ dcls,
                                                           const float* __restrict__ stats, uint16_t* __restrict__ dqkv,
                                                           float* __restrict__ dpol_part, int N, int H, int BH) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[LB * 128], sV[LB * 128], sQ[LB * 128], sDO[LB * 128], sP[LB * 128], sDS[LB * 128];
  __shared__ float sLB[LB], sDC[LB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4, q4 = li >> 2, p4 = li & 3;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const HeadPtrs P = head_ptrs(qkv, dO, b, h, N, H);
  const int k0 = blockIdx.x * LB;
  {
    uint4 kreg[2], vreg[2];
    load_rows(P.base, P.ldq, P.kcol, k0, N, tid, kreg);
    load_rows(P.base, P.ldq, P.vcol, k0, N, tid, vreg);
    put_rows(sK, k0, N, tid, kreg);
    put_rows(sV, k0, N, tid, vreg);
    stage_key_meta(sLB, sDC, size, dcls, b, k0, N, H, tid, BIAS, POLICY);
  }
  float dpol[4][4];                                   // POLICY: d policy[key 16 jt + 4 g + r] from this lane's queries
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) dpol[jt][r] = 0.f;
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    dk[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int il = 16 * wave + li;
  uint4 qreg[2], oreg[2];
  for (int q0 = 0; q0 < N; q0 += LB) {
    load_rows(P.base, P.ldq, P.qcol, q0, N, tid, qreg);
    load_rows(P.dobase, P.ldo, 0, q0, N, tid, oreg);
    __syncthreads();                      // every wave is done with the previous block's Q, dO, P, dS images (first pass: K, V staged)
    put_rows(sQ, q0, N, tid, qreg);
    put_rows(sDO, q0, N, tid, oreg);
    __syncthreads();
    // phase 1: this wave's 16 queries against the block's 64 keys
    const int iq = q0 + il;
    bf16x8 qf[2], of[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(sQ + qswz(il, 4 * ks + g));
      of[ks] = *reinterpret_cast<const bf16x8*>(sDO + qswz(il, 4 * ks + g));
    }
    float t[4][4], dp[4][4];
    scores_block<BIAS && !POLICY>(sK, sV, sLB, sDC, qf, of, k0, N, iq == 0, li, g, t, dp);
    // rows beyond N are zero rows of Q and dO: their p must not reach dV / dK -> lse = +inf gives p = 0, dS = 0
    const float lse = iq < N ? stats[(size_t)bh * N + iq] : INFINITY;
    const float delta = iq < N ? stats[(size_t)BH * N + (size_t)bh * N + iq] : 0.f;
    if (POLICY) {
      const float padd = iq < N ? (1e-6f / (float)N) * stats[(size_t)2 * BH * N + (size_t)bh * N + iq] : 0.f;
      write_p_ds<true, true>(sP, sDS, t, dp, lse, delta, il, g, sLB, iq - k0, padd, N - k0, dpol);
    } else {
      write_p_ds(sP, sDS, t, dp, lse, delta, il, g);
    }
    __syncthreads();                      // P and dS rows of all four waves are in LDS
    // phase 2: dK^T[d][key] += sum_query Q[query][d] dS[query][key], dV^T[d][key] += sum_query dO[query][d] P[query][key]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = 32 * ks + 8 * g + q4;
      const int chk = 2 * wave + (p4 >> 1);
      const bf16x8 dsf = lds_tr_pair(sDS + dswz8(r0, chk, p4 & 1), sDS + dswz8(r0 + 4, chk, p4 & 1));   // B[k = query][col = key]
      const bf16x8 pf = lds_tr_pair(sP + dswz8(r0, chk, p4 & 1), sP + dswz8(r0 + 4, chk, p4 & 1));
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int ch = 2 * d + (p4 >> 1);
        const bf16x8 qt = lds_tr_pair(sQ + qswz(r0, ch) + 8 * (p4 & 1), sQ + qswz(r0 + 4, ch) + 8 * (p4 & 1));       // A[row = d][k = query]
        const bf16x8 ot = lds_tr_pair(sDO + qswz(r0, ch) + 8 * (p4 & 1), sDO + qswz(r0 + 4, ch) + 8 * (p4 & 1));
        dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf, dk[d], 0, 0, 0);
        dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot, pf, dv[d], 0, 0, 0);
      }
    }
  }
  if (POLICY) {
    // d policy of the block's 64 keys: over the 16 queries of a lane row (DPP / shuffles), then over the four waves through the LDS
    __syncthreads();                      // sP is free
    float* red = reinterpret_cast<float*>(sP);                      // [wave][64 keys]
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = dpol[jt][r];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (li == 0) red[wave * LB + 16 * jt + 4 * g + r] = v;
      }
    __syncthreads();
    if (tid < LB && k0 + tid < N)
      dpol_part[(size_t)bh * N + k0 + tid] = (red[tid] + red[LB + tid]) + (red[2 * LB + tid] + red[3 * LB + tid]);
  }
  // accumulator d: rows d-index 16 d + 4 g + r, column key k0 + 16 wave + li
  const int key = k0 + 16 * wave + li;
  if (key < N) {
    uint16_t* krow = dqkv + (size_t)b * N * P.ldq + (size_t)key * P.ldq + P.kcol + 4 * g;
    uint16_t* vrow = dqkv + (size_t)b * N * P.ldq + (size_t)key * P.ldq + P.vcol + 4 * g;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      uint2 pk;
      pk.x = pack_bf16x2(dk[d][0], dk[d][1]);
      pk.y = pack_bf16x2(dk[d][2], dk[d][3]);
      *reinterpret_cast<uint2*>(krow + 16 * d) = pk;
      pk.x = pack_bf16x2(dv[d][0], dv[d][1]);
      pk.y = pack_bf16x2(dv[d][2], dv[d][3]);
      *reinterpret_cast<uint2*>(vrow + 16 * d) = pk;
    }
  }
}

}  // namespace

extern "C" size_t tr_attention_bwd_long_workspace_floats(int B, int N, int H) { return (size_t)3 * B * H * N; }

// Same contract as tr_attention_bwd_bf16 for any N (used beyond 224 tokens); ws: tr_attention_bwd_long_workspace_floats(B,N,H) floats.
extern "C" int tr_attention_bwd_long_bf16(const uint16_t* qkv, const uint16_t* dout, const float* size, const float* dcls, uint16_t* dqkv,
                                          float* ws, size_t ws_floats, int B, int N, int H, tr_stream_t s) {
  TR_REQUIRE(qkv && dout && dqkv && ws, TR_ERR_NULL, "tr_attention_bwd_long_bf16: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 1, TR_ERR_SHAPE, "tr_attention_bwd_long_bf16: bad shape B=%d N=%d H=%d", B, N, H);
  TR_REQUIRE(ws_floats >= tr_attention_bwd_long_workspace_floats(B, N, H), TR_ERR_SHAPE, "tr_attention_bwd_long_bf16: workspace too small");
  TR_REQUIRE(tr_aligned16(qkv) && tr_aligned16(dout) && tr_aligned16(dqkv), TR_ERR_ALIGN, "tr_attention_bwd_long_bf16: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  const int nb = (N + LB - 1) / LB, BH = B * H;
  const dim3 grid(nb, BH);
  tr_prof_note("attention_bwd_long", 18.0 * BH * (double)N * N * 64, 2.0 * B * N * 8.0 * H * 64);
  if (size != nullptr) {
    hipLaunchKernelGGL((attn_bwd_stats_kernel<true, false>), grid, dim3(256), 0, st, qkv, dout, size, dcls, ws, N, H, BH);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<true, false>), grid, dim3(256), 0, st, qkv, dout, size, dcls, ws, dqkv, N, H, BH);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, false>), grid, dim3(256), 0, st, qkv, dout, size, dcls, ws, dqkv, static_cast<float*>(nullptr), N, H, BH);
  } else {
    hipLaunchKernelGGL((attn_bwd_stats_kernel<false, false>), grid, dim3(256), 0, st, qkv, dout, size, dcls, ws, N, H, BH);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<false, false>), grid, dim3(256), 0, st, qkv, dout, size, dcls, ws, dqkv, N, H, BH);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, false>), grid, dim3(256), 0, st, qkv, dout, size, dcls, ws, dqkv, static_cast<float*>(nullptr), N, H, BH);
  }
  TR_CHECK_LAUNCH("tr_attention_bwd_long_bf16");
  return TR_OK;
}

// DyViT training beyond 224 tokens: the backward of tr_attention_policy_bf16 (same contract as tr_attention_policy_bwd_bf16, which
// forwards here for N > 224).  ws: tr_attention_bwd_long_workspace_floats(B,N,H) floats.
extern "C" int tr_attention_policy_bwd_long_bf16(const uint16_t* qkv, const uint16_t* dout, const float* policy, uint16_t* dqkv, float* dpol_part,
                                                 float* ws, size_t ws_floats, int B, int N, int H, tr_stream_t s) {
  TR_REQUIRE(qkv && dout && policy && dqkv && dpol_part && ws, TR_ERR_NULL, "tr_attention_policy_bwd_long_bf16: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 1, TR_ERR_SHAPE, "tr_attention_policy_bwd_long_bf16: bad shape B=%d N=%d H=%d", B, N, H);
  TR_REQUIRE(ws_floats >= tr_attention_bwd_long_workspace_floats(B, N, H), TR_ERR_SHAPE, "tr_attention_policy_bwd_long_bf16: workspace too small");
  TR_REQUIRE(tr_aligned16(qkv) && tr_aligned16(dout) && tr_aligned16(dqkv), TR_ERR_ALIGN,
             "tr_attention_policy_bwd_long_bf16: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  const int nb = (N + LB - 1) / LB, BH = B * H;
  const dim3 grid(nb, BH);
  const float* none = nullptr;
  tr_prof_note("attention_bwd_long<policy>", 18.0 * BH * (double)N * N * 64, 2.0 * B * N * 8.0 * H * 64);
  hipLaunchKernelGGL((attn_bwd_stats_kernel<false, true>), grid, dim3(256), 0, st, qkv, dout, policy, none, ws, N, H, BH);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<false, true>), grid, dim3(256), 0, st, qkv, dout, policy, none, ws, dqkv, N, H, BH);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, true>), grid, dim3(256), 0, st, qkv, dout, policy, none, ws, dqkv, dpol_part, N, H, BH);
  TR_CHECK_LAUNCH("tr_attention_policy_bwd_long_bf16");
  return TR_OK;
}
