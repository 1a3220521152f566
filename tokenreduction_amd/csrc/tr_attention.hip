// Multi-head self-attention for ViT sequences (head_dim 64) on gfx950 MFMA: N <= 224 tokens register-resident; longer ones by an
// online-softmax kernel over 128-key chunks (any N), or -- when column sums are wanted -- a two-pass kernel (N <= 608).
//
// Replaces  attn = softmax(q k^T * dh^-0.5); x = attn @ v   (topk.py:44-51 == evit.py:66-73 == deit_viz.py:43-51)
// and emits the CLS query's softmax row per head (attn[:, :, 0, :], topk.py:59) as a side output, so the
// B*H*N*N matrix the reference keeps alive for the Top-K score is never materialised.
//
// One workgroup (4 waves) per (image, head).  K [N][64] is staged once into LDS row-major with the
// 16-byte chunk index XOR-swizzled by (key>>1)&7; V is staged TRANSPOSED (Vt[d][key]) so that both
// MFMA operands are 16-byte LDS reads (tools/lds_sim.py: all reads/writes conflict-free).
// Each wave owns 32-query blocks and keeps the WHOLE score row in registers (no online softmax):
//   S^T[key][q] = K Q^T       v_mfma_f32_32x32x16_bf16, lane = query, 16 keys per lane per 32-key block
//   softmax over keys         lane-local over registers + one exchange with lane^32
//   O^T[d][q]  = Vt P^T       the exponentiated S^T accumulator is fed straight back as the MFMA B operand
//                             ("accumulator tile as the next MFMA's operand", cdna guide section 3); the k-order
//                             permutation this implies (key bits 2<->3 inside each 32-key block) is applied
//                             once, when V is transposed into LDS.
// Arithmetic: bf16 operands, fp32 accumulate/softmax; P is rounded to bf16 for P.V, the normaliser is the
// fp32 sum of the un-rounded exponentials (same rounding points as oracle precision="bf16").
#include "tr_common.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ int kswz(int key, int chunk) { return key * 128 + ((chunk ^ ((key >> 1) & 7)) << 4); }
__device__ __forceinline__ int swap23(int k) { return (k & ~0xC) | ((k & 4) << 1) | ((k & 8) >> 1); }

// one step of the transpose-reduce over the 32 query lanes of a half-wave: the lane pair (xor BIT) splits the CNT live
// values between them and adds the partner's copy, so 16 -> 8 -> 4 -> 2 -> 1 values per lane in a fixed summation order
template <int CNT, int BIT>
__device__ __forceinline__ void colsum_step(float* v, int ql) {
  const bool hi = (ql & BIT) != 0;
#pragma unroll
  for (int i = 0; i < CNT / 2; ++i) {
    const float send = hi ? v[i] : v[i + CNT / 2];
    const float keep = hi ? v[i + CNT / 2] : v[i];
    v[i] = keep + __shfl_xor(send, BIT, 64);
  }
}

// COLSUM: also emit, per (image, head, wave), the column sums over this wave's queries of the softmax matrix -- the token
// weights K-Medoids needs from the previous block's attention (kmedoids.py:240) -- without materialising B*H*N*N.
// POLICY: DyViT's training-time softmax_with_policy (dyvit.py:39-51): `size` then carries the keep policy [B,N] of 1/0 and
//   attn = (exp(s - max) * pol + eps/N) / (sum_k exp(s - max) * pol + eps),  pol[q][k] = policy[k], 1 on the diagonal.
template <int NKB, bool COLSUM, bool POLICY, bool BIAS>
__global__ __launch_bounds__(256, 2) void attention_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                           float* __restrict__ cls_rows, const float* __restrict__ size,
                                                           float* __restrict__ colsum_part, int N, int H) {
  constexpr int RS = NKB * 64 + 16;  // Vt row stride in bytes: odd multiple of 16 -> conflict-free b128 column reads
  __shared__ __attribute__((aligned(16))) unsigned char sK[NKB * 32 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sVt[64 * RS];
  __shared__ __attribute__((aligned(16))) float sLB[NKB * 32];   // log2(size[key]): ToMe's proportional attention (tome.py:48-49), 0 without sizes

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int ldq = 3 * H * 64;
  const uint16_t* base = qkv + (size_t)b * N * ldq;
  const int qcol = h * 64, kcol = H * 64 + h * 64, vcol = 2 * H * 64 + h * 64;
#ifdef TR_ATT_STAMPS   // diagnostic build (tools/attn_stamps.py): clock64 stamps of the first query block go to cls_rows
  long long stamp[8];
  int nst = 0;
#define TR_STAMP() do { if (nst < 8) stamp[nst++] = clock64(); } while (0)
  TR_STAMP();
#else
#define TR_STAMP() do { } while (0)
#endif

  // ---- stage K (row-major, swizzled) and V transposed; keys >= N are zero rows.  All 2 * NKB 16-byte loads of a thread are
  // issued BEFORE the first LDS write: as rolled loops (load, wait, write, next) the 2 * NKB global round trips ran back to
  // back and were 13.8 k of a workgroup's 33 k cycles at N = 197 (profiles/r01_attention_lab.md).  The registers are free here
  // (nothing of the query loop is live yet).
  uint4 kreg[NKB], vreg[NKB];
#pragma unroll
  for (int it = 0; it < NKB; ++it) {
    const int g = tid + 256 * it;                       // NKB * 32 * 8 chunks = NKB per thread
    const int key = g >> 3, c = g & 7;
    // branch-free (clamped row, zeroed below): a predicated load puts an exec-mask branch and, with it, a wait between the loads
    kreg[it] = *reinterpret_cast<const uint4*>(base + (size_t)min(key, N - 1) * ldq + kcol + c * 8);
  }
  // V: lanes 0-31 take the 32 keys of a block for d-chunk c, lanes 32-63 chunk c+1
#pragma unroll
  for (int it = 0; it < NKB; ++it) {
    const int u = wave + 4 * it;                        // NKB * 4 (block, chunk-pair) units = NKB per wave
    const int kb = u >> 2, c = 2 * (u & 3) + (lane >> 5);
    const int key = kb * 32 + (lane & 31);
    vreg[it] = *reinterpret_cast<const uint4*>(base + (size_t)min(key, N - 1) * ldq + vcol + c * 8);
  }
#pragma unroll
  for (int it = 0; it < NKB; ++it) {
    const int g = tid + 256 * it;
    if ((g >> 3) >= N) kreg[it] = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(sK + kswz(g >> 3, g & 7)) = kreg[it];
  }
#pragma unroll
  for (int it = 0; it < NKB; ++it) {
    const int u = wave + 4 * it;
    const int kb = u >> 2, c = 2 * (u & 3) + (lane >> 5);
    const int kl = lane & 31;
    if (kb * 32 + kl >= N) vreg[it] = make_uint4(0u, 0u, 0u, 0u);
    unsigned short* dst = reinterpret_cast<unsigned short*>(sVt + (8 * c) * RS + (kb * 32 + swap23(kl)) * 2);
    const unsigned int w[4] = {vreg[it].x, vreg[it].y, vreg[it].z, vreg[it].w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned int word = w[e >> 1];
      dst[(e * RS) >> 1] = (unsigned short)((e & 1) ? (word >> 16) : (word & 0xffffu));
    }
  }
  for (int key = tid; key < NKB * 32; key += 256)
    sLB[key] = POLICY ? (key < N ? size[(size_t)b * N + key] : 0.f)
                      : ((size != nullptr && key < N) ? __builtin_amdgcn_logf(size[(size_t)b * N + key]) : 0.f);   // v_log_f32 = log2
  __syncthreads();
  TR_STAMP();

  const int ql = lane & 31, hh = lane >> 5;
  const int nqb = (N + 31) >> 5;
  const float c_exp = 0.125f * 1.44269504088896340736f;  // dh^-0.5 * log2(e), dh = 64
  float colacc[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) colacc[kb] = 0.f;

  for (int qb = wave; qb < nqb; qb += 4) {
    const int q = qb * 32 + ql;
    const uint16_t* qrow = base + (size_t)min(q, N - 1) * ldq + qcol + 8 * hh;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s);
#ifdef TR_ATT_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    TR_STAMP();

    f32x16 sacc[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + kswz(kb * 32 + ql, 2 * s + hh));
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[kb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep K-fragment reads per key block: bounds live registers
    }
    TR_STAMP();
    // ---- softmax over keys (rows of S^T): registers of this lane + the other half-wave
    float mx = -INFINITY;
    float l = 0.f;
    if (!POLICY && !BIAS) {
      // no key bias (DeiT / Top-K / EViT / DyViT ...): the maximum is taken on the raw scores and the scale folded into the
      // exponent's fma, exp2(s * c - max * c), on register PAIRS (v_pk_fma_f32, v_pk_add_f32): 3 VALU instructions per element
      // instead of 5 (scale, max3, sub, exp, add -> max3, half a pk_fma, exp, half a pk_add).  The softmax is the longer half
      // of a query block (stamps: 4.1 k of 8 k cycles), so this is what the kernel's time follows.
      typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (NKB - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (key >= N) sacc[NKB - 1][r] = -INFINITY;
      }
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float nm = -mx * c_exp;
      f32x2 l2 = {0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2 t = {sacc[kb][r], sacc[kb][r + 1]};
          t = t * c_exp + nm;
          f32x2 pp;
          pp[0] = __builtin_amdgcn_exp2f(t[0]);
          pp[1] = __builtin_amdgcn_exp2f(t[1]);
          sacc[kb][r] = pp[0];
          sacc[kb][r + 1] = pp[1];
          l2 += pp;
        }
      l = l2[0] + l2[1];
    } else if (!POLICY) {
      // key bias log2(size[key]) (ToMe's proportional attention, ATS / heuristic masks as log2 0 = -inf): same packed arithmetic,
      // the bias comes from LDS four keys at a time (registers 4g..4g+3 of a block hold keys 8g + 4hh + 0..3)
      typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(&sLB[kb * 32 + 8 * g + 4 * hh]);
          f32x2 t0 = {sacc[kb][4 * g], sacc[kb][4 * g + 1]}, t1 = {sacc[kb][4 * g + 2], sacc[kb][4 * g + 3]};
          t0 = t0 * c_exp + f32x2{b4.x, b4.y};
          t1 = t1 * c_exp + f32x2{b4.z, b4.w};
          float e[4] = {t0[0], t0[1], t1[0], t1[1]};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (kb == NKB - 1 && kb * 32 + 8 * g + 4 * hh + u >= N) e[u] = -INFINITY;
            sacc[kb][4 * g + u] = e[u];
            mx = fmaxf(mx, e[u]);
          }
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float nm = (mx == -INFINITY) ? 0.f : -mx;         // every key masked: all weights exp2(-inf) = 0, as before
      f32x2 l2 = {0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2 t = {sacc[kb][r], sacc[kb][r + 1]};
          t = t + nm;
          f32x2 pp;
          pp[0] = __builtin_amdgcn_exp2f(t[0]);
          pp[1] = __builtin_amdgcn_exp2f(t[1]);
          sacc[kb][r] = pp[0];
          sacc[kb][r + 1] = pp[1];
          l2 += pp;
        }
      l = l2[0] + l2[1];
    } else {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        // logits in the log2 domain: (q.k) * dh^-0.5 * log2(e) + log2(size[key])
        sacc[kb][r] = POLICY ? sacc[kb][r] * c_exp : sacc[kb][r] * c_exp + sLB[key];
        if (kb == NKB - 1 && key >= N) sacc[kb][r] = -INFINITY;
        mx = fmaxf(mx, sacc[kb][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __builtin_amdgcn_exp2f(sacc[kb][r] - mx);
        if (POLICY) {
          const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          p *= (key == q) ? 1.0f : sLB[key];                     // attn_policy = policy + (1 - policy) * eye
        }
        sacc[kb][r] = p;
        l += p;
      }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = POLICY ? 1.0f / (l + 1e-6f) : 1.0f / l;
    if (POLICY) {                                                // (attn + eps/N) / (sum + eps); padded keys stay 0
      const float add = 1e-6f / (float)N;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (kb < NKB - 1 || key < N) sacc[kb][r] += add;
        }
    }

    TR_STAMP();
    // ---- O^T = Vt P^T
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (__bf16)sacc[kb][8 * s + j];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sVt + (db * 32 + ql) * RS + (kb * 32 + 16 * s + 8 * hh) * 2);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }

    TR_STAMP();
#ifdef TR_ATT_NO_STORE
    if (q == 0 && o[0][0] == 123.456f) out[0] = 1;
#else
    if (q < N) {
      uint16_t* orow = out + ((size_t)b * N + q) * (H * 64) + h * 64 + 4 * hh;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 pk;
          pk.x = pack_bf16x2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
          pk.y = pack_bf16x2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(orow + db * 32 + 8 * g) = pk;
        }
    }
#endif
    TR_STAMP();
    if (COLSUM) {
      const float wq = q < N ? inv : 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = sacc[kb][r] * wq;
        colsum_step<16, 1>(v, ql);
        colsum_step<8, 2>(v, ql);
        colsum_step<4, 4>(v, ql);
        colsum_step<2, 8>(v, ql);
        colacc[kb] += v[0] + __shfl_xor(v[0], 16, 64);
      }
    }
#ifndef TR_ATT_STAMPS
    if (cls_rows != nullptr && qb == 0 && ql == 0) {
      float* crow = cls_rows + ((size_t)b * H + h) * N;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key < N) crow[key] = sacc[kb][r] * inv;
        }
    }
#endif
  }
#ifdef TR_ATT_STAMPS
  if (lane == 0 && cls_rows != nullptr) {
    long long* dbg = reinterpret_cast<long long*>(cls_rows) + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int i = 0; i < 8; ++i) dbg[i] = i < nst ? stamp[i] - stamp[0] : -1;
  }
#endif
  if (COLSUM && (ql & 16) == 0) {
    // after the transpose-reduce, lane bits (b0,b1,b2,b3) of ql select register r = 8 b0 + 4 b1 + 2 b2 + b3
    const int r = ((ql & 1) << 3) | ((ql & 2) << 1) | ((ql & 4) >> 1) | ((ql & 8) >> 3);
    float* crow = colsum_part + (((size_t)b * H + h) * 4 + wave) * N;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (key < N) crow[key] = colacc[kb];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// N <= 224, round 3: the same whole-row-in-registers softmax on 16-QUERY blocks (v_mfma_f32_16x16x32_bf16), three workgroups per CU.
//
// Why (profiles/r02_final_pmc_sq.json): attention_kernel above keeps 32 queries x 224 keys = 112 score registers per
// lane, so two waves per SIMD is all the register file takes, two workgroups per CU is all the LDS takes (K + V^T = 58 KB), and a
// wave spends its 2 x 8 k cycles per workgroup in order: matrix pipe busy 15 %, HBM at 3.1 TB/s -- bound by neither.  Here:
//   * S^T block = 16 keys x 16 queries: a lane holds 4 consecutive keys of ONE query per block, 56 registers for 224 keys, so three
//     waves per SIMD fit (168 VGPRs) and the waves of three workgroups in different phases overlap MFMA, softmax VALU and staging;
//   * LDS rows are allocated for ceil(N/16)*16 keys (208 at N = 197): K + V = 52 KB -> three workgroups per CU;
//   * V is staged ROW-major like K (one 16-byte LDS write per 16 bytes loaded; the transposing scatter above took 8 two-byte writes)
//     and read transposed by ds_read_b64_tr_b16.  V image: 16-byte chunk c of key row r sits at chunk c ^ 4 (r>>1 & 1) with its two
//     8-byte halves swapped when (r>>2 & 1): conflict-free for the transposed reads (tools/lds_sim.py);
//   * the P.V operand rows are PERMUTED through the transposed read's four independently addressed column pieces, so that a lane's
//     output registers of a d-block pair are 8 consecutive head dimensions: 16-byte output stores.
// Layouts (cdna guide section 3, 16x16x32): S^T[kb] = K_kb Q^T: lane (q = lane & 15, g = lane >> 4), register r = key kb*16 + 4g + r.
// P^T as the B operand of O^T += V^T P^T over the key pair (2t, 2t+1): k index 8g + j <-> key 32t + 16 (j>>2) + 4g + (j&3), the same
// keys the two transposed reads of V deliver.  O^T[db] register r <-> head dimension 32 (db>>1) + 8g + 4 (db&1) + r.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
__device__ __forceinline__ bf16x8 lds_tr_pair16(const unsigned char* p0, const unsigned char* p1) {
  const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0));
  const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p1));
  const s16x8_t c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}

// reductions over the four lanes (lane >> 4 = 0..3) that share a query: two VALU half / row swaps instead of two LDS round trips
// (v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second; v_permlane16_swap: odd 16-lane rows of the
// first <-> even rows of the second; with both operands the same value, the two results are the value of the two partners)
__device__ __forceinline__ float quad_rows_max(float v) {
  // (the elements of the builtin's result are copied to scalars first: __builtin_bit_cast applied to `a[1]` directly read element 0
  // -- hipcc 7.2 -- and the reduction silently lost the partner's value)
  const auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned a0 = a[0], a1 = a[1];
  v = fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1));
  const auto c = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned c0 = c[0], c1 = c[1];
  return fmaxf(__builtin_bit_cast(float, c0), __builtin_bit_cast(float, c1));
}
__device__ __forceinline__ float quad_rows_sum(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned a0 = a[0], a1 = a[1];
  v = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
  const auto c = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned c0 = c[0], c1 = c[1];
  return __builtin_bit_cast(float, c0) + __builtin_bit_cast(float, c1);
}
#ifndef TR_ATT16_WAVES
#define TR_ATT16_WAVES 4
#endif
#ifndef TR_ATT16_SB
#define TR_ATT16_SB 3
#endif
constexpr int A16_NW = TR_ATT16_WAVES, A16_NT = 64 * A16_NW;     // waves / threads per workgroup
template <int NP, bool COLSUM, bool POLICY, bool BIAS>
__global__ __launch_bounds__(A16_NT, (COLSUM || POLICY) ? 2 : 3) void attention16_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                             float* __restrict__ cls_rows, const float* __restrict__ size,
                                                             float* __restrict__ colsum_part, int N, int H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
  constexpr int NKB = 2 * NP;                       // 16-key blocks
  const int R = ((N + 15) >> 4) << 4;               // key rows held in LDS
  unsigned char* sK = smem16;
  unsigned char* sV = sK + R * 128;
  float* sLB = reinterpret_cast<float*>(sV + R * 128);     // POLICY: keep policy; BIAS: log2(size[key]) (ToMe / key masks)

  const int tid = threadIdx.x, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  // (round 4: rotating the query-block slots per workgroup -- so that the co-resident workgroups' 4-block waves do not meet on one SIMD --
  // measured 40.3 vs 37.9 us at N = 197 on the same box: not kept, profiles/r04_lab.md)
  const int wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int ldq = 3 * H * 64;
  const uint16_t* base = qkv + (size_t)b * N * ldq;
  const int qcol = h * 64, kcol = H * 64 + h * 64, vcol = 2 * H * 64 + h * 64;
  const int nqb = (N + 15) >> 4;

  // ---- stage K and V (all loads issued before the first LDS write); the first query block's fragments are requested alongside
  constexpr int NIT = (NP * 256 + A16_NT - 1) / A16_NT;     // 16-byte chunks per thread and matrix
  uint4 kreg[NIT], vreg[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = tid + A16_NT * it, key = c >> 3, ch = c & 7;
    kreg[it] = *reinterpret_cast<const uint4*>(base + (size_t)min(key, N - 1) * ldq + kcol + ch * 8);
    vreg[it] = *reinterpret_cast<const uint4*>(base + (size_t)min(key, N - 1) * ldq + vcol + ch * 8);
  }
  bf16x8 qf[2];
  {
    const uint16_t* qrow = base + (size_t)min(wave * 16 + li, N - 1) * ldq + qcol + 8 * g;
    qf[0] = *reinterpret_cast<const bf16x8*>(qrow);
    qf[1] = *reinterpret_cast<const bf16x8*>(qrow + 32);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = tid + A16_NT * it, key = c >> 3, ch = c & 7;
    if (key < R) {
      uint4 k4 = kreg[it], v4 = vreg[it];
      if (key >= N) { k4 = make_uint4(0u, 0u, 0u, 0u); v4 = make_uint4(0u, 0u, 0u, 0u); }
      *reinterpret_cast<uint4*>(sK + kswz(key, ch)) = k4;
      if ((key >> 2) & 1) v4 = make_uint4(v4.z, v4.w, v4.x, v4.y);
      *reinterpret_cast<uint4*>(sV + key * 128 + ((ch ^ (((key >> 1) & 1) << 2)) << 4)) = v4;
    }
  }
  if (POLICY || BIAS) {
    for (int key = tid; key < R; key += A16_NT)
      sLB[key] = POLICY ? (key < N ? size[(size_t)b * N + key] : 0.f)
                        : ((size != nullptr && key < N) ? __builtin_amdgcn_logf(size[(size_t)b * N + key]) : 0.f);   // v_log_f32 = log2
  }
  __syncthreads();

  const float c_exp = 0.125f * 1.44269504088896340736f;  // dh^-0.5 * log2(e), dh = 64
  const bool last_empty = R < NKB * 16;                   // the last pair's second 16-key block lies wholly past the keys
  // K fragment of block kb, k-step s: row kb*16 + li, logical chunk 4s + g; (row >> 1) & 7 = (li >> 1) & 7 for every kb
  const unsigned char* kbase = sK + li * 128;
  const int kx0 = ((g ^ ((li >> 1) & 7)) << 4), kx1 = (((4 + g) ^ ((li >> 1) & 7)) << 4);
  // transposed V reads: lane 4q'+p' of a 16-lane group addresses key row 32t + 4g + q' (+16), piece p' of the d-block's four
  const int q4 = li >> 2, p4 = li & 3;
  const unsigned char* vb[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const int c = 4 * (db >> 1) + p4, hb = db & 1;
    vb[db] = sV + (4 * g + q4) * 128 + ((c ^ (((q4 >> 1) & 1) << 2)) << 4) + ((hb ^ (g & 1)) << 3);
  }
  float colacc[4] = {0.f, 0.f, 0.f, 0.f};                // COLSUM: lane li owns key block li

  for (int qb = wave; qb < nqb; qb += A16_NW) {
    const int q = qb * 16 + li;
    // ---- S^T = K Q^T
    f32x4 sacc[NKB];
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};     // C = 0 as an inline constant of a chain's first MFMA (no v_mov per accumulator register)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb < NKB - 1 || !last_empty) {
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kbase + kb * 2048 + kx0);
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kbase + kb * 2048 + kx1);
        sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[0], z4, 0, 0, 0);
        sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[1], sacc[kb], 0, 0, 0);
      } else {
        sacc[kb] = z4;
      }
      if ((kb & TR_ATT16_SB) == TR_ATT16_SB) __builtin_amdgcn_sched_barrier(0);      // K-fragment reads stay with their group of key blocks: bounds the live registers
    }
    // the next block's query fragments fly under this block's softmax and P.V
    {
      const int qn = qb + A16_NW < nqb ? qb + A16_NW : qb;
      const uint16_t* qrow = base + (size_t)min(qn * 16 + li, N - 1) * ldq + qcol + 8 * g;
      qf[0] = *reinterpret_cast<const bf16x8*>(qrow);
      qf[1] = *reinterpret_cast<const bf16x8*>(qrow + 32);
    }
    // ---- softmax over keys: this lane's registers, then the four lanes (g = 0..3) that share the query
    float mx = -INFINITY, l = 0.f;
    if (!POLICY && !BIAS) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (kb >= NKB - 2 && kb * 16 + 4 * g + r >= N) sacc[kb][r] = -INFINITY;
          mx = fmaxf(mx, sacc[kb][r]);
        }
      mx = quad_rows_max(mx);
      const float nm = -mx * c_exp;
      // exp2(s * c - max * c) and the row sum on register PAIRS (v_pk_fma_f32, v_pk_add_f32): 3 VALU instructions per element with the exp
      typedef __attribute__((ext_vector_type(2))) float f32x2;
      f32x2 l2 = {0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          f32x2 t = {sacc[kb][r], sacc[kb][r + 1]};
          t = t * c_exp + nm;
          f32x2 pp;
          pp[0] = __builtin_amdgcn_exp2f(t[0]);
          pp[1] = __builtin_amdgcn_exp2f(t[1]);
          sacc[kb][r] = pp[0];
          sacc[kb][r + 1] = pp[1];
          l2 += pp;
        }
      l = l2[0] + l2[1];
    } else {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!POLICY && (kb < NKB - 1 || !last_empty)) b4 = *reinterpret_cast<const float4*>(&sLB[kb * 16 + 4 * g]);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // logits in the log2 domain: (q.k) * dh^-0.5 * log2(e) [+ log2(size[key])]
          sacc[kb][r] = sacc[kb][r] * c_exp + bb[r];
          if (kb >= NKB - 2 && kb * 16 + 4 * g + r >= N) sacc[kb][r] = -INFINITY;
          mx = fmaxf(mx, sacc[kb][r]);
        }
      }
      mx = quad_rows_max(mx);
      const float nm = (mx == -INFINITY) ? 0.f : -mx;          // every key masked: all weights exp2(-inf) = 0
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        float4 p4v = make_float4(1.f, 1.f, 1.f, 1.f);
        if (POLICY && (kb < NKB - 1 || !last_empty)) p4v = *reinterpret_cast<const float4*>(&sLB[kb * 16 + 4 * g]);
        const float pw[4] = {p4v.x, p4v.y, p4v.z, p4v.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __builtin_amdgcn_exp2f(sacc[kb][r] + nm);
          if (POLICY) pv *= (kb * 16 + 4 * g + r == q) ? 1.0f : pw[r];      // attn_policy = policy + (1 - policy) * eye
          sacc[kb][r] = pv;
          l += pv;
        }
      }
    }
    l = quad_rows_sum(l);
    const float inv = POLICY ? 1.0f / (l + 1e-6f) : 1.0f / l;
    if (POLICY) {                                                // (attn + eps/N) / (sum + eps); padded keys stay 0
      const float add = 1e-6f / (float)N;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kb < NKB - 2 || kb * 16 + 4 * g + r < N) sacc[kb][r] += add;
    }

    // ---- O^T = V^T P^T
    f32x4 o[4];
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      bf16x8 pf;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pf[j] = (__bf16)sacc[2 * t][j];
        pf[4 + j] = (__bf16)sacc[2 * t + 1][j];
      }
      const bool half = (t == NP - 1) && last_empty;        // rows R.. are not in LDS: re-read the first block's rows (their P is 0)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const unsigned char* p0 = vb[db] + t * 4096;
        const bf16x8 vf = lds_tr_pair16(p0, half ? p0 : p0 + 2048);
        o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, t == 0 ? z4 : o[db], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (q < N) {
      uint16_t* orow = out + ((size_t)b * N + q) * (H * 64) + h * 64 + 8 * g;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        uint4 pk;
        pk.x = pack_bf16x2(o[2 * u][0] * inv, o[2 * u][1] * inv);
        pk.y = pack_bf16x2(o[2 * u][2] * inv, o[2 * u][3] * inv);
        pk.z = pack_bf16x2(o[2 * u + 1][0] * inv, o[2 * u + 1][1] * inv);
        pk.w = pack_bf16x2(o[2 * u + 1][2] * inv, o[2 * u + 1][3] * inv);
        *reinterpret_cast<uint4*>(orow + 32 * u) = pk;
      }
    }
    if (COLSUM) {
      const float wq = q < N ? inv : 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sres = row16_sum(sacc[kb][r] * wq);      // over the 16 queries of the block (the DPP row = equal g)
          if (li == kb) colacc[r] += sres;
        }
    }
    if (cls_rows != nullptr && qb == 0 && li == 0) {
      float* crow = cls_rows + ((size_t)b * H + h) * N;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const int key = kb * 16 + 4 * g;
        if (kb < NKB - 2 || key + 3 < N) {       // rows of N floats are only 4-byte aligned: four scalar stores, one branch
#pragma unroll
          for (int r = 0; r < 4; ++r) crow[key + r] = sacc[kb][r] * inv;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key + r < N) crow[key + r] = sacc[kb][r] * inv;
        }
      }
    }
  }
  if (COLSUM && li < NKB) {
    float* crow = colsum_part + (((size_t)b * H + h) * 4 + wave) * N;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = li * 16 + 4 * g + r;
      if (key < N) crow[key] = colacc[r];
    }
  }
}

// ---- long sequences (224 < N <= 608, i.e. 384^2 inputs: N = 577) -----------------------------------------------------------
// Same MFMA scheme, but the score row no longer fits the register file, so the keys are walked in chunks of CH 32-key blocks,
// twice: pass 1 finds every query's row maximum and normaliser (running max / rescaled sum), pass 2 recomputes the scores,
// exponentiates against the final maximum and feeds P.V.  Probabilities are therefore final when they are produced, which
// keeps the CLS-row and column-sum side outputs exact (no retro-active rescaling).  K and V^T of one (image, head) fill the LDS
// (2 x 77 KB at N = 577): one workgroup per CU.
constexpr int LCH = 4;   // key blocks per chunk: 64 score registers live

template <bool COLSUM>
__global__ __launch_bounds__(256, 1) void attention_long_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                                float* __restrict__ cls_rows, const float* __restrict__ size,
                                                                float* __restrict__ colsum_part, int N, int H, int NKB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int RS = NKB * 64 + 16;
  unsigned char* sK = smem;                               // [NKB*32][128 B] swizzled
  unsigned char* sVt = sK + (size_t)NKB * 32 * 128;       // [64][RS]
  float* sLB = reinterpret_cast<float*>(sVt + (size_t)64 * RS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int ldq = 3 * H * 64;
  const uint16_t* base = qkv + (size_t)b * N * ldq;
  const int qcol = h * 64, kcol = H * 64 + h * 64, vcol = 2 * H * 64 + h * 64;
  // NKB is a run-time value here: the staging loops go in groups of EIGHT 16-byte loads per thread, all issued before the
  // first LDS write of the group (as rolled load-wait-write loops they were 2 * NKB dependent global round trips, 38 at N = 577)
  for (int it0 = 0; it0 < NKB; it0 += 8) {
    uint4 kreg[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = tid + 256 * (it0 + u);
      const int key = g >> 3, c = g & 7;
      kreg[u] = make_uint4(0u, 0u, 0u, 0u);
      if (it0 + u < NKB && key < N) kreg[u] = *reinterpret_cast<const uint4*>(base + (size_t)key * ldq + kcol + c * 8);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = tid + 256 * (it0 + u);
      if (it0 + u < NKB) *reinterpret_cast<uint4*>(sK + kswz(g >> 3, g & 7)) = kreg[u];
    }
  }
  // V transposed: lanes 0-31 take the 32 keys of a block for d-chunk c, lanes 32-63 chunk c+1
  for (int it0 = 0; it0 < NKB; it0 += 8) {
    uint4 vreg[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int un = wave + 4 * (it0 + u);
      const int kb = un >> 2, c = 2 * (un & 3) + (lane >> 5);
      const int key = kb * 32 + (lane & 31);
      vreg[u] = make_uint4(0u, 0u, 0u, 0u);
      if (it0 + u < NKB && key < N) vreg[u] = *reinterpret_cast<const uint4*>(base + (size_t)key * ldq + vcol + c * 8);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (it0 + u < NKB) {
        const int un = wave + 4 * (it0 + u);
        const int kb = un >> 2, c = 2 * (un & 3) + (lane >> 5);
        const int kl = lane & 31;
        unsigned short* dst = reinterpret_cast<unsigned short*>(sVt + (size_t)(8 * c) * RS + (kb * 32 + swap23(kl)) * 2);
        const unsigned int w[4] = {vreg[u].x, vreg[u].y, vreg[u].z, vreg[u].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned int word = w[e >> 1];
          dst[(e * RS) >> 1] = (unsigned short)((e & 1) ? (word >> 16) : (word & 0xffffu));
        }
      }
    }
  }
  for (int key = tid; key < NKB * 32; key += 256)
    sLB[key] = (size != nullptr && key < N) ? __builtin_amdgcn_logf(size[(size_t)b * N + key]) : 0.f;
  __syncthreads();

  const int ql = lane & 31, hh = lane >> 5;
  const int nqb = (N + 31) >> 5;
  const float c_exp = 0.125f * 1.44269504088896340736f;
  const int r_own = ((ql & 1) << 3) | ((ql & 2) << 1) | ((ql & 4) >> 1) | ((ql & 8) >> 3);   // register this lane owns after the reduce
  float* colrow = COLSUM ? colsum_part + (((size_t)b * H + h) * 4 + wave) * N : nullptr;
  if (COLSUM && (ql & 16) == 0)                                     // zeroed by the lanes that accumulate below (same-thread ordering)
    for (int kb = 0; kb < NKB; ++kb) {
      const int key = kb * 32 + (r_own & 3) + 8 * (r_own >> 2) + 4 * hh;
      if (key < N) colrow[key] = 0.f;
    }

  for (int qb = wave; qb < nqb; qb += 4) {
    const int q = qb * 32 + ql;
    const uint16_t* qrow = base + (size_t)min(q, N - 1) * ldq + qcol + 8 * hh;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s);

    auto scores = [&](int kb0, f32x16* sacc) __attribute__((always_inline)) {
#pragma unroll
      for (int c = 0; c < LCH; ++c) {
        const int kb = kb0 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[c][r] = 0.f;
        if (kb < NKB) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + kswz(kb * 32 + ql, 2 * s + hh));
            sacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[c], 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          sacc[c][r] = key < N ? sacc[c][r] * c_exp + sLB[min(key, NKB * 32 - 1)] : -INFINITY;
        }
      }
    };

    // ---- pass 1: row maximum and normaliser
    float mx = -INFINITY, l = 0.f;
    for (int kb0 = 0; kb0 < NKB; kb0 += LCH) {
      f32x16 sacc[LCH];
      scores(kb0, sacc);
      float cm = mx;
#pragma unroll
      for (int c = 0; c < LCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) cm = fmaxf(cm, sacc[c][r]);
      cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
      float add = 0.f;
#pragma unroll
      for (int c = 0; c < LCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) add += __builtin_amdgcn_exp2f(sacc[c][r] - cm);
      l = l * __builtin_amdgcn_exp2f(mx - cm) + add;     // mx = -inf on the first chunk: exp2(-inf) = 0
      mx = cm;
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const float wq = q < N ? inv : 0.f;

    // ---- pass 2: P = exp2(S - mx), O^T = Vt P^T
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    for (int kb0 = 0; kb0 < NKB; kb0 += LCH) {
      f32x16 sacc[LCH];
      scores(kb0, sacc);
#pragma unroll
      for (int c = 0; c < LCH; ++c) {
        const int kb = kb0 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[c][r] = __builtin_amdgcn_exp2f(sacc[c][r] - mx);
        if (kb < NKB) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (__bf16)sacc[c][8 * s + j];
#pragma unroll
            for (int db = 0; db < 2; ++db) {
              const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sVt + (size_t)(db * 32 + ql) * RS + (kb * 32 + 16 * s + 8 * hh) * 2);
              o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
            }
          }
          if (cls_rows != nullptr && qb == 0 && ql == 0) {
            float* crow = cls_rows + ((size_t)b * H + h) * N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
              if (key < N) crow[key] = sacc[c][r] * inv;
            }
          }
          if (COLSUM) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = sacc[c][r] * wq;
            colsum_step<16, 1>(v, ql);
            colsum_step<8, 2>(v, ql);
            colsum_step<4, 4>(v, ql);
            colsum_step<2, 8>(v, ql);
            const float tot = v[0] + __shfl_xor(v[0], 16, 64);
            const int key = kb * 32 + (r_own & 3) + 8 * (r_own >> 2) + 4 * hh;
            if ((ql & 16) == 0 && key < N) colrow[key] += tot;       // one owner lane per key, query blocks in order: deterministic
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (q < N) {
      uint16_t* orow = out + ((size_t)b * N + q) * (H * 64) + h * 64 + 4 * hh;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 pk;
          pk.x = pack_bf16x2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
          pk.y = pack_bf16x2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(orow + db * 32 + 8 * g) = pk;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Long sequences (224 < N, the 384^2 inputs: N = 577) without column sums: keys go through the LDS in chunks of 128 with an
// online softmax, one workgroup per (image, head, group of four 32-query blocks).  The register-resident formulation above
// would need K and V^T of all 608 keys in LDS (152 KiB: one workgroup = one wave per SIMD per CU) and scores the keys twice;
// here a workgroup holds 34 KiB, so three of them share a CU and overlap each other's staging.
//   running maximum: kept as an INTEGER in the log2 domain (ceil of the largest logit seen), so the factor between two chunks'
//   reference points is an exact power of two: rescaling O and l is exact in fp32, and the bf16 rounding of P commutes with it
//   -- the result equals the two-pass formulation with the global ceil(max) as its reference point.
//   The CLS row (query 0) is written as raw log2-domain logits chunk by chunk and normalised by the same two lanes at the end.
constexpr int FKB = 4;                       // 32-key blocks per chunk
constexpr int FRS = FKB * 64 + 16;           // V^T row stride in bytes
#ifndef TR_FLASH_WGS
#define TR_FLASH_WGS 3
#endif
// POLICY (DyViT training at 384^2, softmax_with_policy dyvit.py:39-51): `size` carries the keep policy [B,N]; with e = exp(s - max),
//   a = e * pi (pi = policy[key], 1 on the diagonal): out = (sum_k a_k v_k + eps/N sum_k v_k) / (sum_k a_k + eps).  eps is relative to the
//   TRUE row maximum, the accumulators to the integer reference point m >= max: eps' = eps 2^(max - m); sum_k v_k (all valid keys) is
//   collected once per workgroup on the vector ALUs while the chunks pass through the LDS.
template <bool POLICY>
__global__ __launch_bounds__(256, TR_FLASH_WGS) void attention_flash_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                                 float* __restrict__ cls_rows, const float* __restrict__ size,
                                                                 float* __restrict__ stats, int N, int H, int nqg) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[FKB * 32 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char sVt[64 * FRS];
  __shared__ float sLB[FKB * 32];
  __shared__ float sVsum[POLICY ? 64 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the query groups of one (image, head) all walk the same K and V: keep them on ONE XCD so that its L2 serves the re-reads
  // (r04a: 632 MB fetched per launch at N = 577 against 227 MB algorithmic with the groups dealt round-robin over the eight L2s)
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lid / nqg, qg = lid - bh * nqg;
  const int b = bh / H, h = bh - b * H;
  const int ldq = 3 * H * 64;
  const uint16_t* base = qkv + (size_t)b * N * ldq;
  const int qcol = h * 64, kcol = H * 64 + h * 64, vcol = 2 * H * 64 + h * 64;
  const int ql = lane & 31, hh = lane >> 5;
  const int nqb = (N + 31) >> 5;
  const int qb = qg * 4 + wave;                      // this wave's query block (may lie past the end: it still helps staging)
  const int q = qb * 32 + ql;
  const bool live = qb < nqb;
  const float c_exp = 0.125f * 1.44269504088896340736f;

  bf16x8 qf[4];
  {
    const uint16_t* qrow = base + (size_t)min(q, N - 1) * ldq + qcol + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s);
  }
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m = -INFINITY, l = 0.f;                      // m: integer-valued reference point; l: this lane's share of the row sum
  float mt = -INFINITY, vs = 0.f;                    // POLICY: the row's true maximum; this thread's share of sum_k v_k[d = tid >> 2]
  float* crow = (cls_rows != nullptr && qb == 0 && ql == 0) ? cls_rows + ((size_t)b * H + h) * N : nullptr;

  const int nch = (N + FKB * 32 - 1) / (FKB * 32);
  for (int ch = 0; ch < nch; ++ch) {
    const int key0 = ch * FKB * 32;
    __syncthreads();                                 // the previous chunk's fragment reads are done
    {
      // stage the chunk: 4 K loads and 4 V loads per thread, all issued before the first LDS write
      uint4 kreg[FKB], vreg[FKB];
#pragma unroll
      for (int it = 0; it < FKB; ++it) {
        const int g = tid + 256 * it;
        const int key = key0 + (g >> 3), c = g & 7;
        kreg[it] = *reinterpret_cast<const uint4*>(base + (size_t)min(key, N - 1) * ldq + kcol + c * 8);   // branch-free; zeroed below
      }
#pragma unroll
      for (int it = 0; it < FKB; ++it) {
        const int u = wave + 4 * it;
        const int kb = u >> 2, c = 2 * (u & 3) + (lane >> 5);
        const int key = key0 + kb * 32 + (lane & 31);
        vreg[it] = *reinterpret_cast<const uint4*>(base + (size_t)min(key, N - 1) * ldq + vcol + c * 8);
      }
      if (tid < FKB * 32) {
        const int key = key0 + tid;
        if (POLICY) sLB[tid] = key < N ? size[(size_t)b * N + key] : 0.f;
        else sLB[tid] = (size != nullptr && key < N) ? __builtin_amdgcn_logf(size[(size_t)b * N + key]) : 0.f;
      }
#pragma unroll
      for (int it = 0; it < FKB; ++it) {
        const int g = tid + 256 * it;
        if (key0 + (g >> 3) >= N) kreg[it] = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(sK + kswz(g >> 3, g & 7)) = kreg[it];
      }
#pragma unroll
      for (int it = 0; it < FKB; ++it) {
        const int u = wave + 4 * it;
        const int kb = u >> 2, c = 2 * (u & 3) + (lane >> 5);
        if (key0 + kb * 32 + (lane & 31) >= N) vreg[it] = make_uint4(0u, 0u, 0u, 0u);
        unsigned short* dst = reinterpret_cast<unsigned short*>(sVt + (8 * c) * FRS + (kb * 32 + swap23(lane & 31)) * 2);
        const unsigned int w[4] = {vreg[it].x, vreg[it].y, vreg[it].z, vreg[it].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned int word = w[e >> 1];
          dst[(e * FRS) >> 1] = (unsigned short)((e & 1) ? (word >> 16) : (word & 0xffffu));
        }
      }
    }
    __syncthreads();
    if (POLICY) {              // sum_k v_k: row d = tid >> 2 of V^T, a quarter of the chunk's keys per thread (rows past N are zero)
      const unsigned char* vrow = sVt + (tid >> 2) * FRS + (tid & 3) * (FKB * 16);
#pragma unroll
      for (int c = 0; c < FKB; ++c) {
        const uint4 w = *reinterpret_cast<const uint4*>(vrow + 16 * c);
        const unsigned int ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) vs += __builtin_bit_cast(float, ww[e] << 16) + __builtin_bit_cast(float, ww[e] & 0xffff0000u);
      }
    }
    if (!live) continue;

    f32x16 sacc[FKB];
#pragma unroll
    for (int kb = 0; kb < FKB; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + kswz(kb * 32 + ql, 2 * s + hh));
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[kb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < FKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        sacc[kb][r] = POLICY ? sacc[kb][r] * c_exp : sacc[kb][r] * c_exp + sLB[kl];   // log2-domain logit (+ log2 size[key]: proportional attention)
        if (key0 + kl >= N) sacc[kb][r] = -INFINITY;
        mx = fmaxf(mx, sacc[kb][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (POLICY) mt = fmaxf(mt, mx);
    if (crow != nullptr) {
#pragma unroll
      for (int kb = 0; kb < FKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key < N) crow[key] = sacc[kb][r];
        }
    }
    // a chunk (or, with key masks, a prefix of chunks) may be masked entirely: the reference point stays at -inf until
    // the first valid key, and a -inf reference is replaced by 0 in the exponents (all of them are exp2(-inf) = 0 then)
    const float m_new = fmaxf(m, ceilf(mx));
    const float alpha = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - m_new);   // exact power of two
    m = m_new;
    const float m_ref = (m_new == -INFINITY) ? 0.f : m_new;
    float lsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < FKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pv = __builtin_amdgcn_exp2f(sacc[kb][r] - m_ref);
        if (POLICY) {                                           // attn_policy = policy + (1 - policy) * eye
          const int kl = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          pv *= (key0 + kl == q) ? 1.0f : sLB[kl];
        }
        sacc[kb][r] = pv;
        lsum += pv;
      }
    l = l * alpha + lsum;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
#pragma unroll
    for (int kb = 0; kb < FKB; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (__bf16)sacc[kb][8 * s + j];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sVt + (db * 32 + ql) * FRS + (kb * 32 + 16 * s + 8 * hh) * 2);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[db], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  if (POLICY) {
    vs += __shfl_xor(vs, 1, 64);
    vs += __shfl_xor(vs, 2, 64);
    if ((tid & 3) == 0) sVsum[tid >> 2] = vs;
    __syncthreads();
  }
  if (!live) return;
  l += __shfl_xor(l, 32, 64);
  float inv = 1.0f / l;
  if (POLICY) {
    const float eps = 1e-6f * __builtin_amdgcn_exp2f(mt - m);          // (attn + eps/N) / (sum + eps), eps in the accumulators' units
    const float add = eps / (float)N;
    inv = 1.0f / (l + eps);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] += add * sVsum[db * 32 + 8 * (r >> 2) + 4 * hh + (r & 3)];
  }
  if (q < N) {
    uint16_t* orow = out + ((size_t)b * N + q) * (H * 64) + h * 64 + 4 * hh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf16x2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
        pk.y = pack_bf16x2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(orow + db * 32 + 8 * g) = pk;
      }
  }
  // per-query reference point and 1 / normaliser for attention_colsum_kernel: rows 0 and 1 of the [B,H,4,N] column-sum buffer
  if (stats != nullptr && q < N && hh == 0) {
    float* st = stats + ((size_t)b * H + h) * 4 * N;
    st[q] = m;
    st[N + q] = inv;
  }
  if (crow != nullptr) {
    // the two lanes that own query 0 wrote its logits (keys with ((key >> 2) & 1) == hh): same thread, same addresses
    for (int key = 0; key < N; ++key)
      if (((key >> 2) & 1) == hh) crow[key] = __builtin_amdgcn_exp2f(crow[key] - m) * inv;
  }
}

// Column sums of the softmax matrix for long sequences (K-Medoids at 384^2 inputs): second pass after attention_flash_kernel,
// which leaves every query's reference point m and 1 / normaliser in rows 0 and 1 of the [B,H,4,N] partial buffer.  One workgroup
// per (image, head, chunk of 128 keys) holds the K chunk in LDS and walks ALL query blocks: S^T = K Q^T again (half the FLOPs of
// the attention itself), p = exp2(s * c - m_q) / l_q, summed over the queries of each wave by the same transpose-reduce as
// attention_kernel<COLSUM>.  Waves 0,2 and 1,3 are combined (fixed order) into rows 2 and 3; the launcher zeroes rows 0 and 1
// afterwards, so the consumer's sum over the four rows is unchanged.
__global__ __launch_bounds__(256, 3) void attention_colsum_kernel(const uint16_t* __restrict__ qkv, float* __restrict__ part, int N,
                                                                  int H, int nkc) {
  __shared__ __attribute__((aligned(16))) unsigned char sK[FKB * 32 * 128];
  __shared__ float sAcc[2][FKB * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);        // the key chunks of one (image, head) re-read the same queries: one XCD
  const int bh = lid / nkc, kc = lid - bh * nkc;
  const int b = bh / H, h = bh - b * H;
  const int ldq = 3 * H * 64;
  const uint16_t* base = qkv + (size_t)b * N * ldq;
  const int qcol = h * 64, kcol = H * 64 + h * 64;
  const int ql = lane & 31, hh = lane >> 5;
  const int nqb = (N + 31) >> 5;
  const int key0 = kc * FKB * 32;
  const float c_exp = 0.125f * 1.44269504088896340736f;
  float* pb = part + (size_t)bh * 4 * N;                 // rows 0,1: m and 1/l per query (read); rows 2,3: column sums (written)
  {
    uint4 kreg[FKB];
#pragma unroll
    for (int it = 0; it < FKB; ++it) {
      const int g = tid + 256 * it;
      kreg[it] = *reinterpret_cast<const uint4*>(base + (size_t)min(key0 + (g >> 3), N - 1) * ldq + kcol + (g & 7) * 8);
    }
#pragma unroll
    for (int it = 0; it < FKB; ++it) {
      const int g = tid + 256 * it;
      if (key0 + (g >> 3) >= N) kreg[it] = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(sK + kswz(g >> 3, g & 7)) = kreg[it];
    }
  }
  __syncthreads();
  float colacc[FKB];
#pragma unroll
  for (int kb = 0; kb < FKB; ++kb) colacc[kb] = 0.f;
  for (int qb = wave; qb < nqb; qb += 4) {
    const int q = qb * 32 + ql;
    const int qc = min(q, N - 1);
    const uint16_t* qrow = base + (size_t)qc * ldq + qcol + 8 * hh;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s);
    const float mq = pb[qc];
    const float wq = q < N ? pb[N + qc] : 0.f;
#pragma unroll
    for (int kb = 0; kb < FKB; ++kb) {
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + kswz(kb * 32 + ql, 2 * s + hh));
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc, 0, 0, 0);
      }
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        v[r] = key < N ? __builtin_amdgcn_exp2f(sacc[r] * c_exp - mq) * wq : 0.f;
      }
      colsum_step<16, 1>(v, ql);
      colsum_step<8, 2>(v, ql);
      colsum_step<4, 4>(v, ql);
      colsum_step<2, 8>(v, ql);
      colacc[kb] += v[0] + __shfl_xor(v[0], 16, 64);
    }
  }
  // after the transpose-reduce, lane bits (b0,b1,b2,b3) of ql select register r = 8 b0 + 4 b1 + 2 b2 + b3
  const int r = ((ql & 1) << 3) | ((ql & 2) << 1) | ((ql & 4) >> 1) | ((ql & 8) >> 3);
  const int kl = (r & 3) + 8 * (r >> 2) + 4 * hh;
  if (wave >= 2 && (ql & 16) == 0) {
#pragma unroll
    for (int kb = 0; kb < FKB; ++kb) sAcc[wave - 2][kb * 32 + kl] = colacc[kb];
  }
  __syncthreads();
  if (wave < 2 && (ql & 16) == 0) {
    float* crow = pb + (size_t)(2 + wave) * N;
#pragma unroll
    for (int kb = 0; kb < FKB; ++kb) {
      const int key = key0 + kb * 32 + kl;
      if (key < N) crow[key] = colacc[kb] + sAcc[wave][kb * 32 + kl];
    }
  }
}

template <int NP>
int launch_attention16(const uint16_t* qkv, uint16_t* out, float* cls_rows, const float* size, float* colsum_part, int B, int N, int H,
                       hipStream_t st, bool policy = false) {
  const int R = ((N + 15) >> 4) << 4;
  const size_t lds = (size_t)R * 256 + ((policy || size) ? (size_t)R * 4 : 0);
  if (policy)
    hipLaunchKernelGGL((attention16_kernel<NP, false, true, false>), dim3(B * H), dim3(A16_NT), lds, st, qkv, out, cls_rows, size, colsum_part, N, H);
  else if (colsum_part && size)
    hipLaunchKernelGGL((attention16_kernel<NP, true, false, true>), dim3(B * H), dim3(A16_NT), lds, st, qkv, out, cls_rows, size, colsum_part, N, H);
  else if (colsum_part)
    hipLaunchKernelGGL((attention16_kernel<NP, true, false, false>), dim3(B * H), dim3(A16_NT), lds, st, qkv, out, cls_rows, size, colsum_part, N, H);
  else if (size)
    hipLaunchKernelGGL((attention16_kernel<NP, false, false, true>), dim3(B * H), dim3(A16_NT), lds, st, qkv, out, cls_rows, size, colsum_part, N, H);
  else
    hipLaunchKernelGGL((attention16_kernel<NP, false, false, false>), dim3(B * H), dim3(A16_NT), lds, st, qkv, out, cls_rows, size, colsum_part, N, H);
  return 0;
}

template <int NKB>
int launch_attention(const uint16_t* qkv, uint16_t* out, float* cls_rows, const float* size, float* colsum_part, int B, int N, int H,
                     hipStream_t st, bool policy = false) {
  // BIAS: the keys carry log2(size) / a mask (ToMe, ATS, heuristic masks); without it the softmax takes its packed fast path
  if (policy)
    hipLaunchKernelGGL((attention_kernel<NKB, false, true, true>), dim3(B * H), dim3(256), 0, st, qkv, out, cls_rows, size, colsum_part, N, H);
  else if (colsum_part && size)
    hipLaunchKernelGGL((attention_kernel<NKB, true, false, true>), dim3(B * H), dim3(256), 0, st, qkv, out, cls_rows, size, colsum_part, N, H);
  else if (colsum_part)
    hipLaunchKernelGGL((attention_kernel<NKB, true, false, false>), dim3(B * H), dim3(256), 0, st, qkv, out, cls_rows, size, colsum_part, N, H);
  else if (size)
    hipLaunchKernelGGL((attention_kernel<NKB, false, false, true>), dim3(B * H), dim3(256), 0, st, qkv, out, cls_rows, size, colsum_part, N, H);
  else
    hipLaunchKernelGGL((attention_kernel<NKB, false, false, false>), dim3(B * H), dim3(256), 0, st, qkv, out, cls_rows, size, colsum_part, N, H);
  return 0;
}

}  // namespace

extern "C" int tr_attention_bf16(const uint16_t* qkv, uint16_t* out, float* cls_rows, const float* size, float* colsum_part, int B,
                                 int N, int H, tr_stream_t s) {
  TR_REQUIRE(qkv && out, TR_ERR_NULL, "tr_attention_bf16: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 1, TR_ERR_SHAPE, "tr_attention_bf16: bad shape B=%d N=%d H=%d", B, N, H);
  TR_REQUIRE(N <= 608 || colsum_part == nullptr || size == nullptr, TR_ERR_SHAPE,
             "tr_attention_bf16: column sums together with a key bias need N <= 608 (N=%d: K and V^T of one head must fit the LDS)", N);
  TR_REQUIRE(tr_aligned16(qkv) && tr_aligned16(out), TR_ERR_ALIGN, "tr_attention_bf16: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  tr_prof_note(N <= 224 ? "attention_kernel" : "attention_flash_kernel", 4.0 * B * H * (double)N * N * 64, 2.0 * B * N * 4.0 * H * 64);
#ifdef TR_ATT_LAB          // lab builds only (tools/attn_lab.py): smallest N that takes the online-softmax kernel, from the environment
  static const int flash_min = [] {
    const char* e = getenv("TR_ATT_FLASH_MIN");
    return e ? atoi(e) : 225;
  }();
#else
  constexpr int flash_min = 225;   // N <= 224: whole score row in registers; beyond: online softmax over 128-key chunks
#endif
  if (N >= flash_min && (colsum_part == nullptr || size == nullptr)) {
    const int nqg = ((N + 31) / 32 + 3) / 4;
    hipLaunchKernelGGL(attention_flash_kernel<false>, dim3(B * H * nqg), dim3(256), 0, st, qkv, out, cls_rows, size, colsum_part, N, H, nqg);
    if (colsum_part != nullptr) {
      // column sums by a second pass over the keys (rows 2,3 of every [4][N] block), then the per-query statistics the first pass
      // left in rows 0,1 are cleared: the consumer adds all four rows
      const int nkc = (N + FKB * 32 - 1) / (FKB * 32);
      hipLaunchKernelGGL(attention_colsum_kernel, dim3(B * H * nkc), dim3(256), 0, st, qkv, colsum_part, N, H, nkc);
      hipError_t e = hipMemset2DAsync(colsum_part, (size_t)4 * N * sizeof(float), 0, (size_t)2 * N * sizeof(float), (size_t)B * H, st);
      TR_REQUIRE(e == hipSuccess, TR_ERR_LAUNCH, "tr_attention_bf16: clearing the statistics rows failed: %s", hipGetErrorString(e));
    }
    TR_CHECK_LAUNCH("tr_attention_bf16");
    return TR_OK;
  }
  if (N > 224) {
    const int nkb = (N + 31) / 32;
    const size_t lds = (size_t)nkb * 32 * 128 + (size_t)64 * (nkb * 64 + 16) + (size_t)nkb * 32 * 4;
    if (colsum_part) TR_RESERVE_LDS(reinterpret_cast<const void*>(attention_long_kernel<true>), lds, "tr_attention_bf16");
    else TR_RESERVE_LDS(reinterpret_cast<const void*>(attention_long_kernel<false>), lds, "tr_attention_bf16");
    if (colsum_part)
      hipLaunchKernelGGL(attention_long_kernel<true>, dim3(B * H), dim3(256), lds, st, qkv, out, cls_rows, size, colsum_part, N, H, nkb);
    else
      hipLaunchKernelGGL(attention_long_kernel<false>, dim3(B * H), dim3(256), lds, st, qkv, out, cls_rows, size, colsum_part, N, H, nkb);
    TR_CHECK_LAUNCH("tr_attention_bf16");
    return TR_OK;
  }
  static const bool use_old = [] { const char* e = getenv("TR_ATT_OLD"); return e && atoi(e) != 0; }();   // lab: the 32-query kernel
  if (!use_old) {
    switch ((N + 31) / 32) {
      case 1: launch_attention16<1>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
      case 2: launch_attention16<2>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
      case 3: launch_attention16<3>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
      case 4: launch_attention16<4>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
      case 5: launch_attention16<5>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
      case 6: launch_attention16<6>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
      default: launch_attention16<7>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
    }
    TR_CHECK_LAUNCH("tr_attention_bf16");
    return TR_OK;
  }
  switch ((N + 31) / 32) {
    case 1: launch_attention<1>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
    case 2: launch_attention<2>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
    case 3: launch_attention<3>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
    case 4: launch_attention<4>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
    case 5: launch_attention<5>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
    case 6: launch_attention<6>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
    default: launch_attention<7>(qkv, out, cls_rows, size, colsum_part, B, N, H, st); break;
  }
  TR_CHECK_LAUNCH("tr_attention_bf16");
  return TR_OK;
}

// a11: Policy_Attention.forward dyvit.py:53-67 with softmax_with_policy (:39-51) -- the attention of DyViT's TRAINING forward,
// where tokens are masked by a keep policy instead of being removed.  Beyond 224 tokens (384^2 inputs): the online-softmax kernel.
extern "C" int tr_attention_policy_bf16(const uint16_t* qkv, uint16_t* out, const float* policy, int B, int N, int H, tr_stream_t s) {
  TR_REQUIRE(qkv && out && policy, TR_ERR_NULL, "tr_attention_policy_bf16: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 1, TR_ERR_SHAPE, "tr_attention_policy_bf16: bad shape B=%d N=%d H=%d", B, N, H);
  TR_REQUIRE(tr_aligned16(qkv) && tr_aligned16(out), TR_ERR_ALIGN, "tr_attention_policy_bf16: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  if (N > 224) {
    const int nqg = ((N + 31) / 32 + 3) / 4;
    tr_prof_note("attention_flash_kernel<policy>", 4.0 * B * H * (double)N * N * 64, 2.0 * B * N * 4.0 * H * 64);
    hipLaunchKernelGGL(attention_flash_kernel<true>, dim3(B * H * nqg), dim3(256), 0, st, qkv, out, static_cast<float*>(nullptr), policy,
                       static_cast<float*>(nullptr), N, H, nqg);
    TR_CHECK_LAUNCH("tr_attention_policy_bf16");
    return TR_OK;
  }
  switch ((N + 31) / 32) {
    case 1: launch_attention<1>(qkv, out, nullptr, policy, nullptr, B, N, H, st, true); break;
    case 2: launch_attention<2>(qkv, out, nullptr, policy, nullptr, B, N, H, st, true); break;
    case 3: launch_attention<3>(qkv, out, nullptr, policy, nullptr, B, N, H, st, true); break;
    case 4: launch_attention<4>(qkv, out, nullptr, policy, nullptr, B, N, H, st, true); break;
    case 5: launch_attention<5>(qkv, out, nullptr, policy, nullptr, B, N, H, st, true); break;
    case 6: launch_attention<6>(qkv, out, nullptr, policy, nullptr, B, N, H, st, true); break;
    default: launch_attention<7>(qkv, out, nullptr, policy, nullptr, B, N, H, st, true); break;
  }
  TR_CHECK_LAUNCH("tr_attention_policy_bf16");
  return TR_OK;
}
