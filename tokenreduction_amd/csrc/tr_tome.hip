// ToMe -- token merging (models/tome.py) on gfx950: bipartite soft matching + size-weighted merge.
//
//   tr_tome_match             bipartite_soft_matching (tome.py:230-277, class_token=True) on metric = k.mean(1) (tome.py:58):
//                             cosine scores between even- and odd-position tokens, row max/argmax, descending rank of the
//                             row maxima, split into merged (src -> dst) and unmerged tokens.  One workgroup per image; the
//                             whole problem (<= 113 x 112 x 64 MACs) lives in LDS, scores in 4x4 register tiles.  Integer outputs; ties: row argmax ->
//                             first index (torch CPU max), rank -> lowest index first (torch's argsort order is unspecified).
//   tr_tome_merge_layernorm   merge_wavg (tome.py:309-323): x = merge(x*size) / merge(size), size = merge(size), with the
//                             pending residual add (x + attn.proj output, tome.py:84) in front and norm2 (tome.py:101) behind,
//                             in ONE pass.  Sources are added to their destination in edge order (= the order torch's CPU
//                             scatter_add applies them), so the fp32 path reproduces the reference's rounding.
// HBM traffic is the algorithmic minimum (and the matching kernel's first phase runs at 4.4 TB/s on it): match reads the K third of qkv once (2*N*H*64 B per image, bf16) and writes
// 4*(na + r) B of indices; merge reads each input row once and writes each output row once.
#include "tr_common.h"
#include "tr_rowops.h"

namespace {

constexpr int TOME_MAX_N = 600;      // tokens incl. CLS: 197 at 224^2 inputs, 577 at 384^2 (metric rows live in LDS: N*65 floats)
constexpr int MST = 65;              // metric row stride in floats (odd: conflict-free column walks)

// order-preserving map float -> uint32 (a < b  <=>  key(a) < key(b)), so (score, lowest j) maxima reduce with one ds_max_u64
__device__ __forceinline__ unsigned long long score_key(float v, int j) {
  unsigned int u = __float_as_uint(v);
  u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
  return ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned int)j);
}

// sum over heads of one 8-wide bf16 chunk of K (head stride 64 elements): NH loads in flight, added in head order
template <int NH>
__device__ __forceinline__ void head_sum_bf16(const uint16_t* kp, int H, float* m) {
  uint4 u[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) u[h] = *reinterpret_cast<const uint4*>(kp + min(h, H - 1) * 64);
#pragma unroll
  for (int h = 0; h < NH; ++h)
    if (h < H) {
      m[0] += __uint_as_float(u[h].x << 16); m[1] += __uint_as_float(u[h].x & 0xffff0000u);
      m[2] += __uint_as_float(u[h].y << 16); m[3] += __uint_as_float(u[h].y & 0xffff0000u);
      m[4] += __uint_as_float(u[h].z << 16); m[5] += __uint_as_float(u[h].z & 0xffff0000u);
      m[6] += __uint_as_float(u[h].w << 16); m[7] += __uint_as_float(u[h].w & 0xffff0000u);
    }
}

constexpr int TMT = 1024;   // 16 waves per image: the kernel is a chain of short latency-bound phases, one workgroup per CU
template <bool F32>
__global__ __launch_bounds__(TMT) void tome_match_kernel(const void* __restrict__ qkv, int32_t* __restrict__ unm_idx,
                                                         int32_t* __restrict__ src_idx, int32_t* __restrict__ dst_idx, int N, int H,
                                                         int r) {
  extern __shared__ __attribute__((aligned(16))) float s_m[];      // [N][MST], dynamic: 50 KB at N = 197, 150 KB at N = 577
  __shared__ unsigned long long s_key[(TOME_MAX_N + 1) / 2];
  __shared__ int s_edge[(TOME_MAX_N + 1) / 2];
  __shared__ unsigned char s_unm[(TOME_MAX_N + 1) / 2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int na = (N + 1) >> 1, nb = N >> 1;
  const int ldq = 3 * H * 64;
  // metric = mean over heads of K (post-bias), then metric / metric.norm(dim=-1)  (tome.py:58, :255).  One 8-wide chunk of a
  // token per lane (16-B loads for bf16), head sum sequential in fp32 like a strided torch mean; the 8 lanes of a token
  // share the sum of squares through the wave.
  for (int item = tid; item < N * 8; item += TMT) {
    const int n = item >> 3, c = item & 7;
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = 0.f;
    const size_t e0 = ((size_t)b * N + n) * ldq + H * 64 + c * 8;
    if (!F32 && H <= 12) {
      // all heads' chunks are requested before the first add (head index clamped: branch-free); summed in head order as before.
      // As a rolled loop over H this was H dependent global round trips per item, half of the kernel's time at N = 197.
      // (Six loads when the model has at most six heads: DeiT-S; twelve otherwise.)
      const uint16_t* kp = reinterpret_cast<const uint16_t*>(qkv) + e0;
      if (H <= 6) head_sum_bf16<6>(kp, H, m);
      else head_sum_bf16<12>(kp, H, m);
    } else
    for (int h = 0; h < H; ++h) {
      if (F32) {
        const float4 u0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(qkv) + e0 + h * 64);
        const float4 u1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(qkv) + e0 + h * 64 + 4);
        m[0] += u0.x; m[1] += u0.y; m[2] += u0.z; m[3] += u0.w; m[4] += u1.x; m[5] += u1.y; m[6] += u1.z; m[7] += u1.w;
      } else {
        const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(qkv) + e0 + h * 64);
        m[0] += __uint_as_float(u.x << 16); m[1] += __uint_as_float(u.x & 0xffff0000u);
        m[2] += __uint_as_float(u.y << 16); m[3] += __uint_as_float(u.y & 0xffff0000u);
        m[4] += __uint_as_float(u.z << 16); m[5] += __uint_as_float(u.z & 0xffff0000u);
        m[6] += __uint_as_float(u.w << 16); m[7] += __uint_as_float(u.w & 0xffff0000u);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = m[k] / (float)H;
    float nrm;
    if (F32) {
      // validation path: fp64 sums rounded once -- within 0.5 ulp of exact, so the ranking can only differ from the
      // reference's fp32 matmul where its own rounding (a few ulp) decides, i.e. on near-ties below ~4e-7
      double ss = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) ss += (double)m[k] * (double)m[k];
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      ss += __shfl_xor(ss, 4, 64);
      nrm = (float)sqrt(ss);
    } else {
      float ss = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) ss = fmaf(m[k], m[k], ss);
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      ss += __shfl_xor(ss, 4, 64);
      nrm = sqrtf(ss);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s_m[n * MST + c * 8 + k] = m[k] / nrm;
  }
  for (int i = tid; i < na; i += TMT) s_key[i] = score_key(-INFINITY, 0);    // row 0 (CLS) stays -inf: never merged (tome.py:259)
  __syncthreads();
  if (!F32) {
    // scores[i][j] = a_i . b_j (a = even tokens, b = odd tokens) on the matrix cores with FP32 OPERANDS (v_mfma_f32_16x16x4_f32: exact
    // fp32 products, fp32 accumulation -- the precision class of the fmaf chain it replaces, NOT a bf16 product): one 16 x 16 tile of the
    // score matrix per wave and step, 16 MFMAs over the 64 metric columns, 32 LDS reads per tile.  (Round 4.  The 4 x 4 register tiles on
    // the VALU read 8 floats from LDS per 16 FMAs: 1.3 MB of LDS traffic per image, ~10 k cycles with the 2-way conflicts of the row
    // stride -- the longest phase of the kernel at N = 197.)  Row max / argmax: the 16 lanes that hold a row's columns reduce an
    // order-preserving (score, lowest j) key with DPP, one ds_max_u64 per row and tile.
    const int lane = tid & 63, wave = tid >> 6;
    const int il = lane & 15, kq = lane >> 4;
    const int nbi = (na + 15) >> 4, nbj = (nb + 15) >> 4;
    for (int t = wave; t < nbi * nbj; t += TMT / 64) {
      const int i0 = (t / nbj) * 16, j0 = (t % nbj) * 16;
      const float* ap = s_m + (size_t)(2 * min(i0 + il, na - 1)) * MST + kq;          // MFMA A: row i0 + il, k = 4 ks + kq
      const float* bp = s_m + (size_t)(2 * min(j0 + il, nb - 1) + 1) * MST + kq;      // MFMA B: column j0 + il
      // all 32 fragment values first (one LDS round trip per tile, not one per MFMA), then two independent accumulator chains
      float av[16], bv[16];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) { av[ks] = ap[4 * ks]; bv[ks] = bp[4 * ks]; }
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ks += 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks + 1], bv[ks + 1], acc1, 0, 0, 0);
      }
      acc += acc1;
      // acc[e]: row i0 + 4 kq + e, column j0 + il
      const int j = j0 + il;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned long long key = j < nb ? score_key(acc[e], j) : 0ull;
#pragma unroll
        for (int step = 0; step < 4; ++step) {
          const int ctl = step == 0 ? 0xB1 : step == 1 ? 0x4E : step == 2 ? 0x141 : 0x140;      // quad swaps, half-row mirror, row mirror
          unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32), olo, ohi;
          switch (ctl) {
            case 0xB1: olo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true); ohi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true); break;
            case 0x4E: olo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true); ohi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true); break;
            case 0x141: olo = __builtin_amdgcn_mov_dpp(lo, 0x141, 0xF, 0xF, true); ohi = __builtin_amdgcn_mov_dpp(hi, 0x141, 0xF, 0xF, true); break;
            default: olo = __builtin_amdgcn_mov_dpp(lo, 0x140, 0xF, 0xF, true); ohi = __builtin_amdgcn_mov_dpp(hi, 0x140, 0xF, 0xF, true); break;
          }
          const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
          key = other > key ? other : key;
        }
        const int i = i0 + 4 * kq + e;
        if (il == 0 && i > 0 && i < na) atomicMax(&s_key[i], key);
      }
    }
  } else {
  // scores[i][j] = a_i . b_j (a = even tokens, b = odd tokens) in 4x4 register tiles, rows/columns STRIDED over the tile grid
    // (i = ti + nti*ii, j = tj + ntj*jj) so the lanes of a wave read consecutive b rows (conflict-free, a rows broadcast);
    // row max/argmax (ties -> lowest j, torch's CPU max) through ds_max_u64 on an order-preserving (score, ~j) key
    const int nti = (na + 3) >> 2, ntj = (nb + 3) >> 2;
    for (int t = tid; t < nti * ntj; t += TMT) {
      const int ti = t / ntj, tj = t - ti * ntj;
      const float* ap[4];
      const float* bp[4];
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        ap[q] = s_m + (size_t)(2 * min(ti + nti * q, na - 1)) * MST;
        bp[q] = s_m + (size_t)(2 * min(tj + ntj * q, nb - 1) + 1) * MST;
      }
      float best[4];
      int arg[4];
      if (F32) {
        double acc[4][4];
  #pragma unroll
        for (int ii = 0; ii < 4; ++ii)
  #pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = 0.0;
        for (int d = 0; d < 64; ++d) {
          double av[4], bv[4];
  #pragma unroll
          for (int q = 0; q < 4; ++q) { av[q] = (double)ap[q][d]; bv[q] = (double)bp[q][d]; }
  #pragma unroll
          for (int ii = 0; ii < 4; ++ii)
  #pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[ii][jj] += av[ii] * bv[jj];
        }
  #pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          best[ii] = -INFINITY; arg[ii] = 0;
  #pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = tj + ntj * jj;
            const float v = (float)acc[ii][jj];
            if (j < nb && v > best[ii]) { best[ii] = v; arg[ii] = j; }     // j ascending, strict >: first index wins ties
          }
        }
      } else {
        float acc[4][4];
  #pragma unroll
        for (int ii = 0; ii < 4; ++ii)
  #pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = 0.f;
  #pragma unroll 8
        for (int d = 0; d < 64; ++d) {
          float av[4], bv[4];
  #pragma unroll
          for (int q = 0; q < 4; ++q) { av[q] = ap[q][d]; bv[q] = bp[q][d]; }
  #pragma unroll
          for (int ii = 0; ii < 4; ++ii)
  #pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = fmaf(av[ii], bv[jj], acc[ii][jj]);
        }
  #pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          best[ii] = -INFINITY; arg[ii] = 0;
  #pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = tj + ntj * jj;
            if (j < nb && acc[ii][jj] > best[ii]) { best[ii] = acc[ii][jj]; arg[ii] = j; }
          }
        }
      }
  #pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = ti + nti * ii;
        if (i > 0 && i < na) atomicMax(&s_key[i], score_key(best[ii], arg[ii]));
      }
    }
  }
  __syncthreads();
  // descending rank of the row maxima (ties: lowest index first) = argsort(descending); keys order exactly like the floats.
  // Sixteen lanes per row, each counting every sixteenth rival, summed over the DPP row (round 4: one thread per row walked all
  // na rivals with a dependent LDS read each -- 2 of the 16 waves busy for ~4 us at N = 197).
  {
    const int c = tid & 15;
    for (int i = tid >> 4; i < ((na + 63) & ~63); i += TMT / 16) {           // whole waves stay in the loop: the DPP sum needs its 16 lanes
      const int ic = min(i, na - 1);
      const unsigned int vi = (unsigned int)(s_key[ic] >> 32);
      int cnt = 0;
      for (int j = c; j < na; j += 16) {
        const unsigned int vj = (unsigned int)(s_key[j] >> 32);
        cnt += (vj > vi) || (vj == vi && j < ic);
      }
      int tot = cnt;
      tot += __builtin_amdgcn_mov_dpp(tot, 0xB1, 0xF, 0xF, true);
      tot += __builtin_amdgcn_mov_dpp(tot, 0x4E, 0xF, 0xF, true);
      tot += __builtin_amdgcn_mov_dpp(tot, 0x141, 0xF, 0xF, true);
      tot += __builtin_amdgcn_mov_dpp(tot, 0x140, 0xF, 0xF, true);
      if (c == 0 && i < na) {
        s_edge[tot] = i;
        s_unm[i] = tot >= r;
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < r; e += TMT) {
    const int i = s_edge[e];
    src_idx[(size_t)b * r + e] = i;
    dst_idx[(size_t)b * r + e] = (int)(0xffffffffu - (unsigned int)(s_key[i] & 0xffffffffull));
  }
  // unmerged tokens, ascending (tome.py:275-277: keeps the class token first): position = unmerged tokens before it, again 16 lanes per row
  {
    const int c = tid & 15;
    for (int i = tid >> 4; i < ((na + 63) & ~63); i += TMT / 16) {
      const int ic = min(i, na - 1);
      int cnt = 0;
      for (int j = c; j < ic; j += 16) cnt += s_unm[j];
      cnt += __builtin_amdgcn_mov_dpp(cnt, 0xB1, 0xF, 0xF, true);
      cnt += __builtin_amdgcn_mov_dpp(cnt, 0x4E, 0xF, 0xF, true);
      cnt += __builtin_amdgcn_mov_dpp(cnt, 0x141, 0xF, 0xF, true);
      cnt += __builtin_amdgcn_mov_dpp(cnt, 0x140, 0xF, 0xF, true);
      if (c == 0 && i < na && s_unm[i]) unm_idx[(size_t)b * (na - r) + cnt] = i;
    }
  }
}

// one wave per OUTPUT row: [unmerged even tokens | all odd tokens]
template <bool F32, int NCH>
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
  float4 v[NCH];
  float sz;
  // (x + pending residual) * size of one input token, accumulated into v
  auto add_token = [&](int t, bool first) __attribute__((always_inline)) {
    const float s = sb ? sb[t] : 1.0f;
    // branch-free loads (chunk index clamped; lanes past the row hold a harmless copy of its last chunk, never stored or summed)
    float4 a[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) a[c] = ln_nt_load4(xb + (size_t)t * D + 4 * min(lane + 64 * c, nchunks - 1));
    if (delta) {
      float4 d[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) d[c] = load_delta4<F32>(delta, dbase + (size_t)t * D + 4 * min(lane + 64 * c, nchunks - 1));
#pragma unroll
      for (int c = 0; c < NCH; ++c) { a[c].x += d[c].x; a[c].y += d[c].y; a[c].z += d[c].z; a[c].w += d[c].w; }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (first) v[c] = make_float4(a[c].x * s, a[c].y * s, a[c].z * s, a[c].w * s);
      else { v[c].x += a[c].x * s; v[c].y += a[c].y * s; v[c].z += a[c].z * s; v[c].w += a[c].w * s; }
    }
    if (first) sz = s; else sz += s;
  };
  if (p < n_unm) {
    add_token(2 * unm_idx[(size_t)b * n_unm + p], true);
  } else {
    const int j = p - n_unm;
    // edge order = the order torch's CPU scatter_add applies the sources.  Lane l holds edge 64*blk + l; the edges that end in
    // this dst token come out of one ballot, lowest edge first.  (A scalar loop over the r edges paid one dependent index load
    // per edge: ~8 us per dst row at r = 16, the kernel's long pole.)  The first 64 edges' indices are requested BEFORE the token's own
    // row, so that their round trip runs under it instead of after it (round 4).
    int dl = -1, sl = 0;
    if (lane < r) {
      dl = dst_idx[(size_t)b * r + lane];
      sl = src_idx[(size_t)b * r + lane];
    }
    add_token(2 * j + 1, true);
    for (int e0 = 0; e0 < r; e0 += 64) {
      if (e0 > 0) {
        dl = -1;
        if (e0 + lane < r) {
          dl = dst_idx[(size_t)b * r + e0 + lane];
          sl = src_idx[(size_t)b * r + e0 + lane];
        }
      }
      unsigned long long m = __ballot(dl == j);
      while (m) {
        const int e = __builtin_ctzll(m);
        m &= m - 1;
        add_token(2 * __builtin_amdgcn_readlane(sl, e), false);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (lane + 64 * c < nchunks) { v[c].x /= sz; v[c].y /= sz; v[c].z /= sz; v[c].w /= sz; }
  const size_t orow = (size_t)b * N_out + p;
  if (lane == 0) size_out[orow] = sz;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (lane + 64 * c < nchunks) ln_nt_store4(v[c], x_out + orow * D + 4 * (lane + 64 * c));
  // norm2
  ln_row_store<F32, NCH>(v, nchunks, lane, D, eps, gamma, beta,
                    F32 ? (void*)(reinterpret_cast<float*>(y) + orow * D) : (void*)(reinterpret_cast<uint16_t*>(y) + orow * D));
}

}  // namespace

extern "C" int tr_tome_match(const void* qkv, int qkv_is_f32, int32_t* unm_idx, int32_t* src_idx, int32_t* dst_idx, int B, int N,
                             int H, int r, tr_stream_t s) {
  TR_REQUIRE(qkv && unm_idx && src_idx && dst_idx, TR_ERR_NULL, "tr_tome_match: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 3 && N <= TOME_MAX_N, TR_ERR_SHAPE, "tr_tome_match: need 3 <= N <= %d (N=%d)", TOME_MAX_N, N);
  TR_REQUIRE(r >= 1 && r <= (N - 1) / 2, TR_ERR_SHAPE, "tr_tome_match: r=%d must be in [1, (N-1)/2] for N=%d (tome.py:253)", r, N);
  TR_REQUIRE(tr_aligned16(qkv), TR_ERR_ALIGN, "tr_tome_match: qkv must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  const size_t lds = (size_t)N * MST * sizeof(float);
  if (qkv_is_f32) {
    TR_RESERVE_LDS(reinterpret_cast<const void*>(tome_match_kernel<true>), lds, "tr_tome_match");
    hipLaunchKernelGGL(tome_match_kernel<true>, dim3(B), dim3(TMT), lds, st, qkv, unm_idx, src_idx, dst_idx, N, H, r);
  } else {
    TR_RESERVE_LDS(reinterpret_cast<const void*>(tome_match_kernel<false>), lds, "tr_tome_match");
    hipLaunchKernelGGL(tome_match_kernel<false>, dim3(B), dim3(TMT), lds, st, qkv, unm_idx, src_idx, dst_idx, N, H, r);
  }
  TR_CHECK_LAUNCH("tr_tome_match");
  return TR_OK;
}

extern "C" int tr_tome_merge_layernorm(const float* x, const void* delta, int f32_path, const float* size_in, const int32_t* unm_idx,
                                       const int32_t* src_idx, const int32_t* dst_idx, const float* gamma, const float* beta,
                                       float* x_out, float* size_out, void* y, int B, int N, int r, int D, float eps, tr_stream_t s) {
  TR_REQUIRE(x && unm_idx && src_idx && dst_idx && gamma && beta && x_out && size_out && y, TR_ERR_NULL, "tr_tome_merge_layernorm: null pointer");
  TR_REQUIRE(B > 0 && N >= 3 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAX_CHUNKS, TR_ERR_SHAPE, "tr_tome_merge_layernorm: bad shape B=%d N=%d D=%d", B, N, D);
  TR_REQUIRE(r >= 1 && r <= (N - 1) / 2, TR_ERR_SHAPE, "tr_tome_merge_layernorm: r=%d out of range for N=%d", r, N);
  TR_REQUIRE(x_out != x, TR_ERR_SHAPE, "tr_tome_merge_layernorm: needs a distinct x_out");
  TR_REQUIRE(tr_aligned16(x) && tr_aligned16(delta) && tr_aligned16(x_out) && tr_aligned16(y) && tr_aligned16(gamma) && tr_aligned16(beta),
             TR_ERR_ALIGN, "tr_tome_merge_layernorm: pointers must be 16-byte aligned");
  const int N_out = N - r;
  const int rblocks = (N_out + 3) / 4;
  hipStream_t st = static_cast<hipStream_t>(s);
  if (f32_path)
    TR_DISPATCH_NCH(D, hipLaunchKernelGGL((tome_merge_layernorm_kernel<true, NCH>), dim3(B * rblocks), dim3(256), 0, st, x, delta, size_in,
                                          unm_idx, src_idx, dst_idx, gamma, beta, x_out, size_out, y, N, r, D, eps));
  else
    TR_DISPATCH_NCH(D, hipLaunchKernelGGL((tome_merge_layernorm_kernel<false, NCH>), dim3(B * rblocks), dim3(256), 0, st, x, delta, size_in,
                                          unm_idx, src_idx, dst_idx, gamma, beta, x_out, size_out, y, N, r, D, eps));
  TR_CHECK_LAUNCH("tr_tome_merge_layernorm");
  return TR_OK;
}
