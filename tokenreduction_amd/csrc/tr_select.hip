// CLS-attention Top-K token selection (+ EViT complement), one workgroup per image.
//
//   scores[b,j] = mean_h attn[b,h,0,1+j]                    topk.py:59-60 == evit.py:81-82
//   idx[b,:K]   = torch.topk(scores, K, largest, sorted)    topk.py:61    == evit.py:83   (descending score order)
//   compl[b,:]  = sorted(set(range(P)) - set(idx))          evit.py:25-46 complement_idx  (ascending)
//
// Integer work, bit-exact by construction: every token computes its RANK by counting
// (#scores greater, ties broken by lower index first) over the P<=1024 scores held in LDS -- as order-preserving integer keys, with
// NaN ranked LARGEST like torch.topk does, so the ranks are a permutation of 0..P-1 for ANY input (a NaN score used to compare
// false against everything: several tokens got rank 0 and other idx slots stayed unwritten) -- and writes itself
// to idx[rank] when rank < K.  That yields the sorted order directly, is deterministic, and needs no sort
// network; P^2 = 38k compares per image at P=196 is noise next to the block's GEMMs.  HBM traffic is the
// algorithmic minimum: 4*H*N bytes of CLS rows in, 4*K (+4*(P-K) +4*P) bytes out per image.
#include "tr_common.h"

namespace {

constexpr int MAX_P = 1024;

// L lanes per token (1: P <= 256, the 224^2 inputs; 4: beyond, where one thread per token walked 576 rivals three times over and the
// kernel took 62 us at B = 64): lane c of a token's quad counts every L-th group of four rivals, the quad sums with two DPP adds.
template <int L>
__global__ __launch_bounds__(256 * L) void cls_topk_kernel(const float* __restrict__ cls_rows, int32_t* __restrict__ idx,
                                                           int32_t* __restrict__ compl_idx, float* __restrict__ scores, int H, int N,
                                                           int K) {
  constexpr int T = 256 * L;
  __shared__ __attribute__((aligned(16))) unsigned int s_sc[MAX_P + 4];       // order keys: key(a) < key(b) <=> a < b; -0 == +0; NaN above +inf
  __shared__ unsigned char s_drop[MAX_P];
  const int b = blockIdx.x;
  const int P = N - 1;
  const int tid = threadIdx.x;
  const float* rows = cls_rows + (size_t)b * H * N;
  for (int j = tid; j < P; j += T) {
    float acc = 0.f;
    if (H <= 12) {                       // every head's value requested before the first add (a rolled loop waits for each in turn)
      float v[12];
#pragma unroll
      for (int h = 0; h < 12; ++h) v[h] = rows[(size_t)min(h, H - 1) * N + 1 + j];
#pragma unroll
      for (int h = 0; h < 12; ++h)
        if (h < H) acc += v[h];          // sequential over heads, like a strided torch sum
    } else {
      for (int h = 0; h < H; ++h) acc += rows[(size_t)h * N + 1 + j];
    }
    const float sc = acc / (float)H;
    unsigned int u = __float_as_uint(sc + 0.0f);                     // -0 -> +0
    u = (sc != sc) ? 0xffffffffu : (u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u));
    s_sc[j] = u;
    if (scores) scores[(size_t)b * P + j] = sc;
  }
  if (tid < 4) s_sc[P + tid] = 0u;             // padding of the last group of four: below every real key (those are never 0)
  __syncthreads();
  const int c = tid % L;
  const int Pr = (P + 255) & ~255;             // whole quads stay in the loops: the DPP sums need their lanes
  for (int i = tid / L; i < Pr; i += 256) {
    const int ic = min(i, P - 1);
    const unsigned int si = s_sc[ic];
    int rank = 0;
    for (int j = 4 * c; j < P; j += 4 * L) {   // four rivals per LDS read
      const uint4 q = *reinterpret_cast<const uint4*>(&s_sc[j]);
      rank += (q.x > si) || (q.x == si && j < ic);
      rank += (q.y > si) || (q.y == si && j + 1 < ic);
      rank += (q.z > si) || (q.z == si && j + 2 < ic);
      rank += (q.w > si) || (q.w == si && j + 3 < ic);
    }
    if (L == 4) {
      rank += __builtin_amdgcn_mov_dpp(rank, 0xB1, 0xF, 0xF, true);
      rank += __builtin_amdgcn_mov_dpp(rank, 0x4E, 0xF, 0xF, true);
    }
    if (c == 0 && i < P) {
      if (rank < K) idx[(size_t)b * K + rank] = i;
      s_drop[i] = rank >= K;
    }
  }
  if (compl_idx == nullptr) return;
  __syncthreads();
  for (int i = tid / L; i < Pr; i += 256) {
    const int ic = min(i, P - 1);
    int pos = 0;
    for (int j = c; j < ic; j += L) pos += s_drop[j];
    if (L == 4) {
      pos += __builtin_amdgcn_mov_dpp(pos, 0xB1, 0xF, 0xF, true);
      pos += __builtin_amdgcn_mov_dpp(pos, 0x4E, 0xF, 0xF, true);
    }
    if (c == 0 && i < P && s_drop[i]) compl_idx[(size_t)b * (P - K) + pos] = i;
  }
}

}  // namespace

extern "C" int tr_cls_topk(const float* cls_rows, int32_t* idx, int32_t* compl_idx, float* scores, int B, int H, int N, int K,
                           tr_stream_t s) {
  TR_REQUIRE(cls_rows && idx, TR_ERR_NULL, "tr_cls_topk: null pointer");
  TR_REQUIRE(B > 0 && H > 0 && N >= 2 && N - 1 <= MAX_P, TR_ERR_SHAPE, "tr_cls_topk: need 2 <= N <= %d (N=%d)", MAX_P + 1, N);
  TR_REQUIRE(K >= 1 && K <= N - 1, TR_ERR_SHAPE, "tr_cls_topk: K=%d out of range for N=%d", K, N);
  if (N - 1 <= 256) hipLaunchKernelGGL(cls_topk_kernel<1>, dim3(B), dim3(256), 0, static_cast<hipStream_t>(s), cls_rows, idx, compl_idx, scores, H, N, K);
  else hipLaunchKernelGGL(cls_topk_kernel<4>, dim3(B), dim3(1024), 0, static_cast<hipStream_t>(s), cls_rows, idx, compl_idx, scores, H, N, K);
  TR_CHECK_LAUNCH("tr_cls_topk");
  return TR_OK;
}
