// Row-wise kernels on the fp32 residual stream (HBM-bound; one wave per token row, 16-byte accesses):
//   tr_layernorm_bf16         [x += pending bf16 residual;] nn.LayerNorm(eps=1e-6) -> bf16   (topk.py:86 norm1, :201 norm)
//   tr_gather_layernorm_bf16  Top-K gather/compact (topk.py:89-93) [+ EViT fused token evit.py:111-123]
//                             fused with norm2 (topk.py:95): the compacted residual stream and its
//                             normalised bf16 copy are produced in ONE pass over the kept rows.
//   tr_im2col_bf16            PatchEmbed unfold (timm PatchEmbed, call site topk.py:181)
//   tr_cls_pos_rows           cls_token + pos_embed[0] (topk.py:183-186)
#include "tr_common.h"
#include "tr_rowops.h"

namespace {

// x[row] (+= delta[row], written back) -> y[row] = LayerNorm(x[row]).  delta is the bf16 output of the preceding Linear
// (attn.proj or mlp.fc2): the residual add `x = x + drop_path(...)` (topk.py:87 / :95) is folded into the norm that
// reads x next, so the GEMMs never read-modify-write the fp32 stream.
// delta2 (bf16 path only): a SECOND pending residual, added after the first -- and write_x == 0: the sum is normalised but NOT
// written back.  Together they let the eval executor skip the stream write of a norm2 that no reduction follows: norm2 reads
// x + d_attn without storing it, the next norm1 reads x + d_attn + d_mlp (same fp32 additions in the same order: bit-identical) and
// writes the stream once -- 22 instead of 24 bytes per element and block.
template <bool F32, int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, long ldx, float* x_out, long ldxo, const void* __restrict__ delta, long ldd,
                                                        const uint16_t* __restrict__ delta2, long ldd2, int write_x,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        void* __restrict__ y, int M, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nchunks = D >> 2;
  const float* xr = x + (size_t)row * ldx;
  float* xo = x_out + (size_t)row * ldxo;            // == xr in the eval executor; a fresh tape slot in the training forward
  float4 v[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    v[c] = ln_nt_load4(xr + 4 * min(lane + 64 * c, nchunks - 1));     // branch-free: all loads of the row go out in one batch
  if (delta == nullptr && xo != xr && write_x) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (lane + 64 * c < nchunks) ln_nt_store4(v[c], xo + 4 * (lane + 64 * c));
  }
  if (delta != nullptr) {
    float4 d[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) d[c] = load_delta4<F32>(delta, (size_t)row * ldd + 4 * min(lane + 64 * c, nchunks - 1));
    if (!F32 && delta2 != nullptr) {
      float4 e[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) e[c] = load_delta4<false>(delta2, (size_t)row * ldd2 + 4 * min(lane + 64 * c, nchunks - 1));
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        v[c].x += d[c].x; v[c].y += d[c].y; v[c].z += d[c].z; v[c].w += d[c].w;
        d[c] = e[c];
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (lane + 64 * c < nchunks) {
        v[c].x += d[c].x; v[c].y += d[c].y; v[c].z += d[c].z; v[c].w += d[c].w;
        if (write_x) ln_nt_store4(v[c], xo + 4 * (lane + 64 * c));
      }
  }
  ln_row_store<F32, NCH>(v, nchunks, lane, D, eps, gamma, beta,
                    F32 ? (void*)(reinterpret_cast<float*>(y) + (size_t)row * D) : (void*)(reinterpret_cast<uint16_t*>(y) + (size_t)row * D));
}

// D = 128 * CPL <= 512 (DeiT-S: 384): HALF a wave per row, CPL float4 chunks per lane -- every lane busy (the one-wave-per-row
// kernel above leaves a quarter of the lanes idle at D = 384 and splits a row's 1.5 KiB into a full and a half request).
// Same arithmetic as ln_row_store (two-pass statistics, sums over the row's 32 lanes by xor-shuffles 16..1).
template <int CPL>
__global__ __launch_bounds__(256) void layernorm_half_kernel(const float* x, long ldx, float* x_out, long ldxo, const uint16_t* __restrict__ delta, long ldd,
                                                             const uint16_t* __restrict__ delta2, long ldd2, int write_x,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             uint16_t* __restrict__ y, int M, float eps) {
  constexpr int D = 128 * CPL;
  const int sub = threadIdx.x & 31;
  const int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 5);      // two rows per wave, any number of waves per workgroup
  if (row >= M) return;
  typedef __attribute__((ext_vector_type(4))) float f4;
  typedef __attribute__((ext_vector_type(2))) unsigned u2;
  const f4* xr = reinterpret_cast<const f4*>(x + (size_t)row * ldx);
  f4* xo = reinterpret_cast<f4*>(x_out + (size_t)row * ldxo);
  f4 v[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) v[c] = LN_LOAD(xr + sub + 32 * c);
  if (delta == nullptr && xo != xr && write_x) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) LN_STORE(v[c], xo + sub + 32 * c);
  }
  if (delta != nullptr) {
    const u2* dr = reinterpret_cast<const u2*>(delta + (size_t)row * ldd);
    u2 d[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) d[c] = LN_LOAD(dr + sub + 32 * c);
    if (delta2 != nullptr) {               // the second pending residual: (x + delta) + delta2
      const u2* er = reinterpret_cast<const u2*>(delta2 + (size_t)row * ldd2);
      u2 e[CPL];
#pragma unroll
      for (int c = 0; c < CPL; ++c) e[c] = LN_LOAD(er + sub + 32 * c);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        v[c][0] += __uint_as_float(d[c][0] << 16); v[c][1] += __uint_as_float(d[c][0] & 0xffff0000u);
        v[c][2] += __uint_as_float(d[c][1] << 16); v[c][3] += __uint_as_float(d[c][1] & 0xffff0000u);
        d[c] = e[c];
      }
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      v[c][0] += __uint_as_float(d[c][0] << 16); v[c][1] += __uint_as_float(d[c][0] & 0xffff0000u);
      v[c][2] += __uint_as_float(d[c][1] << 16); v[c][3] += __uint_as_float(d[c][1] & 0xffff0000u);
      if (write_x) LN_STORE(v[c], xo + sub + 32 * c);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CPL; ++c) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const float a = v[c][0] - mean, b = v[c][1] - mean, cc = v[c][2] - mean, d = v[c][3] - mean;
    q += (a * a + b * b) + (cc * cc + d * d);
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)D + eps);
  u2* yr = reinterpret_cast<u2*>(y + (size_t)row * D);
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = sub + 32 * c;
    const f4 g = *reinterpret_cast<const f4*>(gamma + 4 * ch);
    const f4 b = *reinterpret_cast<const f4*>(beta + 4 * ch);
    u2 pk;
    pk[0] = pack_bf16x2((v[c][0] - mean) * rstd * g[0] + b[0], (v[c][1] - mean) * rstd * g[1] + b[1]);
    pk[1] = pack_bf16x2((v[c][2] - mean) * rstd * g[2] + b[2], (v[c][3] - mean) * rstd * g[3] + b[3]);
    yr[ch] = pk;
  }
}

// grid: B * (ceil(rows/4) [+ 1]) blocks; wave w of a gather block handles output row r = 4*blk + w of image b.
template <bool F32, int NCH>
__global__ __launch_bounds__(256) void gather_layernorm_kernel(const float* __restrict__ x, const void* __restrict__ delta,
                                                               const int32_t* __restrict__ idx,
                                                               const int32_t* __restrict__ compl_idx,
                                                               const float* __restrict__ scores,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ x_out, void* __restrict__ y, int N, int K,
                                                               int N_out, int D, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool fuse = compl_idx != nullptr;
  const int grows = fuse ? N_out - 1 : N_out;          // rows that are plain gathers; EViT's fused token is row K+1 = grows
  const int rblocks = (grows + 3) >> 2;
  const int bpi = rblocks + (fuse ? 1 : 0);            // the fused token gets a block of its own (all four waves)
  const int b = blockIdx.x / bpi;
  const int lb = blockIdx.x % bpi;
  const bool fused_block = fuse && lb == rblocks;
  const int r = fused_block ? grows : lb * 4 + wave;
  if (!fused_block && r >= grows) return;
  const int nchunks = D >> 2;
  const int P = N - 1;
  const float* xb = x + (size_t)b * N * D;
  const bool has_d = delta != nullptr;                 // pending residual (proj output), same row layout as x
  const size_t dbase = (size_t)b * N * D;
  float4 v[NCH];
  __shared__ float4 part[3][64 * NCH];                 // partial sums of waves 1..3 of a fused block
  if (fused_block) {
    // EViT fused token: sum over the NOT-kept tokens, weighted by their (un-normalised) CLS attention
#pragma unroll
    for (int c = 0; c < NCH; ++c) v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int32_t* cb = compl_idx + (size_t)b * (P - K);
    const float* sb = scores + (size_t)b * P;
    // The four waves take the complement tokens j = wave, wave+4, ...; lane l of a wave holds index and weight of its token
    // 64*blk + l, the row loop reads them with v_readlane (wave-uniform) and fetches FOUR rows per step before accumulating.
    // (As one wave walking all tokens, three dependent round trips per token, this row was the kernel's long pole: 81 us per
    // launch against 27 us for the plain gather.)  Partials are combined in wave order: deterministic.
    const int n_c = P - K;
    const int n_w = (n_c - wave + 3) >> 2;                 // tokens of this wave
    for (int i0 = 0; i0 < n_w; i0 += 64) {
      int tl = 0;
      float wl = 0.f;
      if (i0 + lane < n_w) {
        tl = cb[wave + 4 * (i0 + lane)];
        wl = sb[tl];
      }
      const int cnt = min(64, n_w - i0);
      for (int j = 0; j < cnt; j += 4) {
        float4 a[4][NCH];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int jj = min(j + u, cnt - 1);                 // past the end: a valid row with weight 0
          const int t = __builtin_amdgcn_readlane(tl, jj);
          w[u] = (j + u < cnt) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wl), jj)) : 0.f;
          const float* xr = xb + (size_t)(1 + t) * D;
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const int ch = min(lane + 64 * c, nchunks - 1);                  // branch-free: the step's loads go out in one batch
            a[u][c] = *reinterpret_cast<const float4*>(xr + 4 * ch);
            if (has_d) {
              const float4 d = load_delta4<F32>(delta, dbase + (size_t)(1 + t) * D + 4 * ch);
              a[u][c].x += d.x; a[u][c].y += d.y; a[u][c].z += d.z; a[u][c].w += d.w;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < NCH; ++c)
            if (lane + 64 * c < nchunks) {
              v[c].x += a[u][c].x * w[u]; v[c].y += a[u][c].y * w[u]; v[c].z += a[u][c].z * w[u]; v[c].w += a[u][c].w * w[u];
            }
      }
    }
    if (wave > 0) {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (lane + 64 * c < nchunks) part[wave - 1][lane + 64 * c] = v[c];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (lane + 64 * c < nchunks) {
          const float4 q = part[p][lane + 64 * c];
          v[c].x += q.x; v[c].y += q.y; v[c].z += q.z; v[c].w += q.w;
        }
  } else {
    int src = r;
    if (idx != nullptr && r > 0) src = 1 + idx[(size_t)b * K + (r - 1)];
    const float* xr = xb + (size_t)src * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) v[c] = ln_nt_load4(xr + 4 * min(lane + 64 * c, nchunks - 1));   // branch-free: one batch
    if (has_d) {
      float4 d[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) d[c] = load_delta4<F32>(delta, dbase + (size_t)src * D + 4 * min(lane + 64 * c, nchunks - 1));
#pragma unroll
      for (int c = 0; c < NCH; ++c) { v[c].x += d[c].x; v[c].y += d[c].y; v[c].z += d[c].z; v[c].w += d[c].w; }
    }
  }
  const size_t orow = (size_t)b * N_out + r;
  if (x_out != nullptr) {
    float* xo = x_out + orow * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (lane + 64 * c < nchunks) ln_nt_store4(v[c], xo + 4 * (lane + 64 * c));
  }
  ln_row_store<F32, NCH>(v, nchunks, lane, D, eps, gamma, beta,
                    F32 ? (void*)(reinterpret_cast<float*>(y) + orow * D) : (void*)(reinterpret_cast<uint16_t*>(y) + orow * D));
}

// one thread = 8 consecutive pixels of one patch row -> one 16-byte bf16 store
template <bool F32>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, void* __restrict__ cols, int B, int C, int H,
                                                     int W, int patch, long total) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int per_row = patch >> 3;          // 8-pixel groups per patch row
  const int kcols = C * patch * patch;     // im2col row length
  const int groups_per_row = kcols >> 3;
  const long rowid = t / groups_per_row;   // (b, py, px)
  const int g = (int)(t - rowid * groups_per_row);
  const int gw = W / patch, gh = H / patch;
  const int b = (int)(rowid / (gh * gw));
  const int pp = (int)(rowid - (long)b * gh * gw);
  const int py = pp / gw, px = pp - py * gw;
  const int c = g / (patch * per_row);
  const int rem = g - c * patch * per_row;
  const int iy = rem / per_row, ixg = rem - iy * per_row;
  const float* src = img + (((size_t)b * C + c) * H + (size_t)py * patch + iy) * W + (size_t)px * patch + ixg * 8;
  const float4 a = *reinterpret_cast<const float4*>(src);
  const float4 d = *reinterpret_cast<const float4*>(src + 4);
  if (F32) {
    float* o = reinterpret_cast<float*>(cols) + rowid * kcols + (size_t)g * 8;
    *reinterpret_cast<float4*>(o) = a;
    *reinterpret_cast<float4*>(o + 4) = d;
  } else {
    uint4 pk;
    pk.x = pack_bf16x2(a.x, a.y); pk.y = pack_bf16x2(a.z, a.w);
    pk.z = pack_bf16x2(d.x, d.y); pk.w = pack_bf16x2(d.z, d.w);
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(cols) + rowid * kcols + (size_t)g * 8) = pk;
  }
}

__global__ __launch_bounds__(256) void cls_pos_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x,
                                                      int B, int N, int D) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= B * D) return;
  const int b = t / D, d = t - b * D;
  x[(size_t)b * N * D + d] = cls[d] + pos[d];
}

}  // namespace

static int layernorm_impl(bool f32, const float* x, long ldx, float* x_out, long ldxo, const void* delta, long ldd, const float* gamma,
                          const float* beta, void* y, int M, int D, float eps, tr_stream_t s, const uint16_t* delta2 = nullptr, long ldd2 = 0) {
  TR_REQUIRE(x && gamma && beta && y, TR_ERR_NULL, "tr_layernorm: null pointer");
  const int write_x = x_out != nullptr;           // tr_layernorm2_bf16: no stream write
  if (!write_x) { x_out = const_cast<float*>(x); ldxo = ldx; }
  TR_REQUIRE(ldxo % 4 == 0 && ldxo >= D && tr_aligned16(x_out), TR_ERR_SHAPE, "tr_layernorm: bad x_out stride %ld", ldxo);
  if (delta2) TR_REQUIRE(!f32 && delta && ldd2 % 4 == 0 && ldd2 >= D && ((uintptr_t)delta2 & 7u) == 0, TR_ERR_SHAPE,
                         "tr_layernorm: a second residual needs the first, the bf16 path and a stride %% 4 == 0 (ldd2=%ld)", ldd2);
  TR_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAX_CHUNKS && ldx % 4 == 0 && ldx >= D, TR_ERR_SHAPE,
             "tr_layernorm: need D %% 4 == 0, D <= 1024, ldx %% 4 == 0 (M=%d D=%d ldx=%ld)", M, D, ldx);
  if (delta) TR_REQUIRE(ldd % 4 == 0 && ldd >= D && ((uintptr_t)delta & 7u) == 0, TR_ERR_SHAPE, "tr_layernorm: bad delta stride %ld", ldd);
  TR_REQUIRE(tr_aligned16(x) && tr_aligned16(gamma) && tr_aligned16(beta) && tr_aligned16(y), TR_ERR_ALIGN,
             "tr_layernorm: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(s);
  tr_prof_note("layernorm_kernel", 0.0, (double)M * D * (f32 ? 4.0 : 2.0) + (double)M * D * 4.0 + (delta ? (double)M * D * (f32 ? 4.0 : 2.0) : 0.0) +
                                            (delta2 ? (double)M * D * 2.0 : 0.0) + ((delta && write_x) ? (double)M * D * 4.0 : 0.0));
#ifndef TR_LN_NO_HALF
  if (!f32 && D == 384) {
    static const int lnb = [] { const char* e = getenv("TR_LN_BLOCK"); const int b = e ? atoi(e) : 0; return (b == 64 || b == 128) ? b : 256; }();   // lab: waves per workgroup
    const int rpb = lnb / 32;
    hipLaunchKernelGGL(layernorm_half_kernel<3>, dim3((M + rpb - 1) / rpb), dim3(lnb), 0, st, x, ldx, x_out, ldxo, static_cast<const uint16_t*>(delta), ldd,
                       delta2, ldd2, write_x, gamma, beta, static_cast<uint16_t*>(y), M, eps);
    TR_CHECK_LAUNCH("tr_layernorm");
    return TR_OK;
  }
#endif
  if (f32) TR_DISPATCH_NCH(D, hipLaunchKernelGGL((layernorm_kernel<true, NCH>), dim3((M + 3) / 4), dim3(256), 0, st, x, ldx, x_out, ldxo, delta, ldd, delta2, ldd2, write_x, gamma, beta, y, M, D, eps));
  else TR_DISPATCH_NCH(D, hipLaunchKernelGGL((layernorm_kernel<false, NCH>), dim3((M + 3) / 4), dim3(256), 0, st, x, ldx, x_out, ldxo, delta, ldd, delta2, ldd2, write_x, gamma, beta, y, M, D, eps));
  TR_CHECK_LAUNCH("tr_layernorm");
  return TR_OK;
}

extern "C" int tr_layernorm_bf16(float* x, long ldx, const uint16_t* delta, long ldd, const float* gamma, const float* beta,
                                 uint16_t* y, int M, int D, float eps, tr_stream_t s) {
  return layernorm_impl(false, x, ldx, x, ldx, delta, ldd, gamma, beta, y, M, D, eps, s);
}
extern "C" int tr_layernorm_bf16_to(const float* x, long ldx, float* x_out, long ldxo, const uint16_t* delta, long ldd, const float* gamma,
                                    const float* beta, uint16_t* y, int M, int D, float eps, tr_stream_t s) {
  return layernorm_impl(false, x, ldx, x_out, ldxo, delta, ldd, gamma, beta, y, M, D, eps, s);
}
// y = LayerNorm(x + delta [+ delta2]) with the sum written to x_out -- or, x_out == NULL, not written at all (see layernorm_kernel)
extern "C" int tr_layernorm2_bf16(const float* x, long ldx, float* x_out, long ldxo, const uint16_t* delta, long ldd, const uint16_t* delta2,
                                  long ldd2, const float* gamma, const float* beta, uint16_t* y, int M, int D, float eps, tr_stream_t s) {
  TR_REQUIRE(delta != nullptr, TR_ERR_NULL, "tr_layernorm2_bf16: needs a pending residual");
  return layernorm_impl(false, x, ldx, x_out, ldxo, delta, ldd, gamma, beta, y, M, D, eps, s, delta2, ldd2);
}
extern "C" int tr_layernorm_f32(float* x, long ldx, const float* delta, long ldd, const float* gamma, const float* beta, float* y,
                                int M, int D, float eps, tr_stream_t s) {
  return layernorm_impl(true, x, ldx, x, ldx, delta, ldd, gamma, beta, y, M, D, eps, s);
}

static int gather_layernorm_impl(bool f32, const float* x, const void* delta, const int32_t* idx, const int32_t* compl_idx,
                                 const float* scores, const float* gamma, const float* beta, float* x_out, void* y, int B, int N,
                                 int K, int D, float eps, tr_stream_t s) {
  TR_REQUIRE(x && gamma && beta && y, TR_ERR_NULL, "tr_gather_layernorm: null pointer");
  TR_REQUIRE(B > 0 && N > 1 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAX_CHUNKS, TR_ERR_SHAPE,
             "tr_gather_layernorm: bad shape B=%d N=%d D=%d", B, N, D);
  int N_out = N;
  if (idx != nullptr) {
    TR_REQUIRE(K >= 1 && K <= N - 1, TR_ERR_SHAPE, "tr_gather_layernorm: K=%d out of range for N=%d", K, N);
    TR_REQUIRE(x_out != nullptr && x_out != x, TR_ERR_NULL, "tr_gather_layernorm: gather needs a distinct x_out");
    N_out = K + 1;
    if (compl_idx != nullptr) {
      TR_REQUIRE(scores != nullptr, TR_ERR_NULL, "tr_gather_layernorm: fuse needs scores");
      N_out = K + 2;
    }
  } else {
    TR_REQUIRE(compl_idx == nullptr, TR_ERR_SHAPE, "tr_gather_layernorm: compl_idx without idx");
  }
  TR_REQUIRE(tr_aligned16(x) && tr_aligned16(gamma) && tr_aligned16(beta) && tr_aligned16(y) && tr_aligned16(x_out) &&
                 tr_aligned16(delta),
             TR_ERR_ALIGN, "tr_gather_layernorm: pointers must be 16-byte aligned");
  const int rblocks = compl_idx != nullptr ? (N_out - 1 + 3) / 4 + 1 : (N_out + 3) / 4;   // + one block per image for the fused token
  tr_prof_note("gather_layernorm_kernel", 0.0, (double)B * N_out * D * ((delta ? 6.0 : 4.0) + (x_out ? 4.0 : 0.0) + (f32 ? 4.0 : 2.0)) +
                                                    (compl_idx ? (double)B * (N - 1 - K) * D * (delta ? 6.0 : 4.0) : 0.0));
  hipStream_t st = static_cast<hipStream_t>(s);
  if (f32)
    TR_DISPATCH_NCH(D, hipLaunchKernelGGL((gather_layernorm_kernel<true, NCH>), dim3(B * rblocks), dim3(256), 0, st, x, delta, idx, compl_idx,
                                          scores, gamma, beta, x_out, y, N, K, N_out, D, eps));
  else
    TR_DISPATCH_NCH(D, hipLaunchKernelGGL((gather_layernorm_kernel<false, NCH>), dim3(B * rblocks), dim3(256), 0, st, x, delta, idx, compl_idx,
                                          scores, gamma, beta, x_out, y, N, K, N_out, D, eps));
  TR_CHECK_LAUNCH("tr_gather_layernorm");
  return TR_OK;
}

// out = x + delta: the residual stream as the reference sees it after a block (x itself is only updated by the NEXT norm)
template <bool F32>
__global__ __launch_bounds__(256) void residual_snapshot_kernel(const float* __restrict__ x, const void* __restrict__ delta,
                                                                float* __restrict__ out, size_t nchunks) {
  const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= nchunks) return;
  float4 v = *reinterpret_cast<const float4*>(x + 4 * c);
  if (delta != nullptr) {
    const float4 d = load_delta4<F32>(delta, 4 * c);
    v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
  }
  *reinterpret_cast<float4*>(out + 4 * c) = v;
}

// dst[b][n] = src[n]: one per-block key mask for the whole batch (Heuristic family, heuristic.py:247-258)
__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int N) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < B * N) dst[e] = src[e % N];
}

extern "C" int tr_broadcast_rows(const float* src, float* dst, int B, int N, tr_stream_t s) {
  TR_REQUIRE(src && dst, TR_ERR_NULL, "tr_broadcast_rows: null pointer");
  TR_REQUIRE(B > 0 && N > 0, TR_ERR_SHAPE, "tr_broadcast_rows: bad shape B=%d N=%d", B, N);
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3((B * N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(s), src, dst, B, N);
  TR_CHECK_LAUNCH("tr_broadcast_rows");
  return TR_OK;
}

extern "C" int tr_residual_snapshot(const float* x, const void* delta, int delta_is_f32, float* out, size_t n, tr_stream_t s) {
  TR_REQUIRE(x && out, TR_ERR_NULL, "tr_residual_snapshot: null pointer");
  TR_REQUIRE(n > 0 && n % 4 == 0, TR_ERR_SHAPE, "tr_residual_snapshot: element count must be a positive multiple of 4");
  TR_REQUIRE(tr_aligned16(x) && tr_aligned16(out) && ((uintptr_t)delta & 7u) == 0, TR_ERR_ALIGN, "tr_residual_snapshot: misaligned pointer");
  hipStream_t st = static_cast<hipStream_t>(s);
  const size_t nch = n / 4;
  const unsigned nb = (unsigned)((nch + 255) / 256);
  if (delta_is_f32) hipLaunchKernelGGL(residual_snapshot_kernel<true>, dim3(nb), dim3(256), 0, st, x, delta, out, nch);
  else hipLaunchKernelGGL(residual_snapshot_kernel<false>, dim3(nb), dim3(256), 0, st, x, delta, out, nch);
  TR_CHECK_LAUNCH("tr_residual_snapshot");
  return TR_OK;
}

extern "C" int tr_gather_layernorm_bf16(const float* x, const uint16_t* delta, const int32_t* idx, const int32_t* compl_idx,
                                        const float* scores, const float* gamma, const float* beta, float* x_out, uint16_t* y, int B,
                                        int N, int K, int D, float eps, tr_stream_t s) {
  return gather_layernorm_impl(false, x, delta, idx, compl_idx, scores, gamma, beta, x_out, y, B, N, K, D, eps, s);
}
extern "C" int tr_gather_layernorm_f32(const float* x, const float* delta, const int32_t* idx, const int32_t* compl_idx,
                                       const float* scores, const float* gamma, const float* beta, float* x_out, float* y, int B, int N,
                                       int K, int D, float eps, tr_stream_t s) {
  return gather_layernorm_impl(true, x, delta, idx, compl_idx, scores, gamma, beta, x_out, y, B, N, K, D, eps, s);
}

static int im2col_impl(bool f32, const float* img, void* cols, int B, int C, int H, int W, int patch, tr_stream_t s) {
  TR_REQUIRE(img && cols, TR_ERR_NULL, "tr_im2col: null pointer");
  TR_REQUIRE(B > 0 && C > 0 && patch >= 8 && patch % 8 == 0 && H % patch == 0 && W % patch == 0, TR_ERR_SHAPE,
             "tr_im2col: need patch %% 8 == 0 and H,W multiples of patch (H=%d W=%d patch=%d)", H, W, patch);
  TR_REQUIRE(tr_aligned16(img) && tr_aligned16(cols), TR_ERR_ALIGN, "tr_im2col: pointers must be 16-byte aligned");
  const long total = (long)B * C * H * W / 8;
  tr_prof_note("im2col_kernel", 0.0, (double)B * C * H * W * (f32 ? 8.0 : 6.0));
  hipStream_t st = static_cast<hipStream_t>(s);
  if (f32) hipLaunchKernelGGL(im2col_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, img, cols, B, C, H, W, patch, total);
  else hipLaunchKernelGGL(im2col_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, img, cols, B, C, H, W, patch, total);
  TR_CHECK_LAUNCH("tr_im2col");
  return TR_OK;
}

extern "C" int tr_im2col_bf16(const float* img, uint16_t* cols, int B, int C, int H, int W, int patch, tr_stream_t s) {
  return im2col_impl(false, img, cols, B, C, H, W, patch, s);
}
extern "C" int tr_im2col_f32(const float* img, float* cols, int B, int C, int H, int W, int patch, tr_stream_t s) {
  return im2col_impl(true, img, cols, B, C, H, W, patch, s);
}

extern "C" int tr_cls_pos_rows(const float* cls_token, const float* pos_embed, float* x, int B, int N, int D, tr_stream_t s) {
  TR_REQUIRE(cls_token && pos_embed && x, TR_ERR_NULL, "tr_cls_pos_rows: null pointer");
  TR_REQUIRE(B > 0 && N > 0 && D > 0, TR_ERR_SHAPE, "tr_cls_pos_rows: bad shape");
  hipLaunchKernelGGL(cls_pos_kernel, dim3((B * D + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(s), cls_token, pos_embed, x,
                     B, N, D);
  TR_CHECK_LAUNCH("tr_cls_pos_rows");
  return TR_OK;
}
