// bf16 Linear layers on MFMA for gfx950:  out = epilogue(A[M,K] * W[N,K]^T + bias[N]).
//
// Replaces the nn.Linear calls of the reference's block (SURVEY.md 8a rows a1,a3,a4,a5):
//   qkv      topk.py:44     (TR_EPI_BF16)          proj     topk.py:52 + residual :87 (TR_EPI_RESID_F32)
//   mlp.fc1  timm Mlp + GELU (TR_EPI_GELU_BF16)    mlp.fc2  + residual topk.py:95     (TR_EPI_RESID_F32)
//   head     topk.py:203    (TR_EPI_F32)           patch_embed.proj topk.py:181-186   (TR_EPI_PATCH_F32)
//
// Design (cdna_hip_programming.md section 5): 128x128x64 tile, 256 threads = 4 waves (2x2), each wave 64x64 as 4x4
// v_mfma_f32_16x16x32_bf16.  Both operands are K-contiguous (nn.Linear stores W as [N,K]), so both fragments
// are plain 16-byte LDS reads.  W is the MFMA "A" operand and the activation the "B" operand: the accumulator
// then holds 4 CONSECUTIVE output columns per lane -> 8/16-byte epilogue stores.  LDS rows are 128 B with the
// 16-byte chunk index XOR-swizzled by (row>>1)&7: conflict-free for the ds_read_b128 lane groups and for the
// ds_write_b128 staging (tools/lds_sim.py).  Register-staged double buffering, one barrier per K-step:
// global loads of tile t+1 are issued before the MFMAs of tile t and written to LDS after them (T14).
// Block ids are remapped so every XCD walks a contiguous run of tiles that share the activation rows (T1).
#include "tr_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand tile

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ float gelu_erf(float x) {
  // 0.5 x (1 + erf(x/sqrt2)); erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below bf16 resolution)
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = 1.0f - poly * __expf(-z * z);
  const float erfv = x < 0.0f ? -e : e;
  return 0.5f * x * (1.0f + erfv);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                           const float* __restrict__ bias, void* __restrict__ outp,
                                                           const float* __restrict__ aux, int aux_i, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];  // [buf][A|W]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int nNt = (N + BN - 1) / BN;
  const int nMt = (M + BM - 1) / BM;
  const int logical = xcd_remap(blockIdx.x, nMt * nNt);
  const int m0 = (logical / nNt) * BM;
  const int n0 = (logical % nNt) * BN;

  // ---- staging assignment: 4 chunks of A and 4 of W per thread per K-step (chunk g = tid + 256*i)
  const int srow = tid >> 3, sc = tid & 7;
  const uint16_t* a_src0 = A + (size_t)min(m0 + srow, M - 1) * K + sc * 8;
  const uint16_t* a_src1 = A + (size_t)min(m0 + srow + 32, M - 1) * K + sc * 8;
  const uint16_t* a_src2 = A + (size_t)min(m0 + srow + 64, M - 1) * K + sc * 8;
  const uint16_t* a_src3 = A + (size_t)min(m0 + srow + 96, M - 1) * K + sc * 8;
  const uint16_t* w_src0 = W + (size_t)min(n0 + srow, N - 1) * K + sc * 8;
  const uint16_t* w_src1 = W + (size_t)min(n0 + srow + 32, N - 1) * K + sc * 8;
  const uint16_t* w_src2 = W + (size_t)min(n0 + srow + 64, N - 1) * K + sc * 8;
  const uint16_t* w_src3 = W + (size_t)min(n0 + srow + 96, N - 1) * K + sc * 8;
  // rows srow+32*i keep (row>>1)&7 == (srow>>1)&7, so all four chunks share one swizzled offset + 32 rows * 128 B
  const int lds_off0 = swz(srow, sc);
  uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
#define GLOAD(k0)                                              \
  do {                                                         \
    ra0 = *reinterpret_cast<const uint4*>(a_src0 + (k0));      \
    ra1 = *reinterpret_cast<const uint4*>(a_src1 + (k0));      \
    ra2 = *reinterpret_cast<const uint4*>(a_src2 + (k0));      \
    ra3 = *reinterpret_cast<const uint4*>(a_src3 + (k0));      \
    rw0 = *reinterpret_cast<const uint4*>(w_src0 + (k0));      \
    rw1 = *reinterpret_cast<const uint4*>(w_src1 + (k0));      \
    rw2 = *reinterpret_cast<const uint4*>(w_src2 + (k0));      \
    rw3 = *reinterpret_cast<const uint4*>(w_src3 + (k0));      \
  } while (0)
#define LSTORE(buf)                                                          \
  do {                                                                       \
    unsigned char* sa_ = smem + (buf) * 2 * TILE_BYTES + lds_off0;           \
    unsigned char* sw_ = sa_ + TILE_BYTES;                                   \
    *reinterpret_cast<uint4*>(sa_) = ra0;                                    \
    *reinterpret_cast<uint4*>(sa_ + 32 * 128) = ra1;                         \
    *reinterpret_cast<uint4*>(sa_ + 64 * 128) = ra2;                         \
    *reinterpret_cast<uint4*>(sa_ + 96 * 128) = ra3;                         \
    *reinterpret_cast<uint4*>(sw_) = rw0;                                    \
    *reinterpret_cast<uint4*>(sw_ + 32 * 128) = rw1;                         \
    *reinterpret_cast<uint4*>(sw_ + 64 * 128) = rw2;                         \
    *reinterpret_cast<uint4*>(sw_ + 96 * 128) = rw3;                         \
  } while (0)

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const int nk = K / BK;

  GLOAD(0);
  LSTORE(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) GLOAD((kt + 1) * BK);
    const unsigned char* sa = smem + buf * 2 * TILE_BYTES;
    const unsigned char* sw = sa + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[4], af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wf[i] = *reinterpret_cast<const bf16x8*>(sw + swz(wn * 64 + i * 16 + frow, 4 * ks + fq));
        af[i] = *reinterpret_cast<const bf16x8*>(sa + swz(wm * 64 + i * 16 + frow, 4 * ks + fq));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) LSTORE(buf ^ 1);
    __syncthreads();
  }

#undef GLOAD
#undef LSTORE
  // ---- epilogue: lane holds D[n = 4*fq + r][m = frow] of each 16x16 tile -> 4 consecutive output columns.
  // All loads (bias, residual / pos_embed) are issued UNCONDITIONALLY from clamped addresses before any use, so the
  // 16 read-modify-write groups overlap instead of paying one dependent HBM round trip each; stores are predicated.
  float4 bv[4];
  int ncol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ncol[i] = n0 + wn * 64 + i * 16 + 4 * fq;
    bv[i] = *reinterpret_cast<const float4*>(bias + min(ncol[i], N - 4));
  }
  size_t orow[4];
  float4 rv[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int mc = min(m0 + wm * 64 + j * 16 + frow, M - 1);
    orow[j] = (size_t)mc;
    const float* posrow = nullptr;
    if (EPI == TR_EPI_PATCH_F32) {
      const int b = mc / aux_i, p = mc - b * aux_i;
      orow[j] = (size_t)b * (aux_i + 1) + 1 + p;
      posrow = aux + (size_t)(1 + p) * N;
    }
    if (EPI == TR_EPI_RESID_F32 || EPI == TR_EPI_PATCH_F32) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int nc = min(ncol[i], N - 4);
        rv[j][i] = (EPI == TR_EPI_RESID_F32)
                       ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(outp) + orow[j] * N + nc)
                       : *reinterpret_cast<const float4*>(posrow + nc);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + frow;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = ncol[i];
      float v0 = acc[i][j][0] + bv[i].x, v1 = acc[i][j][1] + bv[i].y, v2 = acc[i][j][2] + bv[i].z, v3 = acc[i][j][3] + bv[i].w;
      const bool ok = (m < M) && (n < N);
      if (EPI == TR_EPI_BF16 || EPI == TR_EPI_GELU_BF16) {
        if (EPI == TR_EPI_GELU_BF16) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
        uint2 pk;
        pk.x = pack_bf16x2(v0, v1);
        pk.y = pack_bf16x2(v2, v3);
        if (ok) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(outp) + orow[j] * N + n) = pk;
      } else {
        if (EPI == TR_EPI_RESID_F32 || EPI == TR_EPI_PATCH_F32) {
          v0 += rv[j][i].x; v1 += rv[j][i].y; v2 += rv[j][i].z; v3 += rv[j][i].w;
        }
        if (ok) *reinterpret_cast<float4*>(reinterpret_cast<float*>(outp) + orow[j] * N + n) = make_float4(v0, v1, v2, v3);
      }
    }
  }
}

}  // namespace

extern "C" int tr_gemm_bf16(const uint16_t* A, const uint16_t* W, const float* bias, void* out, const float* aux,
                            int aux_i, int M, int N, int K, int epilogue, tr_stream_t s) {
  TR_REQUIRE(A && W && bias && out, TR_ERR_NULL, "tr_gemm_bf16: null pointer");
  TR_REQUIRE(M > 0 && N > 0 && K > 0, TR_ERR_SHAPE, "tr_gemm_bf16: M,N,K must be positive (got %d,%d,%d)", M, N, K);
  TR_REQUIRE(K % BK == 0, TR_ERR_SHAPE, "tr_gemm_bf16: K=%d must be a multiple of %d", K, BK);
  TR_REQUIRE(N % 4 == 0, TR_ERR_SHAPE, "tr_gemm_bf16: N=%d must be a multiple of 4", N);
  TR_REQUIRE(tr_aligned16(A) && tr_aligned16(W) && tr_aligned16(bias) && tr_aligned16(out), TR_ERR_ALIGN,
             "tr_gemm_bf16: pointers must be 16-byte aligned");
  if (epilogue == TR_EPI_PATCH_F32)
    TR_REQUIRE(aux && aux_i > 0 && M % aux_i == 0 && tr_aligned16(aux), TR_ERR_SHAPE,
               "tr_gemm_bf16: PATCH epilogue needs pos_embed and P | M");
  const int nblocks = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  hipStream_t st = static_cast<hipStream_t>(s);
  dim3 grid(nblocks), block(256);
  switch (epilogue) {
    case TR_EPI_BF16: hipLaunchKernelGGL(gemm_bf16_kernel<TR_EPI_BF16>, grid, block, 0, st, A, W, bias, out, aux, aux_i, M, N, K); break;
    case TR_EPI_GELU_BF16: hipLaunchKernelGGL(gemm_bf16_kernel<TR_EPI_GELU_BF16>, grid, block, 0, st, A, W, bias, out, aux, aux_i, M, N, K); break;
    case TR_EPI_RESID_F32: hipLaunchKernelGGL(gemm_bf16_kernel<TR_EPI_RESID_F32>, grid, block, 0, st, A, W, bias, out, aux, aux_i, M, N, K); break;
    case TR_EPI_F32: hipLaunchKernelGGL(gemm_bf16_kernel<TR_EPI_F32>, grid, block, 0, st, A, W, bias, out, aux, aux_i, M, N, K); break;
    case TR_EPI_PATCH_F32: hipLaunchKernelGGL(gemm_bf16_kernel<TR_EPI_PATCH_F32>, grid, block, 0, st, A, W, bias, out, aux, aux_i, M, N, K); break;
    default: TR_REQUIRE(false, TR_ERR_SHAPE, "tr_gemm_bf16: unknown epilogue %d", epilogue);
  }
  TR_CHECK_LAUNCH("tr_gemm_bf16");
  return TR_OK;
}
