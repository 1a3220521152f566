// PatchEmbed + cls_token + pos_embed in ONE launch (timm PatchEmbed, call sites topk.py:181-186; deit_viz.py the same):
//   x[b, 0, :]     = cls_token + pos_embed[0]
//   x[b, 1 + p, :] = W . patch(b, p) + bias + pos_embed[1 + p]          W = proj.weight viewed [D, C*16*16]
// replacing im2col (fp32 image -> bf16 [B*P, 768] columns: 154 MB read + 77 MB written at batch 256) + GEMM (77 MB read again) +
// the cls/pos kernel of the eval forward.  The unfold happens on the way INTO the LDS: a K-step of 64 is four 64-byte runs of one
// image row segment per patch, loaded fp32 into registers, rounded to bf16 and written into the swizzled [rows][64] operand image.
//
// One workgroup (8 waves) per (image, chunk of <= 208 patches, 384 output columns): the image's patches are the tile's rows, so the
// fp32 image is read from HBM exactly once (DeiT-B: once per 384-column half, the second from L2), and the launch has one workgroup per CU
// at batch 256.  Wave w accumulates all 13 row blocks x the column blocks {w, w+8, w+16} (156 accumulator registers): LDS reads per MFMA
// 16 : 39.  W K-slabs arrive by LDS-DMA, double-buffered like A.  The kernel is HBM-bound by design (231 MB per launch at batch 256
// against 29.6 GFLOP: 12 us of matrix time), so what matters is bytes in flight: every thread holds the next K-step's 7 x 16 B.
// Epilogue: three passes over 128 output columns through an LDS staging image, so that every store is a full 128-byte line.
// Summation order: (sum of products + bias) + pos_embed -- the three-launch path adds pos_embed after its first K-step, so the two
// differ in the last bit; the eval executor therefore takes this kernel for EVERY batch size of a supported shape (an image's tokens must
// not depend on its batch), although below ~half a chip of workgroups the three launches are faster (batch 64 DeiT-S: 45 vs 35 us).
#include "tr_common.h"

namespace {

constexpr int PE_RB = 13;                   // 16-row blocks per chunk (196 patches of a 224^2 image -> one chunk)
constexpr int PE_ROWS = PE_RB * 16;
constexpr int PE_NC = 384;                  // output columns per workgroup
constexpr int PE_A_BYTES = PE_ROWS * 128;   // one A slot: [208][64 bf16]
constexpr int PE_W_BYTES = PE_NC * 128;     // one W slot: [384][64 bf16]
constexpr int PE_STAGE_LD = 528;            // epilogue staging row stride (128 fp32 + 16 B)
constexpr int PE_LDS = 2 * (PE_A_BYTES + PE_W_BYTES);
static_assert(PE_LDS >= PE_ROWS * PE_STAGE_LD, "the epilogue staging image reuses the operand ring");
constexpr int PE_ITEMS = 7;                 // float4 loads per thread and K-step (208 rows x 16 / 512 = 6.5)

__device__ __forceinline__ int pswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }      // W slabs (written by LDS-DMA)
// The A image (fp32 pixels -> bf16, written by ds_write_b64) has a swizzle of its own.  A 16-lane store group holds four consecutive rows x
// the four float4 of a 16-pixel run, i.e. the 32-byte chunk PAIR {2 r, 2 r + 1} of four rows: with the ring's (row >> 1) & 7 those four rows
// land on one pair of the 128-byte bank window -- 4-way conflicts on every store (27.7 % of the kernel's LDS cycles, r05 SQ counters;
// tools/lds_sim.py: 16 cycles per wave-instruction, 4 without).  XOR 2 (row & 3) moves the four rows to four pairs; the per-quad term
// keeps the fragment reads (16 rows x 4 chunks per ds_read_b128, banks mod 64) conflict-free.  Same bits out: only where a value sits.
__device__ __forceinline__ int aswz(int row, int chunk) {
  return row * 128 + ((chunk ^ ((2 * (row & 3)) ^ ((0x1320 >> (4 * ((row >> 2) & 3))) & 3))) << 4);      // quad term {0, 2, 3, 1}
}

__device__ __forceinline__ void dma_piece(const uint16_t* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %[ld]\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o], %[b]"
      :
      : [o] "v"(voff), [b] "s"(sbase), [ld] "s"(lds_dst)
      : "memory", "m0");
}

__global__ __launch_bounds__(512, 1) void patch_embed_kernel(const float* __restrict__ img, const uint16_t* __restrict__ W,
                                                             const float* __restrict__ bias, const float* __restrict__ cls,
                                                             const float* __restrict__ pos, float* __restrict__ x, int C, int HW, int gw,
                                                             int P, int D, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / nchunk, chunk = blockIdx.x - b * nchunk;
  const int nb = blockIdx.y;                                   // 384-column slab of the output
  const int p0 = chunk * PE_ROWS, rows = min(PE_ROWS, P - p0);
  const int K = C * 256, nk = K / 64;
  unsigned char* sA = smem;
  unsigned char* sW = smem + 2 * PE_A_BYTES;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // ---- per-thread source offsets (floats, without the channel / row-group term) and LDS destinations of the A items
  // item e = tid + 512 it  ->  q = e & 3 (float4 of the 16-pixel run), m = (e >> 2) % rows_pad, r = (e >> 2) / rows_pad (image row of the K-step)
  // ordered (r, m, q): consecutive threads walk along an image row (patches px, px+1, ...: contiguous pixels)
  unsigned aoff[PE_ITEMS];
  int adst[PE_ITEMS];
  const float* ibase = img + (size_t)b * C * HW * HW;
#pragma unroll
  for (int it = 0; it < PE_ITEMS; ++it) {
    const int e = tid + 512 * it;
    const int q = e & 3, mr = e >> 2;
    const int r = mr / PE_ROWS, m = mr - r * PE_ROWS;          // r in 0..3 for e < 4 * 208 * 4 = 3328 (it = 6: e < 3584 -> r may reach 4: masked)
    const bool live = r < 4 && m < rows;
    const int p = min(p0 + m, P - 1);
    const int py = p / gw, px = p - py * gw;
    aoff[it] = (unsigned)((py * 16 + min(r, 3)) * HW + px * 16 + q * 4);
    adst[it] = live ? aswz(m, 2 * r + (q >> 1)) + (q & 1) * 8 : -1;
  }
  // rows of the chunk past the last patch: zero operand rows in both slots (never written again)
  for (int i = tid; i < (PE_ROWS - rows) * 8 * 2; i += 512) {
    const int slot = i & 1, c = (i >> 1) & 7, m = rows + (i >> 4);
    *reinterpret_cast<uint4*>(sA + slot * PE_A_BYTES + m * 128 + c * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  // W pieces of this wave: rows 48 w + 8 j + l3 of the slab, LDS position (row, pc) holds logical chunk pc ^ ((row >> 1) & 7)
  unsigned woff[6];
  const int l3 = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int row = wave * 48 + j * 8 + l3;
    woff[j] = (unsigned)(((nb * PE_NC + row) * K + ((pc ^ ((row >> 1) & 7)) << 3)) * 2);
  }
  const unsigned wdst = lds0 + 2 * PE_A_BYTES + wave * 48 * 128;

  f32x4 areg[PE_ITEMS];
  auto issue = [&](int kt, int slot) {
#pragma unroll
    for (int j = 0; j < 6; ++j) dma_piece(W, woff[j] + (unsigned)kt * 128u, wdst + slot * PE_W_BYTES + j * 1024);
    const float* src = ibase + (size_t)(kt >> 2) * HW * HW + (size_t)((kt & 3) * 4) * HW;
#pragma unroll
    for (int it = 0; it < PE_ITEMS; ++it)      // nontemporal: the image is read once -- kept out of the caches the forward lives in (-12 us per forward)
      areg[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + aoff[it]));
  };
  auto commit = [&](int slot) {
#pragma unroll
    for (int it = 0; it < PE_ITEMS; ++it)
      if (adst[it] >= 0) {
        uint2 v;
        v.x = pack_bf16x2(areg[it][0], areg[it][1]);
        v.y = pack_bf16x2(areg[it][2], areg[it][3]);
        *reinterpret_cast<uint2*>(sA + slot * PE_A_BYTES + adst[it]) = v;
      }
  };

  f32x4 acc[PE_RB][3];
#pragma unroll
  for (int i = 0; i < PE_RB; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0, 0);
  commit(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int slot = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, slot ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* a = sA + slot * PE_A_BYTES;
    const unsigned char* w = sW + slot * PE_W_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(w + pswz(16 * (wave + 8 * j) + li, 4 * ks + g));
#pragma unroll
      for (int i = 0; i < PE_RB; ++i) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(a + aswz(16 * i + li, 4 * ks + g));
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af, acc[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) commit(slot ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the W slab of the next step (LDS-DMA: invisible to the compiler's counters)
    __syncthreads();
  }

  // ---- epilogue: pass j stages the columns [128 j, 128 j + 128) of the slab -- wave w holds its columns 16 w .. 16 w + 15
  float* xrow0 = x + ((size_t)b * (P + 1) + 1 + p0) * D + nb * PE_NC;
  const int ch = tid & 31;                                       // this thread's 16-byte column group of a staged row (512 % 32 == 0)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int i = 0; i < PE_RB; ++i)
      *reinterpret_cast<f32x4*>(smem + (16 * i + li) * PE_STAGE_LD + 64 * wave + 16 * g) = acc[i][j];
    __syncthreads();
    const int col = nb * PE_NC + 128 * j + 4 * ch;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
    for (int it = 0; it < PE_RB; ++it) {
      const int m = (tid >> 5) + 16 * it;
      if (m < rows) {
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + m * PE_STAGE_LD + 16 * ch);
        const f32x4 pv = *reinterpret_cast<const f32x4*>(pos + (size_t)(1 + p0 + m) * D + col);
        v = v + bv + pv;
        *reinterpret_cast<f32x4*>(xrow0 + (size_t)m * D + 128 * j + 4 * ch) = v;
      }
    }
    __syncthreads();
  }
  if (chunk == 0 && tid < PE_NC / 4) {                           // the CLS row of this image
    const int col = nb * PE_NC + 4 * tid;
    const f32x4 c = *reinterpret_cast<const f32x4*>(cls + col), pv = *reinterpret_cast<const f32x4*>(pos + col);
    *reinterpret_cast<f32x4*>(x + (size_t)b * (P + 1) * D + col) = c + pv;
  }
}

}  // namespace

// 1 when tr_patch_embed_bf16 takes this shape (else the executor runs im2col + GEMM + cls/pos)
extern "C" int tr_patch_embed_supported(int C, int HW, int patch, int D) {
  return patch == 16 && HW % 16 == 0 && C >= 1 && D % PE_NC == 0 && (size_t)C * HW * HW < ((size_t)1 << 30);
}

extern "C" int tr_patch_embed_bf16(const float* img, const uint16_t* W, const float* bias, const float* cls, const float* pos, float* x, int B,
                                   int C, int HW, int patch, int D, tr_stream_t s) {
  TR_REQUIRE(img && W && bias && cls && pos && x, TR_ERR_NULL, "tr_patch_embed_bf16: null pointer");
  TR_REQUIRE(B > 0 && tr_patch_embed_supported(C, HW, patch, D), TR_ERR_SHAPE,
             "tr_patch_embed_bf16: need patch 16, H = W a multiple of 16, embed_dim a multiple of %d (C=%d HW=%d patch=%d D=%d)", PE_NC, C, HW, patch, D);
  TR_REQUIRE(tr_aligned16(img) && tr_aligned16(W) && tr_aligned16(bias) && tr_aligned16(cls) && tr_aligned16(pos) && tr_aligned16(x), TR_ERR_ALIGN,
             "tr_patch_embed_bf16: pointers must be 16-byte aligned");
  const int gw = HW / 16, P = gw * gw, nchunk = (P + PE_ROWS - 1) / PE_ROWS;
  TR_REQUIRE((size_t)D * C * 256 * 2 < ((size_t)1 << 32), TR_ERR_SHAPE, "tr_patch_embed_bf16: weight beyond the 32-bit offset range");
  hipStream_t st = static_cast<hipStream_t>(s);
  TR_RESERVE_LDS(reinterpret_cast<const void*>(patch_embed_kernel), PE_LDS, "tr_patch_embed_bf16");
  tr_prof_note("patch_embed_kernel", 2.0 * B * P * (double)D * C * 256, (double)B * C * HW * HW * 4.0 + (double)B * (P + 1) * D * 4.0);
  hipLaunchKernelGGL(patch_embed_kernel, dim3(B * nchunk, D / PE_NC), dim3(512), PE_LDS, st, img, W, bias, cls, pos, x, C, HW, gw, P, D, nchunk);
  TR_CHECK_LAUNCH("tr_patch_embed_bf16");
  return TR_OK;
}
