// Fused eval MLP for gfx950:  out = fc2(gelu(fc1(x)))  with the hidden activation never leaving the CU.
//
// The timm Mlp of a block as the reference calls it (topk.py:78 construction, :95 `x = x + self.mlp(self.norm2(x))`; the residual add stays
// in the next LayerNorm kernel) in ONE launch instead of tr_gemm_bf16(TR_EPI_GELU_BF16) + tr_gemm_bf16(TR_EPI_BF16) with a [M, 4D] bf16 round
// trip through HBM between them.  Bit-identical to that pair by construction: same MFMA (16x16x32 bf16), same operand-to-lane maps,
// accumulators that start at the bias, K walked in the same 32-deep steps in the same order, the same GELU fit on the fp32 accumulator
// (evaluated with scalar instead of packed VALU instructions: same operations, same contractions), the hidden rounded to bf16 at the same
// point.  tests/test_hip_ops.py::test_mlp_fused_* compares them bit for bit, tools/mlp_lab.py --stress screens for races.
//
// STATUS (round 5): per 32-hidden-unit step as fast as the pair, not faster (the fc1 + GELU wave's instruction stream); with the stream-K
// schedule (MfSeq below) it ties the pair at 1.5 rounds of blocks and wins beyond (70,001 rows: 176 vs 217-238 us, 938 TFLOP/s), and in the
// model it saves the hidden activation's round trip: headline forward -2 %.  The executor takes it where tr_mlp_fused_wanted says so.
// A second instantiation (RL) also finishes the block -- residual add + the next block's norm1 -- and is OFF by default: slower in the
// model.  Where the cycles go, stamps and ablations of every version: profiles/r05_mlp_lab.md.
//
// Structure (one persistent 512-thread workgroup per CU, 128 token rows per block, D = 384):
//   * Waves 0-3 ("P", one per SIMD) own 32 token rows each and keep their x rows IN REGISTERS as the MFMA B operand (96 VGPRs).
//     Per 32 hidden units (one "step") they compute h^T = W1_step . x^T (48 MFMAs), apply GELU (one step later, sliced between the next
//     step's MFMAs), round to bf16 -- and because the W1 rows of a step are fed in the order  tile t2, row r -> hidden 8(r>>2) + 4 t2 + (r&3),
//     a lane's eight results ARE the B-operand fragment of fc2's MFMA over those 32 hidden units (the S^T = K.Q^T trick of the attention
//     kernel).  2 KB per wave and step go through LDS to the partner wave.
//   * Waves 4-7 ("C", the SIMD partners) own the same 32 token rows and keep the whole 32 x 384 fp32 output accumulator in registers
//     (192 VGPRs); per step 48 MFMAs  acc += W2[:, step] . h.  One epilogue per 128 x 384 x 1536 x 4 FLOP instead of one per tile.
//   * Weights come from a FRAGMENT-MAJOR packed copy (tr_mlp_pack_bf16: per step 24 KiB of W1 fragments + 24 KiB of W2 fragments, each
//     fragment 1 KiB in lane order), so a DMA piece is 1 KiB contiguous -> LDS lane-linear, and every fragment read is a linear
//     conflict-free ds_read_b128.  3-slot ring of 48-KiB entries (entry t = W1 of step t | W2 of step t-2: what time step t consumes),
//     two entries in flight behind counted vmcnt waits, one s_barrier per step, placed BEFORE the step's last window of MFMAs (their
//     fragments are in registers), so the matrix pipe has work while the next step's first fragments arrive.
//   * FLOP per byte through the CU's L2 -> LDS feed: 128 (a 256 x 128 GEMM tile: 85); HBM traffic per 128 rows: 96 KB in, 96 KB out.
#include "tr_common.h"

// {shader cycles, 100-MHz ticks, launches} of mlp_fused_kernel workgroup 8 (its fc1 wave's step loop) since the last read: -DTR_DIAG_CLOCK builds only
__device__ unsigned long long tr_mlp_clock_probe[3];

namespace {

// lab switches (tools/lab/build_variant.sh): TR_ABLATE_NO_DMA (no weight stream), TR_ABLATE_NO_GELU (identity activation), TR_ABLATE_NO_MFMA
#ifdef TR_ABLATE_NO_MFMA
#define MF_MFMA(a, b, c) ([&] { asm volatile("" ::"v"(a), "v"(b)); return c; }())
#else
#define MF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
// gelu2() of tr_common.h element by element: the same operations with the same contractions (bit-identical), but as scalar VALU
// instructions -- packed f32 arithmetic takes two passes and, beside MFMAs, costs more than the two scalar instructions it replaces
__device__ __forceinline__ float mf_gelu1(float x) {
  constexpr float L2E = 1.44269504088896340736f;
  constexpr float C1 = -1.59501577f * L2E, C3 = -7.40112920e-02f * L2E, C5 = 7.03033576e-04f * L2E;
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
  const float x2 = xc * xc;
  float p = __builtin_fmaf(x2, C5, C3);
  p = __builtin_fmaf(p, x2, C1);
  const float z = p * xc;
  const float e = __builtin_amdgcn_exp2f(z) + 1.0f;
  return x * __builtin_amdgcn_rcpf(e);
}
#if defined(TR_ABLATE_NO_GELU)
__device__ __forceinline__ f32x2 mf_gelu2(f32x2 v) { return v; }
#elif defined(MF_GELU_PACKED)
__device__ __forceinline__ f32x2 mf_gelu2(f32x2 v) { return gelu2(v); }
#else
__device__ __forceinline__ f32x2 mf_gelu2(f32x2 v) {
  float a = mf_gelu1(v[0]), b = mf_gelu1(v[1]);
  asm volatile("" : "+v"(a), "+v"(b));         // keeps the SLP vectoriser from re-packing the two chains
  return f32x2{a, b};
}
#endif

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// diagnostic build only (never in the product): s_memtime stamps of workgroup 8, P wave 0 and C wave 4, first 64 steps, written BEHIND the
// output (the lab allocates the room: tools/mlp_lab.py --stamps)
// lab ablations of the residual + LayerNorm epilogue: -DMF_ABL_NO_XSTORE (no stream stores), -DMF_ABL_NO_NSTORE (no norm output stores)
#ifdef MF_ABL_NO_XSTORE
#define MF_XSTORE(x) asm volatile("" ::"v"(v))
#else
#define MF_XSTORE(x) x
#endif
#ifdef MF_ABL_NO_NSTORE
#define MF_NSTORE(x) do { } while (0)
#else
#define MF_NSTORE(x) x
#endif
#ifdef TR_DIAG_STAMPS
#define MF_STAMP(k) ts_[k] = __builtin_amdgcn_s_memtime()
#define MF_STAMP_DECL unsigned long long ts_[4] = {0, 0, 0, 0}
#define MF_STAMP_DUMP(role, step)                                                                                                     \
  do {                                                                                                                                \
    if (bid == 8 && pr == 0 && lane == 0 && (step) < 64) {                                                                            \
      unsigned long long* st_ = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(outp) + out_bytes) + ((role) * 64 + (step)) * 4; \
      st_[0] = ts_[0]; st_[1] = ts_[1]; st_[2] = ts_[2]; st_[3] = ts_[3];                                                             \
    }                                                                                                                                 \
  } while (0)
// epilogue of the residual + LayerNorm variant: eight stamps of workgroup 8's C wave 4 into rows 60, 61 of the P role's table
#define MF_ESTAMP(k) es_[k] = __builtin_amdgcn_s_memtime()
#define MF_ESTAMP_DECL unsigned long long es_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define MF_ESTAMP_DUMP                                                                                                                \
  do {                                                                                                                                \
    if (bid == 8 && pr == 0 && lane == 0) {                                                                                           \
      unsigned long long* st_ = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(outp) + out_bytes) + 60 * 4;   \
      for (int k_ = 0; k_ < 8; ++k_) st_[k_] = es_[k_];                                                                               \
    }                                                                                                                                 \
  } while (0)
#define MF_CLOCK_BEGIN const unsigned long long ck0_ = __builtin_amdgcn_s_memtime(), rt0_ = __builtin_amdgcn_s_memrealtime()
#define MF_CLOCK_END                                                                                                                  \
  do {                                                                                                                                \
    if (bid == 8 && tid == 0) {                                                                                                       \
      unsigned long long* st_ = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(outp) + out_bytes) + 512;      \
      st_[0] = __builtin_amdgcn_s_memtime() - ck0_;                                                                                   \
      st_[1] = __builtin_amdgcn_s_memrealtime() - rt0_;                                                                               \
    }                                                                                                                                 \
  } while (0)
#elif defined(TR_DIAG_CLOCK)
// (tools/lab/clock_probe.py: the same stamps summed into a device symbol -- no buffer is touched, so this build runs inside the model)
#define MF_CLOCK_BEGIN const unsigned long long ck0_ = __builtin_amdgcn_s_memtime(), rt0_ = __builtin_amdgcn_s_memrealtime()
#define MF_CLOCK_END                                                                                                                  \
  do {                                                                                                                                \
    if (bid == 8 && tid == 0) {                                                                                                       \
      atomicAdd(&tr_mlp_clock_probe[0], __builtin_amdgcn_s_memtime() - ck0_);                                                         \
      atomicAdd(&tr_mlp_clock_probe[1], __builtin_amdgcn_s_memrealtime() - rt0_);                                                     \
      atomicAdd(&tr_mlp_clock_probe[2], 1ull);                                                                                        \
    }                                                                                                                                 \
  } while (0)
#define MF_STAMP(k) do { } while (0)
#define MF_STAMP_DECL do { } while (0)
#define MF_STAMP_DUMP(role, step) do { } while (0)
#define MF_ESTAMP(k) do { } while (0)
#define MF_ESTAMP_DECL do { } while (0)
#define MF_ESTAMP_DUMP do { } while (0)
#else
#define MF_CLOCK_BEGIN do { } while (0)
#define MF_CLOCK_END do { } while (0)
#define MF_STAMP(k) do { } while (0)
#define MF_STAMP_DECL do { } while (0)
#define MF_STAMP_DUMP(role, step) do { } while (0)
#define MF_ESTAMP(k) do { } while (0)
#define MF_ESTAMP_DECL do { } while (0)
#define MF_ESTAMP_DUMP do { } while (0)
#endif

constexpr int MF_D = 384;                      // embed dim this instantiation serves
constexpr int MF_KS = MF_D / 32;               // 12: 32-deep k-steps of fc1
constexpr int MF_NI = MF_D / 16;               // 24: 16-column groups of fc2's output
constexpr int MF_W1FR = 2 * MF_KS;             // 24 W1 fragments per step (2 row tiles x 12 k-steps)
constexpr int MF_FR = MF_W1FR + MF_NI;         // 48 fragments = DMA pieces per entry
constexpr int MF_ENTRY = MF_FR * 1024;         // 48 KiB
constexpr int MF_NSLOT = 3;
constexpr int MF_HBUF = 2048;                  // one hidden fragment pair (2 token groups x 64 lanes x 16 B)
constexpr int MF_LDS = MF_NSLOT * MF_ENTRY + 4 * 2 * MF_HBUF;     // 163,840 B: all of the CU's LDS
constexpr int MF_ROWS = 128;                   // token rows per block (4 wave pairs x 32)
#ifndef MF_PQ
#define MF_PQ 0                                // DMA pieces of an entry a P wave issues (0..6, lab: -DMF_PQ=..); a C wave issues 12 - MF_PQ.  0: the P
#endif                                         // wave's instruction stream (GELU) is the step's critical path -- profiles/r05_mlp_lab.md
#ifndef MF_DMA_BURST
#define MF_DMA_SPREAD 1                        // a C wave's pieces one at a time between MFMA pairs (lab: -DMF_DMA_BURST issues them window by window)
#endif
constexpr int MF_PPW = (MF_PQ + 3) / 4;        // ... behind the MFMAs of a P step's windows 1..4 (window 0 carries the bias loads)
[[maybe_unused]] constexpr int MF_CPW = (12 - MF_PQ + 4) / 5;   // ... and of a C step's windows 0..4

__device__ __forceinline__ void mf_piece(const unsigned char* sbase, unsigned voff, unsigned lds_dst) {
#ifndef TR_ABLATE_NO_DMA
  asm volatile(
      "s_mov_b32 m0, %[ld]\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o], %[b]"
      :
      : [o] "v"(voff), [b] "s"(sbase), [ld] "s"(lds_dst)
      : "memory", "m0");
#endif
}

template <int IMM>
__device__ __forceinline__ bf16x8 mf_load_x(const uint16_t* sbase, unsigned voff) {
  bf16x8 v;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "i"(IMM) : "memory");
  return v;
}
// refill IN PLACE ("+v": input and output share the register, so the not-taken side of the uniform branch around it needs no copy)
template <int IMM>
__device__ __forceinline__ void mf_reload_x(bf16x8& v, const uint16_t* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(v) : "v"(voff), "s"(sbase), "i"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ f32x4 mf_load_f4(const float* sbase, unsigned voff) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "i"(IMM) : "memory");
  return v;
}

#define MF_TIE_X(x)                                                                                                                   \
  "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), \
      "+v"(x[11])

// Fragment-major packing of one block's Mlp weights.  pk[step s][fragment f][lane][8 bf16]:
//   f <  24: W1 fragment (tile t2 = f / 12, k-step ks = f % 12): lane (r = l & 15, q = l >> 4) holds W1[32 s + 8 (r >> 2) + 4 t2 + (r & 3)][32 ks + 8 q ..]
//   f >= 24: W2 fragment of column group i = f - 24:             lane (r, q) holds W2[16 i + r][32 s + 8 q ..]
// Behind the Hd / 32 steps: fc2's bias as the C wave's accumulator image -- fragment (i, j) [j = 0, 1: the two row groups hold the same values],
// lane (r, q): b2[16 i + 4 q .. + 3] as fp32 -- so that "the accumulators restart at the bias" is the same 48 loads as "the accumulators
// continue another workgroup's" (stream-K hand-over), from another base address.
__global__ __launch_bounds__(256) void mlp_pack_kernel(const uint16_t* __restrict__ W1, const uint16_t* __restrict__ W2, const float* __restrict__ b2,
                                                       u32x4* __restrict__ pk, int D, int Hd) {
  const int ks_n = D / 32, ni = D / 16, fr = 2 * ks_n + ni;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);          // (step, fragment)
  const int lane = threadIdx.x & 63;
  const int s = g / fr, f = g % fr;
  if (s >= Hd / 32) {
    const int e = g - (Hd / 32) * fr;                          // bias image entry (i, j)
    if (e < 2 * ni) pk[(size_t)g * 64 + lane] = *reinterpret_cast<const u32x4*>(b2 + 16 * (e >> 1) + 4 * (lane >> 4));
    return;
  }
  const int r = lane & 15, q = lane >> 4;
  const uint16_t* src;
  if (f < 2 * ks_n) {
#ifdef MF_P32
    // lab (-DMF_P32: fc1 on v_mfma_f32_32x32x16_bf16): fragment f = 16-deep k-step ks; lane (rho = l & 31, kq = l >> 5) holds
    // W1[32 s + h(rho)][16 ks + 8 kq ..], h(rho = 8 i + 4 g + e) = 8 (g + 2 (i >> 1)) + 4 (i & 1) + e: a lane (token, g) of the 32 x 32 result
    // then holds the hidden groups q = g (registers 0..7) and q = g + 2 (8..15), each in fc2's B-fragment order
    const int rho = lane & 31, kq = lane >> 5;
    const int hh = 8 * (((rho >> 2) & 1) + 2 * (rho >> 4)) + 4 * ((rho >> 3) & 1) + (rho & 3);
    src = W1 + (size_t)(32 * s + hh) * D + 16 * f + 8 * kq;
#else
    const int t2 = f / ks_n, ks = f % ks_n;
    src = W1 + (size_t)(32 * s + 8 * (r >> 2) + 4 * t2 + (r & 3)) * D + 32 * ks + 8 * q;
#endif
  } else {
    const int i = f - 2 * ks_n;
    src = W2 + (size_t)(16 * i + r) * Hd + 32 * s + 8 * q;
  }
  pk[(size_t)g * 64 + lane] = *reinterpret_cast<const u32x4*>(src);
}

// The steps of ONE workgroup in execution order.  Whole-block schedule (scratch == nullptr, or at most one block per workgroup): blocks bid,
// bid + G, ..  Stream-K schedule (more blocks than workgroups): the launch's nblk * NS steps are cut into G equal contiguous ranges; a range
// begins inside a block (its TAIL: steps s0..NS-1) and ends inside another (its HEAD: steps 0..s1-1).  A block is a chain -- fc2's K order is
// kept, so the result stays bit-identical -- hence the workgroup that owns the tail CONTINUES the accumulator of the workgroup that owns the
// head: the head runs FIRST (its fp32 accumulator is published: write-through stores + a counter), the tail LAST, and since a range is longer
// than a block the published accumulator has been waiting for at least (range - NS) steps when it is fetched.
struct MfSeq {
  int head_len, head_blk;      // steps 0 .. head_len-1 of block head_blk first (0: none)
  int full0, fstride, nfull;   // then the whole blocks full0 + k * fstride, k < nfull
  int tail_s0, tail_blk;       // then steps tail_s0 .. NS-1 of block tail_blk (tail_s0 == NS: none)
};
struct MfCur {                 // a position in that sequence: segment -1 = head, 0..nfull-1 = whole blocks, nfull = tail
  int seg, s;
};
__device__ __forceinline__ MfCur mf_first(const MfSeq& q, int NS) {
  if (q.head_len > 0) return MfCur{-1, 0};
  if (q.nfull > 0) return MfCur{0, 0};
  return MfCur{q.nfull, q.tail_s0};
}
__device__ __forceinline__ bool mf_last_of_segment(const MfSeq& q, const MfCur& c, int NS) { return c.seg == -1 ? c.s == q.head_len - 1 : c.s == NS - 1; }
__device__ __forceinline__ int mf_block(const MfSeq& q, const MfCur& c) {
  return c.seg == -1 ? q.head_blk : (c.seg < q.nfull ? q.full0 + c.seg * q.fstride : q.tail_blk);
}
__device__ __forceinline__ MfCur mf_next(const MfSeq& q, const MfCur& c, int NS) {        // past the end: keeps returning valid (unused) steps
  if (!mf_last_of_segment(q, c, NS)) return MfCur{c.seg, c.s + 1};
  const int seg = c.seg + 1;
  if (seg < q.nfull) return MfCur{seg, 0};
  return MfCur{q.nfull, q.tail_s0 < NS ? q.tail_s0 : 0};
}

constexpr int MF_SK_SLOT = 4 * MF_NI * 2 * 1024;     // one workgroup's published accumulator: 4 C waves x 48 fragments x 1 KiB = 192 KiB
constexpr int MF_SK_CNT = 128;                       // bytes per counter (its own line)
constexpr int MF_SK_ERR = 128;                       // behind the slots: the error record of a hand-over poll that ran out (magic, workgroup + 1)
constexpr int MF_SK_SETS = 32;                       // counter sets behind that (one per launch of a forward: TR_MAX_DEPTH), each one line per workgroup
constexpr unsigned MF_SK_ERR_MAGIC = 0x4d46534bu;    // "MFSK"

// RL ("residual + LayerNorm" tail of a transformer block, topk.py:95 followed by the next block's :87 norm1): the C wave's accumulator starts at
// the block's rows of the fp32 residual stream `rl.x` (which already holds x + attention branch) instead of at fc2's bias, so that at the
// block's end it IS the new stream row  x + fc2(gelu(fc1(norm2 x))) + b2  -- written back in place -- and, the wave owning whole rows, the
// next block's norm1 of it (two-pass mean / variance over the row's 4 lanes x 96 values, tr_norm.hip's formulas) goes out as bf16 through
// the same staged whole-line stores.  One launch replaces fc1, fc2 AND the residual-add + LayerNorm kernel behind them; the fc2 output is
// never rounded to bf16 on the way into the stream.
struct MfResid {
  float* x;            // [M, 384] fp32 residual stream, updated in place
  const float* b2;     // fc2 bias
  const float* g;      // next norm1: weight, bias, eps
  const float* b;
  float eps;
};

// NP ("norm prologue": topk.py:95 `self.mlp(self.norm2(x))` with the norm2 INSIDE the launch): the P wave builds its 32 x 384 B operand
// from the fp32 residual stream and the pending bf16 attention residual -- LayerNorm(x + d; g, b) -- instead of loading a normalised bf16
// row.  A token row sits in the four lanes (lane & 15, q = 0..3) of a wave, 96 values each: statistics in-lane + two cross-lane steps.
// Bit-identical to tr_layernorm2_bf16 followed by the plain launch: the sums are formed in layernorm_half_kernel's order (its lane
// `sub` = 8 (ks & 3) + 2 q + h holds the chunks ks, ks + 4, ks + 8 of half h; its butterfly xor 16, 8 is in-lane here, xor 4, 2 are this
// wave's lane ^ 32, ^ 16, xor 1 is the in-lane half h), with its contractions spelled out.  Removes the lazy-norm2 launch in front of the
// Mlp (8 B per element through HBM) and the bf16 round trip of the normalised rows.
struct MfNorm {
  const float* x;        // [M, 384] fp32 residual stream (not written)
  const uint16_t* d;     // [M, 384] bf16 pending residual (the attention branch)
  const float* g;        // norm2 weight, bias, eps
  const float* b;
  float eps;
};

template <bool RL, bool NP>
__global__ __launch_bounds__(512, 2) void mlp_fused_kernel(const uint16_t* __restrict__ xn, const unsigned char* __restrict__ pk,
                                                           const float* __restrict__ b1,
                                                           uint16_t* __restrict__ outp, unsigned char* __restrict__ scratch, int M, int NS,
                                                           unsigned out_bytes, int poll_max, int cset, const MfResid rl, const MfNorm nm) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[MF_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pr = wave & 3;                                   // wave pair = 32-row slice of the block
#ifdef MF_XCD_RANGES
  // lab: consecutive stream-K ranges on ONE XCD (workgroup w runs on XCD w % 8): range index = (w % 8) * (G / 8) + w / 8, so that a hand-over
  // stays behind one L2 except at seven XCD boundaries.  Measured (tools/lab/mlp_xcd_ab.sh): no gain -- 156.5-159.5 vs 154.2-158.2 us at 50,432
  // rows, 181 vs 177-178.5 at 70,001: the write-through stores and sc1 loads of the hand-over go to memory either way
  const int G = gridDim.x;
  const int bid = (G % 8 == 0 && G > 8) ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
#else
  const int G = gridDim.x, bid = blockIdx.x;
#endif
  const int nblk = (M + MF_ROWS - 1) / MF_ROWS;
  if (bid >= nblk) return;
  MfSeq q;
  int T;                              // P: MFMAs of step number t at time t = 0..T-1, its GELU at time t+1; C consumes it at time t+2
  const bool streamk = scratch != nullptr && nblk > G;
  if (!streamk) {
    q = MfSeq{0, 0, bid, G, (nblk - bid + G - 1) / G, NS, 0};
    T = q.nfull * NS;
  } else {
    const long long U = (long long)nblk * NS;
    const int u0 = (int)(U * bid / G), u1 = (int)(U * (bid + 1) / G);
    const int b_first = u0 / NS, s0 = u0 - b_first * NS;          // s0 > 0: this range starts inside block b_first
    const int b_end = u1 / NS, s1 = u1 - b_end * NS;              // s1 > 0: it ends inside block b_end
    const int f0 = s0 > 0 ? b_first + 1 : b_first;
    q = MfSeq{s1, b_end, f0, 1, b_end - f0, s0 > 0 ? s0 : NS, b_first};
    T = u1 - u0;
  }
  q.head_len = __builtin_amdgcn_readfirstlane(q.head_len); q.head_blk = __builtin_amdgcn_readfirstlane(q.head_blk);
  q.full0 = __builtin_amdgcn_readfirstlane(q.full0); q.fstride = __builtin_amdgcn_readfirstlane(q.fstride);
  q.nfull = __builtin_amdgcn_readfirstlane(q.nfull); q.tail_s0 = __builtin_amdgcn_readfirstlane(q.tail_s0);
  q.tail_blk = __builtin_amdgcn_readfirstlane(q.tail_blk);
  T = __builtin_amdgcn_readfirstlane(T);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int frow = lane & 15, fq = lane >> 4;
  unsigned char* const hb = smem + MF_NSLOT * MF_ENTRY + pr * (2 * MF_HBUF);     // this pair's two hidden buffers

  // ---- the weight ring.  Entry e = [W1 of step e | W2 of step e-2]: what time step e consumes.  During time step t the waves issue the
  // pieces of entry t+2 into the slot entry t-1 has left; the barrier at the end of step t publishes entry t+1 (issued a step earlier: every
  // wave waits for its own pieces with a counted vmcnt, this step's stay in flight).
  // The barrier sits BEFORE a step's last window of MFMAs: its fragments are in registers by then, so the matrix pipe has work the moment the
  // barrier opens, under which the first fragments of the next step arrive from LDS.
  MfCur ld_cur = mf_first(q, NS);
  int ld_e = 0, ld_w1 = ld_cur.s, ld_w2 = ld_cur.s, ld_w2n = ld_cur.s, ld_slot = 0;      // W1 step of entry e, W2 step of entries e and e+1 (= W1 steps of e-2, e-1)
  // the 48 pieces of an entry: a P wave issues MF_PQ of them (pieces pr + 4 q), a C wave the other 12 - MF_PQ (pieces 4 MF_PQ + pr + 4 q)
  auto issue_piece_q = [&](int q) __attribute__((always_inline)) {
    const int f = (wave < 4 ? 0 : 4 * MF_PQ) + pr + 4 * q;
    const int step = (f < MF_W1FR) ? ld_w1 : ld_w2;
    mf_piece(pk + (size_t)step * MF_ENTRY + f * 1024, lane16, lds0 + ld_slot * MF_ENTRY + f * 1024);
  };
  auto issue_all = [&]() __attribute__((always_inline)) {
    if (wave < 4) {
#pragma unroll
      for (int q = 0; q < MF_PQ; ++q) issue_piece_q(q);
    } else {
#pragma unroll
      for (int q = 0; q < 12 - MF_PQ; ++q) issue_piece_q(q);
    }
  };
  auto advance_entry = [&]() __attribute__((always_inline)) {
    ++ld_e;
    ld_w2 = ld_w2n;
    ld_w2n = ld_w1;
    ld_cur = mf_next(q, ld_cur, NS);
    ld_w1 = ld_cur.s;
    ld_slot = (ld_slot + 1 == MF_NSLOT) ? 0 : ld_slot + 1;
  };

#ifdef MF_P32
  // =================================================================== lab: P on v_mfma_f32_32x32x16_bf16 (profiles/r06_mlp_lab.md)
  // The fc1 waves only: h^T (32 hidden x 32 tokens) = W1_step . x^T as 24 MFMAs of 32 x 32 x 16 on ONE 16-register accumulator instead of 48
  // of 16 x 16 x 32 on four -- half the MFMA issue slots on the wave whose instruction stream bounds the step.  The fc2 waves, the ring, the
  // hand-over and the epilogue are untouched: the hidden fragments cross LDS anyway, so the P wave writes them where the C wave's
  // 16 x 16 x 32 B-fragment reads expect them.  Not bit-identical to the GEMM pair (another K order inside the instruction).
  if (wave < 4) {
    typedef __attribute__((ext_vector_type(16))) float f32x16;
    const int tc = lane & 31, hg = lane >> 5;     // token of the slice, hidden-group half
    bf16x8 xk[2 * MF_KS];                          // B operand: token tc, columns 16 ks + 8 hg .. (24 k-steps)
    auto x_offset = [&](int blk) __attribute__((always_inline)) {
      const int m = blk * MF_ROWS + pr * 32 + tc;
      return ((unsigned)min(m, M - 1) * MF_D + 8u * hg) * 2u;
    };
    {
      const unsigned xo = x_offset(mf_block(q, mf_first(q, NS)));
#define MF_LX(ks) xk[ks] = mf_load_x<(ks) * 32>(xn, xo)
      MF_LX(0); MF_LX(1); MF_LX(2); MF_LX(3); MF_LX(4); MF_LX(5); MF_LX(6); MF_LX(7); MF_LX(8); MF_LX(9); MF_LX(10); MF_LX(11);
      MF_LX(12); MF_LX(13); MF_LX(14); MF_LX(15); MF_LX(16); MF_LX(17); MF_LX(18); MF_LX(19); MF_LX(20); MF_LX(21); MF_LX(22); MF_LX(23);
#undef MF_LX
    }
    // bias of the lane's 16 hidden units of a step: b1[32 s + 8 hg + 0..7] (registers 0..7), b1[32 s + 8 (hg + 2) + 0..7] (8..15)
    const unsigned boff = (unsigned)hg * 32u;
    f32x4 bn0, bn1, bn2, bn3;
    bn0 = mf_load_f4<0>(b1 + 32 * mf_first(q, NS).s, boff);
    bn1 = mf_load_f4<16>(b1 + 32 * mf_first(q, NS).s, boff);
    bn2 = mf_load_f4<64>(b1 + 32 * mf_first(q, NS).s, boff);
    bn3 = mf_load_f4<80>(b1 + 32 * mf_first(q, NS).s, boff);
    issue_all();
    advance_entry();
    issue_all();
    advance_entry();
#define MF_TIE_XK(lo) "+v"(xk[lo]), "+v"(xk[lo + 1]), "+v"(xk[lo + 2]), "+v"(xk[lo + 3]), "+v"(xk[lo + 4]), "+v"(xk[lo + 5]), "+v"(xk[lo + 6]), "+v"(xk[lo + 7]), "+v"(xk[lo + 8]), "+v"(xk[lo + 9]), "+v"(xk[lo + 10]), "+v"(xk[lo + 11])
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bn0), "+v"(bn1), "+v"(bn2), "+v"(bn3), MF_TIE_XK(0), MF_TIE_XK(12)::"memory");
    __builtin_amdgcn_s_barrier();               // entries 0 and 1 have landed
    asm volatile("" ::: "memory");

    int t = 0, cslot = 0;
    f32x4 p0, p1, p2, p3;                       // the previous step's accumulator (registers 0..3, 4..7, 8..11, 12..15), waiting for its GELU
    p0 = p1 = p2 = p3 = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    bf16x8 wA[4], wB[4];
#define MF_READW(buf, base, k)                                                     \
  buf[0] = *reinterpret_cast<const bf16x8*>((base) + (4 * (k)) * 1024);            \
  buf[1] = *reinterpret_cast<const bf16x8*>((base) + (4 * (k) + 1) * 1024);        \
  buf[2] = *reinterpret_cast<const bf16x8*>((base) + (4 * (k) + 2) * 1024);        \
  buf[3] = *reinterpret_cast<const bf16x8*>((base) + (4 * (k) + 3) * 1024)
    MF_READW(wA, smem + lane16, 0);
    MF_CLOCK_BEGIN;
    // where this lane's two 8-wide groups go in the pair's hidden buffer: fc2's B fragment of token group j = tc >> 4 is read at
    // j * 1024 + (r + 16 q) * 16 by lane (r = token & 15, q = hidden group)
    const unsigned hdst = (unsigned)((tc >> 4) * 1024 + ((tc & 15) + 16 * hg) * 16);
    auto p_step = [&](int sn, const bool last, int next_blk) __attribute__((always_inline)) {
      const unsigned char* slot = smem + cslot * MF_ENTRY + lane16;
      const int nslot = (cslot + 1 == MF_NSLOT) ? 0 : cslot + 1;
      const unsigned char* slot_next = smem + nslot * MF_ENTRY + lane16;
      MF_STAMP_DECL;
      MF_STAMP(0);
      f32x16 a;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = bn0[e]; a[4 + e] = bn1[e]; a[8 + e] = bn2[e]; a[12 + e] = bn3[e]; }
      u32x4 h0 = {0u, 0u, 0u, 0u}, h1 = {0u, 0u, 0u, 0u};
      unsigned xo = 0;
      const bool reload = last && next_blk >= 0;
      if (reload) xo = x_offset(next_blk);
#define MF_GELU_SLICE(k)                                                                                  \
  do {                                                                                                    \
    if ((k) == 0) { const f32x2 g0 = mf_gelu2(f32x2{p0[0], p0[1]}), g1 = mf_gelu2(f32x2{p0[2], p0[3]}); h0[0] = pack_bf16x2(g0[0], g0[1]); h0[1] = pack_bf16x2(g1[0], g1[1]); asm volatile("" : "+v"(h0[0]), "+v"(h0[1])); } \
    if ((k) == 1) { const f32x2 g0 = mf_gelu2(f32x2{p1[0], p1[1]}), g1 = mf_gelu2(f32x2{p1[2], p1[3]}); h0[2] = pack_bf16x2(g0[0], g0[1]); h0[3] = pack_bf16x2(g1[0], g1[1]); asm volatile("" : "+v"(h0[2]), "+v"(h0[3])); } \
    if ((k) == 2) { const f32x2 g0 = mf_gelu2(f32x2{p2[0], p2[1]}), g1 = mf_gelu2(f32x2{p2[2], p2[3]}); h1[0] = pack_bf16x2(g0[0], g0[1]); h1[1] = pack_bf16x2(g1[0], g1[1]); asm volatile("" : "+v"(h1[0]), "+v"(h1[1])); } \
    if ((k) == 3) { const f32x2 g0 = mf_gelu2(f32x2{p3[0], p3[1]}), g1 = mf_gelu2(f32x2{p3[2], p3[3]}); h1[2] = pack_bf16x2(g0[0], g0[1]); h1[3] = pack_bf16x2(g1[0], g1[1]); asm volatile("" : "+v"(h1[2]), "+v"(h1[3])); } \
    if ((k) == 4) {                                                                                       \
      unsigned char* dst = hb + ((t - 1) & 1) * MF_HBUF + hdst;                                           \
      *reinterpret_cast<u32x4*>(dst) = h0;                                                                \
      *reinterpret_cast<u32x4*>(dst + 512) = h1;                                                          \
    }                                                                                                     \
  } while (0)
#define MF_PMFMA(cur, k)                                                                                           \
  do {                                                                                                             \
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[0], xk[4 * (k)], a, 0, 0, 0);                                  \
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[1], xk[4 * (k) + 1], a, 0, 0, 0);                              \
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[2], xk[4 * (k) + 2], a, 0, 0, 0);                              \
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[3], xk[4 * (k) + 3], a, 0, 0, 0);                              \
  } while (0)
#define MF_RX(ks) mf_reload_x<(ks) * 32>(xk[ks], xn, xo)
#define MF_PWIN(cur, nxt, k)                                                                    \
  do {                                                                                          \
    MF_READW(nxt, slot, (k) + 1);                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    MF_PMFMA(cur, k);                                                                           \
    MF_GELU_SLICE(k);                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if ((k) == 0) {                                                                             \
      bn0 = mf_load_f4<0>(b1 + 32 * sn, boff);                                                  \
      bn1 = mf_load_f4<16>(b1 + 32 * sn, boff);                                                 \
      bn2 = mf_load_f4<64>(b1 + 32 * sn, boff);                                                 \
      bn3 = mf_load_f4<80>(b1 + 32 * sn, boff);                                                 \
    }                                                                                           \
    if (reload) { MF_RX(4 * (k)); MF_RX(4 * (k) + 1); MF_RX(4 * (k) + 2); MF_RX(4 * (k) + 3); } \
  } while (0)
      MF_PWIN(wA, wB, 0);
      MF_PWIN(wB, wA, 1);
      MF_PWIN(wA, wB, 2);
      MF_PWIN(wB, wA, 3);
      MF_PWIN(wA, wB, 4);       // wB: window 5
      if (ld_e <= T + 1) advance_entry();
      MF_STAMP(1);
      if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(bn0), "+v"(bn1), "+v"(bn2), "+v"(bn3), MF_TIE_XK(0), MF_TIE_XK(12)::"memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MF_STAMP(2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      MF_STAMP(3);
      MF_READW(wA, slot_next, 0);
      __builtin_amdgcn_sched_barrier(0);
      MF_PMFMA(wB, 5);
      __builtin_amdgcn_sched_barrier(0);
      if (reload) { MF_RX(20); MF_RX(21); MF_RX(22); MF_RX(23); }
#undef MF_PWIN
#undef MF_GELU_SLICE
#pragma unroll
      for (int e = 0; e < 4; ++e) { p0[e] = a[e]; p1[e] = a[4 + e]; p2[e] = a[8 + e]; p3[e] = a[12 + e]; }
      MF_STAMP_DUMP(0, t);
      ++t;
      cslot = nslot;
    };
    {
      MfCur cur = mf_first(q, NS);
      for (int tt = 0; tt < T; ++tt) {
        const MfCur nxt = mf_next(q, cur, NS);
        int last = __builtin_amdgcn_readfirstlane(mf_last_of_segment(q, cur, NS) ? 1 : 0);
        asm volatile("" : "+s"(last));
        p_step(nxt.s, last != 0, (last && tt + 1 < T) ? mf_block(q, nxt) : -1);
        cur = nxt;
      }
    }
    {
      // time T: only the GELU of the last step is left
      const f32x2 g0 = mf_gelu2(f32x2{p0[0], p0[1]}), g1 = mf_gelu2(f32x2{p0[2], p0[3]});
      const f32x2 g2 = mf_gelu2(f32x2{p1[0], p1[1]}), g3 = mf_gelu2(f32x2{p1[2], p1[3]});
      const u32x4 h0 = {pack_bf16x2(g0[0], g0[1]), pack_bf16x2(g1[0], g1[1]), pack_bf16x2(g2[0], g2[1]), pack_bf16x2(g3[0], g3[1])};
      const f32x2 g4 = mf_gelu2(f32x2{p2[0], p2[1]}), g5 = mf_gelu2(f32x2{p2[2], p2[3]});
      const f32x2 g6 = mf_gelu2(f32x2{p3[0], p3[1]}), g7 = mf_gelu2(f32x2{p3[2], p3[3]});
      const u32x4 h1 = {pack_bf16x2(g4[0], g4[1]), pack_bf16x2(g5[0], g5[1]), pack_bf16x2(g6[0], g6[1]), pack_bf16x2(g7[0], g7[1])};
      unsigned char* dst = hb + ((t - 1) & 1) * MF_HBUF + hdst;
      *reinterpret_cast<u32x4*>(dst) = h0;
      *reinterpret_cast<u32x4*>(dst + 512) = h1;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#undef MF_PMFMA
#undef MF_READW
#undef MF_RX
#undef MF_TIE_XK
    if constexpr (RL) __builtin_amdgcn_s_barrier();       // the C waves' barrier in front of the range's last epilogue (see there)
    MF_CLOCK_END;
    return;
  }
#else
  if (wave < 4) {
    // =============================================================== P: fc1 + GELU
#ifdef MF_PRIO_P
    __builtin_amdgcn_s_setprio(MF_PRIO_P);
#endif
    bf16x8 x0[MF_KS], x1[MF_KS];                 // B-operand fragments of this wave's two 16-row groups, all 12 k-steps
    f32x4 bn0, bn1;                              // bias fragments of the NEXT step's two row tiles
    const unsigned boff = (unsigned)fq * 32u;    // lane (.., q) starts at hidden 8 q of the step: b1[32 s + 8 q + 4 t2 + e]
    auto x_offsets = [&](int blk, unsigned& o0, unsigned& o1) __attribute__((always_inline)) {
      const int m = blk * MF_ROWS + pr * 32 + frow;
      o0 = ((unsigned)min(m, M - 1) * MF_D + 8u * fq) * 2u;
      o1 = ((unsigned)min(m + 16, M - 1) * MF_D + 8u * fq) * 2u;
    };
#define MF_LOAD_X(ks)                          \
  x0[ks] = mf_load_x<(ks) * 64>(xn, xo0);      \
  x1[ks] = mf_load_x<(ks) * 64>(xn, xo1)
#define MF_RELOAD_X(ks)                        \
  mf_reload_x<(ks) * 64>(x0[ks], xn, xo0);     \
  mf_reload_x<(ks) * 64>(x1[ks], xn, xo1)
    // NP: x0 / x1 <- LayerNorm(x + d) of block blk's rows, one 16-row group at a time (96 fp32 values per lane live).  The sched_barriers
    // pin the phases: left alone, hipcc issues both groups' loads and all 48 parameter loads up front (121 spilled registers).
    auto ln_fill = [&](int blk) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = min(blk * MF_ROWS + pr * 32 + frow + 16 * j, M - 1);
        const float* xr = nm.x + (size_t)m * MF_D + 8 * fq;
        const uint16_t* dr = nm.d + (size_t)m * MF_D + 8 * fq;
        f32x4 v[MF_KS][2];
        u32x4 dq[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dq[0][i] = *reinterpret_cast<const u32x4*>(dr + 32 * i);
#pragma unroll
        for (int ks = 0; ks < MF_KS; ++ks) {
          v[ks][0] = *reinterpret_cast<const f32x4*>(xr + 32 * ks);
          v[ks][1] = *reinterpret_cast<const f32x4*>(xr + 32 * ks + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (c + 1 < 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dq[(c + 1) & 1][i] = *reinterpret_cast<const u32x4*>(dr + 32 * (4 * (c + 1) + i));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const u32x4 dd = dq[c & 1][i];
            const int ks = 4 * c + i;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              v[ks][h][0] += __uint_as_float(dd[2 * h] << 16); v[ks][h][1] += __uint_as_float(dd[2 * h] & 0xffff0000u);
              v[ks][h][2] += __uint_as_float(dd[2 * h + 1] << 16); v[ks][h][3] += __uint_as_float(dd[2 * h + 1] & 0xffff0000u);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // the row sum: per (ks & 3, h) the three chunks in order, then the tree (see the header of MfNorm)
        float sp[4][2];
#pragma unroll
        for (int ks = 0; ks < MF_KS; ++ks)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float t = (v[ks][h][0] + v[ks][h][1]) + (v[ks][h][2] + v[ks][h][3]);
            sp[ks & 3][h] = (ks < 4 ? 0.f : sp[ks & 3][h]) + t;
          }
        float c0 = (sp[0][0] + sp[2][0]) + (sp[1][0] + sp[3][0]), c1 = (sp[0][1] + sp[2][1]) + (sp[1][1] + sp[3][1]);
        c0 += __shfl_xor(c0, 32); c1 += __shfl_xor(c1, 32);
        c0 += __shfl_xor(c0, 16); c1 += __shfl_xor(c1, 16);
        const float mean = (c0 + c1) / (float)MF_D;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < MF_KS; ++ks)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x4 t = v[ks][h];
            t[0] -= mean; t[1] -= mean; t[2] -= mean; t[3] -= mean;
            v[ks][h] = t;
            const float u = __builtin_fmaf(t[0], t[0], t[1] * t[1]) + __builtin_fmaf(t[2], t[2], t[3] * t[3]);
            sp[ks & 3][h] = (ks < 4 ? 0.f : sp[ks & 3][h]) + u;
          }
        c0 = (sp[0][0] + sp[2][0]) + (sp[1][0] + sp[3][0]); c1 = (sp[0][1] + sp[2][1]) + (sp[1][1] + sp[3][1]);
        c0 += __shfl_xor(c0, 32); c1 += __shfl_xor(c1, 32);
        c0 += __shfl_xor(c0, 16); c1 += __shfl_xor(c1, 16);
        const float rstd = rsqrtf((c0 + c1) / (float)MF_D + nm.eps);
        const float* gr = nm.g + 8 * fq;
        const float* br = nm.b + 8 * fq;
        // normalise, scale, round: the parameters of chunk ks + 1 are requested before chunk ks is computed (two sets of 16 registers)
        f32x4 gb[2][4];
        gb[0][0] = *reinterpret_cast<const f32x4*>(gr); gb[0][1] = *reinterpret_cast<const f32x4*>(gr + 4);
        gb[0][2] = *reinterpret_cast<const f32x4*>(br); gb[0][3] = *reinterpret_cast<const f32x4*>(br + 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < MF_KS; ++ks) {
          if (ks + 1 < MF_KS) {
            gb[(ks + 1) & 1][0] = *reinterpret_cast<const f32x4*>(gr + 32 * (ks + 1)); gb[(ks + 1) & 1][1] = *reinterpret_cast<const f32x4*>(gr + 32 * (ks + 1) + 4);
            gb[(ks + 1) & 1][2] = *reinterpret_cast<const f32x4*>(br + 32 * (ks + 1)); gb[(ks + 1) & 1][3] = *reinterpret_cast<const f32x4*>(br + 32 * (ks + 1) + 4);
          }
          __builtin_amdgcn_sched_barrier(0);
          u32x4 o;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 gg = gb[ks & 1][h], bb = gb[ks & 1][2 + h];
            const f32x4 t = v[ks][h];
            o[2 * h] = pack_bf16x2(__builtin_fmaf(rstd * t[0], gg[0], bb[0]), __builtin_fmaf(rstd * t[1], gg[1], bb[1]));
            o[2 * h + 1] = pack_bf16x2(__builtin_fmaf(rstd * t[2], gg[2], bb[2]), __builtin_fmaf(rstd * t[3], gg[3], bb[3]));
          }
          if (j == 0) x0[ks] = __builtin_bit_cast(bf16x8, o);
          else x1[ks] = __builtin_bit_cast(bf16x8, o);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if constexpr (NP) {
      ln_fill(mf_block(q, mf_first(q, NS)));
    } else {
      unsigned xo0, xo1;
      x_offsets(mf_block(q, mf_first(q, NS)), xo0, xo1);
      MF_LOAD_X(0); MF_LOAD_X(1); MF_LOAD_X(2); MF_LOAD_X(3); MF_LOAD_X(4); MF_LOAD_X(5);
      MF_LOAD_X(6); MF_LOAD_X(7); MF_LOAD_X(8); MF_LOAD_X(9); MF_LOAD_X(10); MF_LOAD_X(11);
    }
    bn0 = mf_load_f4<0>(b1 + 32 * mf_first(q, NS).s, boff);
    bn1 = mf_load_f4<16>(b1 + 32 * mf_first(q, NS).s, boff);
    issue_all();
    advance_entry();
    issue_all();
    advance_entry();
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bn0), "+v"(bn1), MF_TIE_X(x0), MF_TIE_X(x1)::"memory");
    __builtin_amdgcn_s_barrier();               // entries 0 and 1 have landed
    asm volatile("" ::: "memory");

    int t = 0, cslot = 0;
    f32x4 p00, p01, p10, p11;                   // the previous step's accumulators, waiting for their GELU
    p00 = p01 = p10 = p11 = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(p00), "+v"(p01), "+v"(p10), "+v"(p11));      // opaque: no first iteration peeled off to fold GELU(0) (a second copy of the step body)
    bf16x8 wA[4], wB[4];
#define MF_READW(buf, base, k)                                                                  \
  buf[0] = *reinterpret_cast<const bf16x8*>((base) + (2 * (k)) * 1024);                        \
  buf[1] = *reinterpret_cast<const bf16x8*>((base) + (MF_KS + 2 * (k)) * 1024);                \
  buf[2] = *reinterpret_cast<const bf16x8*>((base) + (2 * (k) + 1) * 1024);                    \
  buf[3] = *reinterpret_cast<const bf16x8*>((base) + (MF_KS + 2 * (k) + 1) * 1024)
    MF_READW(wA, smem + lane16, 0);
    MF_CLOCK_BEGIN;
    // One time step t: windows 0..4 of its MFMAs with the GELU of step t-1 under them (at t = 0: of zeros, into a buffer nobody reads) -- the
    // hidden fragments go to the partner through LDS --, the barrier, then window 5 and the first fragment reads of step t+1.
    // last: the block's last step -- the x registers are dead after their last MFMA and are refilled IN PLACE with the next block's rows.
    auto p_step = [&](int sn, const bool last, int next_blk) __attribute__((always_inline)) {
      const unsigned char* slot = smem + cslot * MF_ENTRY + lane16;
      const int nslot = (cslot + 1 == MF_NSLOT) ? 0 : cslot + 1;
      const unsigned char* slot_next = smem + nslot * MF_ENTRY + lane16;
      MF_STAMP_DECL;
      MF_STAMP(0);
      f32x4 a00 = bn0, a01 = bn0, a10 = bn1, a11 = bn1;
      u32x4 h0 = {0u, 0u, 0u, 0u}, h1 = {0u, 0u, 0u, 0u};
      unsigned xo0 = 0, xo1 = 0;
      const bool reload = last && next_blk >= 0;
      if (reload) x_offsets(next_blk, xo0, xo1);
      const bool dma = ld_e <= T + 1;
      // GELU of the previous step in four slices (one per window 0..3; the empty asm pins a slice to its window -- hipcc sinks the whole GELU to
      // the store otherwise, behind 32 MFMAs), a lane's eight values = the partner's B fragment: tile 0 gives k = 8 q + 0..3, tile 1 k = 8 q + 4..7
#define MF_GELU_SLICE(k)                                                                                  \
  do {                                                                                                    \
    if ((k) == 0) { const f32x2 g0 = mf_gelu2(f32x2{p00[0], p00[1]}), g1 = mf_gelu2(f32x2{p00[2], p00[3]}); h0[0] = pack_bf16x2(g0[0], g0[1]); h0[1] = pack_bf16x2(g1[0], g1[1]); asm volatile("" : "+v"(h0[0]), "+v"(h0[1])); } \
    if ((k) == 1) { const f32x2 g0 = mf_gelu2(f32x2{p10[0], p10[1]}), g1 = mf_gelu2(f32x2{p10[2], p10[3]}); h0[2] = pack_bf16x2(g0[0], g0[1]); h0[3] = pack_bf16x2(g1[0], g1[1]); asm volatile("" : "+v"(h0[2]), "+v"(h0[3])); } \
    if ((k) == 2) { const f32x2 g0 = mf_gelu2(f32x2{p01[0], p01[1]}), g1 = mf_gelu2(f32x2{p01[2], p01[3]}); h1[0] = pack_bf16x2(g0[0], g0[1]); h1[1] = pack_bf16x2(g1[0], g1[1]); asm volatile("" : "+v"(h1[0]), "+v"(h1[1])); } \
    if ((k) == 3) { const f32x2 g0 = mf_gelu2(f32x2{p11[0], p11[1]}), g1 = mf_gelu2(f32x2{p11[2], p11[3]}); h1[2] = pack_bf16x2(g0[0], g0[1]); h1[3] = pack_bf16x2(g1[0], g1[1]); asm volatile("" : "+v"(h1[2]), "+v"(h1[3])); } \
    if ((k) == 4) {                                                                                       \
      unsigned char* dst = hb + ((t - 1) & 1) * MF_HBUF + lane16;                                         \
      *reinterpret_cast<u32x4*>(dst) = h0;                                                                \
      *reinterpret_cast<u32x4*>(dst + 1024) = h1;                                                         \
    }                                                                                                     \
  } while (0)
#define MF_PMFMA(cur, k)                                                                        \
  do {                                                                                          \
    a00 = MF_MFMA(cur[0], x0[2 * (k)], a00);                                                    \
    a01 = MF_MFMA(cur[0], x1[2 * (k)], a01);                                                    \
    a10 = MF_MFMA(cur[1], x0[2 * (k)], a10);                                                    \
    a11 = MF_MFMA(cur[1], x1[2 * (k)], a11);                                                    \
    a00 = MF_MFMA(cur[2], x0[2 * (k) + 1], a00);                                                \
    a01 = MF_MFMA(cur[2], x1[2 * (k) + 1], a01);                                                \
    a10 = MF_MFMA(cur[3], x0[2 * (k) + 1], a10);                                                \
    a11 = MF_MFMA(cur[3], x1[2 * (k) + 1], a11);                                                \
  } while (0)
      // windows 0..4: the next window's fragments are requested first, the DMA pieces and the bias loads go out BEHIND the window's MFMAs
#define MF_PWIN(cur, nxt, k)                                                                    \
  do {                                                                                          \
    MF_READW(nxt, slot, (k) + 1);                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    MF_PMFMA(cur, k);                                                                           \
    MF_GELU_SLICE(k);                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if ((k) == 0) {                                                                             \
      /* the next step's bias: older than this step's DMA pieces, so the counted wait at the end of the step retires it */ \
      bn0 = mf_load_f4<0>(b1 + 32 * sn, boff);                                                  \
      bn1 = mf_load_f4<16>(b1 + 32 * sn, boff);                                                 \
    }                                                                                           \
    if ((k) >= 1 && dma) {                                                                      \
      _Pragma("unroll") for (int q_ = MF_PPW * ((k) - 1); q_ < MF_PPW * (k); ++q_)              \
        if (q_ < MF_PQ) issue_piece_q(q_);                                                      \
    }                                                                                           \
    if (!NP && reload) { MF_RELOAD_X(2 * (k)); MF_RELOAD_X(2 * (k) + 1); }                      \
  } while (0)
      MF_PWIN(wA, wB, 0);
      MF_PWIN(wB, wA, 1);
      MF_PWIN(wA, wB, 2);
      MF_PWIN(wB, wA, 3);
      MF_PWIN(wA, wB, 4);       // wB: window 5
      if (dma) advance_entry();
      MF_STAMP(1);
      // end of the step's LDS and DMA business: entry t+1 has landed once only this step's MF_PQ pieces remain (last: the reload is
      // interleaved with them -> drain); the hidden fragments are written, window 5's fragments are in registers.  ONE tied statement on
      // every path (two tied statements on an if/else made hipcc copy the 96 x registers between its two allocations).
      if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt vmcnt(%26)" : "+v"(bn0), "+v"(bn1), MF_TIE_X(x0), MF_TIE_X(x1) : "n"(MF_PQ) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MF_STAMP(2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      MF_STAMP(3);
      // behind the barrier: the step's last window runs on registers while the first fragments of step t+1 (entry t+1: just published) arrive
      MF_READW(wA, slot_next, 0);
      __builtin_amdgcn_sched_barrier(0);
      MF_PMFMA(wB, 5);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NP) {
        // the block's last MFMA has issued: the next block's rows, normalised, take the x registers (the partner finishes its step and
        // waits at the next barrier meanwhile)
        if (reload) {
          ln_fill(next_blk);
          MF_READW(wA, slot_next, 0);      // (again: the copy requested in front of window 5 is not kept through the norm)
        }
      } else {
        if (reload) { MF_RELOAD_X(10); MF_RELOAD_X(11); }
      }
#undef MF_PWIN
#undef MF_GELU_SLICE
      p00 = a00; p01 = a01; p10 = a10; p11 = a11;
      MF_STAMP_DUMP(0, t);
      ++t;
      cslot = nslot;
    };
    // ONE flat loop over the time steps with the position in the sequence carried along -- and the segment's-last-step test opaque: with a
    // nested `for (s = 0; s < NS; ++s)` hipcc peels the last iteration off (the bias pointer wraps there), i.e. compiles the step body twice
    // with two allocations of the 96 x registers and spills between them.
    {
      MfCur cur = mf_first(q, NS);
      for (int tt = 0; tt < T; ++tt) {
        const MfCur nxt = mf_next(q, cur, NS);
        int last = __builtin_amdgcn_readfirstlane(mf_last_of_segment(q, cur, NS) ? 1 : 0);
        asm volatile("" : "+s"(last));
        p_step(nxt.s, last != 0, (last && tt + 1 < T) ? mf_block(q, nxt) : -1);        // (the next step's index: its bias is fetched a step ahead)
        cur = nxt;
      }
    }
    {
      // time T: only the GELU of the last step is left
      const f32x2 g0 = mf_gelu2(f32x2{p00[0], p00[1]}), g1 = mf_gelu2(f32x2{p00[2], p00[3]});
      const f32x2 g2 = mf_gelu2(f32x2{p10[0], p10[1]}), g3 = mf_gelu2(f32x2{p10[2], p10[3]});
      const u32x4 h0 = {pack_bf16x2(g0[0], g0[1]), pack_bf16x2(g1[0], g1[1]), pack_bf16x2(g2[0], g2[1]), pack_bf16x2(g3[0], g3[1])};
      const f32x2 g4 = mf_gelu2(f32x2{p01[0], p01[1]}), g5 = mf_gelu2(f32x2{p01[2], p01[3]});
      const f32x2 g6 = mf_gelu2(f32x2{p11[0], p11[1]}), g7 = mf_gelu2(f32x2{p11[2], p11[3]});
      const u32x4 h1 = {pack_bf16x2(g4[0], g4[1]), pack_bf16x2(g5[0], g5[1]), pack_bf16x2(g6[0], g6[1]), pack_bf16x2(g7[0], g7[1])};
      unsigned char* dst = hb + ((t - 1) & 1) * MF_HBUF + lane16;
      *reinterpret_cast<u32x4*>(dst) = h0;
      *reinterpret_cast<u32x4*>(dst + 1024) = h1;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#undef MF_PMFMA
#undef MF_READW
#undef MF_LOAD_X
#undef MF_RELOAD_X
    if constexpr (RL) __builtin_amdgcn_s_barrier();       // the C waves' barrier in front of the range's last epilogue (see there)
    MF_CLOCK_END;
    return;
  }

#endif      // MF_P32
  // ================================================================= C: fc2, the 32 x 384 accumulator in registers
#ifdef MF_PRIO_C
  __builtin_amdgcn_s_setprio(MF_PRIO_C);
#endif
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)out_bytes, 0x00020000);
  // The accumulators start -- and restart after every block -- at fc2's bias: 48 fragment loads from the bias image behind the packed weights
  // (tr_mlp_pack_bf16).  The tail segment of a stream-K range loads the previous workgroup's accumulator instead: the same loads, another base.
  const unsigned char* const bias_img = pk + (size_t)NS * MF_ENTRY;
  f32x4 acc[MF_NI][2];
  // RL: rows beyond M read as zeros and their stores are dropped (the descriptor's bounds check: num_records = M rows)
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(RL ? reinterpret_cast<unsigned char*>(rl.x) : const_cast<unsigned char*>(bias_img), 0,
                                                                         RL ? (int)(2u * out_bytes) : MF_NI * 2 * 1024, 0x00020000);
  // a lane's first element of block blk in the stream: row blk * 128 + pr * 32 + (lane & 15) [+ 16 j], column 4 (lane >> 4) [+ 16 i]
  auto x_row_off = [&](int blk) __attribute__((always_inline)) { return ((unsigned)(blk * MF_ROWS + pr * 32 + frow) * MF_D + 4u * fq) * 4u; };
  if constexpr (RL) {
    const unsigned vo = x_row_off(mf_block(q, mf_first(q, NS)));
#pragma unroll
    for (int i = 0; i < MF_NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, vo + (unsigned)(j * 16 * MF_D * 4), i * 64, 0));
  } else {
#pragma unroll
    for (int i = 0; i < MF_NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, lane16 + (unsigned)((i * 2 + j) * 1024), 0, 0));
  }
  issue_all();
  advance_entry();
  issue_all();
  advance_entry();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // entries 0 and 1 have landed
  asm volatile("" ::: "memory");

  int t = 0, cslot = 0;
  bf16x8 wA[4], wB[4];
  bf16x8 h0, h1;
#define MF_READW(buf, base, k)                                                     \
  buf[0] = *reinterpret_cast<const bf16x8*>((base) + (4 * (k)) * 1024);            \
  buf[1] = *reinterpret_cast<const bf16x8*>((base) + (4 * (k) + 1) * 1024);        \
  buf[2] = *reinterpret_cast<const bf16x8*>((base) + (4 * (k) + 2) * 1024);        \
  buf[3] = *reinterpret_cast<const bf16x8*>((base) + (4 * (k) + 3) * 1024)
  // time steps 0 and 1: nothing to consume yet; entries 2 and 3 go out.  Behind the second barrier the hidden fragments of number 0 and the first
  // weight fragments of time step 2 are fetched.
  for (; t < 2; ++t) {
    if (ld_e <= T + 1) {
      issue_all();
      advance_entry();
    }
    cslot = (cslot + 1 == MF_NSLOT) ? 0 : cslot + 1;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 - MF_PQ) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  {
    const unsigned char* hsrc = hb + (t & 1) * MF_HBUF + lane16;
    h0 = *reinterpret_cast<const bf16x8*>(hsrc);
    h1 = *reinterpret_cast<const bf16x8*>(hsrc + 1024);
    MF_READW(wA, smem + cslot * MF_ENTRY + MF_W1FR * 1024 + lane16, 0);
  }
  // stream-K hand-over (MfSeq): this workgroup's slot of `scratch` takes the accumulator of its head segment, the counter behind the slots
  // says when (MI355X_MICROARCH.md, hand-offs with sc1 loads in place of the acquire, third row: every storing wave adds to the counter
  // after its own vmcnt(0); the consumer polls with an sc1 load, passes a workgroup barrier, then reads with sc1 loads; whole 128-byte lines
  // per store instruction, 16-byte accesses)
  const __amdgpu_buffer_rsrc_t sk_out = __builtin_amdgcn_make_buffer_rsrc(scratch + (size_t)bid * MF_SK_SLOT, 0, MF_SK_SLOT, 0x00020000);
  // (scratch layout: G accumulator slots | the error record | MF_SK_SETS counter sets of G lines: a launch uses set `cset`, so that ONE memset
  // in front of a forward serves all of its launches)
  unsigned char* const sk_cnt0 = scratch + (size_t)G * MF_SK_SLOT + MF_SK_ERR + (size_t)cset * G * MF_SK_CNT;
  unsigned* const sk_cnt_out = reinterpret_cast<unsigned*>(sk_cnt0 + (size_t)bid * MF_SK_CNT);
  unsigned* const sk_cnt_in = reinterpret_cast<unsigned*>(sk_cnt0 + (size_t)(bid > 0 ? bid - 1 : 0) * MF_SK_CNT);
  // One time step t >= 2: consumes the hidden fragments of number t-2 (in h0, h1 on entry) and the W2 half of entry t: windows 0..4, the barrier,
  // then -- on registers -- window 5, under which the next step's hidden and weight fragments arrive.
  // fin: number t-2 was its segment's last step -- 1: the block is complete (epilogue), 2: a head segment (publish the accumulator);
  // load_next: the next segment is this workgroup's tail -- its accumulator comes from the previous workgroup.
  bool publish_pending = false;
  auto c_step = [&](const int fin, const bool load_next, int blk, int blk_next) __attribute__((always_inline)) {
    const bool epi = fin == 1;
    const unsigned char* slot = smem + cslot * MF_ENTRY + MF_W1FR * 1024 + lane16;
    const int nslot = (cslot + 1 == MF_NSLOT) ? 0 : cslot + 1;
    const unsigned char* slot_next = smem + nslot * MF_ENTRY + MF_W1FR * 1024 + lane16;
    MF_STAMP_DECL;
    MF_STAMP(0);
    const bool dma = ld_e <= T + 1;
#define MF_CMFMA(cur, k)                                                                                           \
  _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) {                                                               \
    acc[4 * (k) + ii][0] = MF_MFMA(cur[ii], h0, acc[4 * (k) + ii][0]);                                             \
    acc[4 * (k) + ii][1] = MF_MFMA(cur[ii], h1, acc[4 * (k) + ii][1]);                                             \
  }
#if defined(MF_DMA_SPREAD)
  /* a piece behind every MFMA pair whose index (0..19 over windows 0..4) crosses a multiple of 20 / (12 - MF_PQ) */           
#define MF_CWIN(cur, nxt, k)                                                                                       \
  do {                                                                                                             \
    MF_READW(nxt, slot, (k) + 1);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) {                                                             \
      acc[4 * (k) + ii][0] = MF_MFMA(cur[ii], h0, acc[4 * (k) + ii][0]);                                           \
      acc[4 * (k) + ii][1] = MF_MFMA(cur[ii], h1, acc[4 * (k) + ii][1]);                                           \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      const int hk_ = 4 * (k) + ii;                                                                                \
      if (((hk_ + 1) * (12 - MF_PQ)) / 20 > (hk_ * (12 - MF_PQ)) / 20) {                                           \
        if (dma) issue_piece_q((hk_ * (12 - MF_PQ)) / 20);                                                         \
      }                                                                                                            \
    }                                                                                                              \
  } while (0)
#else
#define MF_CWIN(cur, nxt, k)                                                                                       \
  do {                                                                                                             \
    MF_READW(nxt, slot, (k) + 1);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    MF_CMFMA(cur, k);                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    if (dma) {                                                                                                     \
      _Pragma("unroll") for (int q_ = MF_CPW * (k); q_ < MF_CPW * ((k) + 1); ++q_)                                 \
        if (q_ < 12 - MF_PQ) issue_piece_q(q_);                                                                    \
    }                                                                                                              \
  } while (0)
#endif
    MF_CWIN(wA, wB, 0);
    MF_CWIN(wB, wA, 1);
    MF_CWIN(wA, wB, 2);
    MF_CWIN(wB, wA, 3);
    MF_CWIN(wA, wB, 4);         // wB: window 5
#undef MF_CWIN
    if (dma) advance_entry();
    MF_STAMP(1);
    if (t <= T) {
      // entry t+1 has landed once only this step's pieces remain (an epilogue's stores, issued before them, are older); every LDS read of this
      // step is in registers (the ring slot and the hidden buffer may be rewritten behind the barrier)
      if (dma) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 - MF_PQ) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing younger went out: the count would retire nothing
      if (publish_pending) {
        // the 48 write-through stores of the head segment's accumulator (issued before this step's pieces) are done: tell the next workgroup
        if (lane == 0) __hip_atomic_fetch_add(sk_cnt_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        publish_pending = false;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    MF_STAMP(2);
    // behind the barrier: the next step's fragments are requested (entry t+1 and hidden number t-1: just published), window 5 runs on registers
    // (the hidden fragments only once window 5 has issued: 253 registers leave no room for two sets of them)
    MF_READW(wA, slot_next, 0);
    __builtin_amdgcn_sched_barrier(0);
    MF_CMFMA(wB, 5);
    __builtin_amdgcn_sched_barrier(0);
#undef MF_CMFMA
    {
      const unsigned char* hsrc = hb + ((t + 1) & 1) * MF_HBUF + lane16;
      h0 = *reinterpret_cast<const bf16x8*>(hsrc);
      h1 = *reinterpret_cast<const bf16x8*>(hsrc + 1024);
    }
    if (fin) {
      // RL: everything lane-dependent in this block is derived HERE from an opaque copy of the lane id -- as loop invariants (row and column
      // of the lane in three layouts, their products with the row pitch) they cost registers through the step loop, which has none to spare
      // (hipcc then reloads one of the loop's own addresses from scratch in every step, behind a vmcnt(0) that drains the DMA pieces)
      int lane_e = lane;
      if constexpr (RL) asm volatile("" : "+v"(lane_e));
      const int frow = lane_e & 15, fq = lane_e >> 4;
      const unsigned lane16 = (unsigned)lane_e * 16u;
      auto x_row_off = [&](int b_) __attribute__((always_inline)) { return ((unsigned)(b_ * MF_ROWS + pr * 32 + frow) * MF_D + 4u * fq) * 4u; };
      // The segment is complete.  fin == 1: the block is, its 32 x 384 fp32 accumulator leaves as bf16 rows.  A 16-row x 64-column slab goes
      // through 2 KiB of LDS -- the hidden buffer just read into h0/h1 (its next writer is the partner's step t+2, two barriers away) -- so
      // that every store covers whole 128-byte lines; swizzle and the one-ahead pipelining are those of gemm_bf16_pc's epilogue.
      // fin == 2 (a stream-K head segment): the fp32 accumulator goes to this workgroup's slot of the scratch (fragment (i, j) of wave w:
      // 1 KiB at ((w * 24 + i) * 2 + j) KiB, a lane's 16 bytes at 16 * lane: whole lines per instruction), write-through, then the counter.
      // Either way the accumulator registers then take their next values chunk by chunk, behind the last read of each: the bias image, or
      // (load_next) the previous workgroup's published accumulator -- ONE definition site for the 192 registers (three conditional ones made
      // hipcc keep two copies: 174 spills).
      unsigned char* stg = hb + ((t + 1) & 1) * MF_HBUF;
      const int rrow = lane_e >> 3, rch = lane_e & 7;
      unsigned voff_out = ((unsigned)(blk * MF_ROWS + pr * 32 + rrow) * MF_D + 8u * rch) * 2u;
      unsigned voff_sk = (unsigned)(pr * MF_NI * 2) * 1024u + lane16;
      unsigned voff_new = load_next ? voff_sk : (RL ? x_row_off(blk_next) : lane16);
      // opaque: computed here, once per block, instead of being hoisted out of the step loop as address registers;
      // and h0/h1 must have ARRIVED before the staging writes overwrite their source
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(voff_out), "+v"(voff_sk), "+v"(voff_new), "+v"(h0), "+v"(h1)::"memory");
      const __amdgpu_buffer_rsrc_t nsrc = __builtin_amdgcn_make_buffer_rsrc(
          load_next ? scratch + (size_t)(bid > 0 ? bid - 1 : 0) * MF_SK_SLOT
                    : (RL ? reinterpret_cast<unsigned char*>(rl.x) : const_cast<unsigned char*>(bias_img)),
          0, load_next ? MF_SK_SLOT : (RL ? (int)(2u * out_bytes) : MF_NI * 2 * 1024), 0x00020000);
      // where fragment (i, j) of the next accumulator lies behind voff_new: i * si + j * sj (the hand-over slot and the bias image: 2 KiB per i,
      // 1 KiB per j; RL's stream rows: 16 columns = 64 B per i, 16 rows per j)
      const int si = (RL && !load_next) ? 64 : 2048, sj = (RL && !load_next) ? 16 * MF_D * 4 : 1024;
      if constexpr (RL) {
        if (epi) {
          // The accumulator (stream row + the sum over the hidden units) + fc2's bias = the new stream row; its mean and variance over the
          // row (this lane's 96 values of each of its two rows, then the row's other three lanes: lane ^ 16, ^ 32), two passes as in
          // tr_norm.hip.  The parameter vectors come through asm loads, two 16-byte fragments in flight: hipcc hoists plain loads to the top
          // (24 x 4 registers beside the 192 of the accumulator: 273 spills, five accumulator fragments living in scratch through the step loop).
          // The three parameter vectors (fc2 bias, the norm's weight and bias: 1.5 KB each) are read per 16-column fragment in the accumulator's
          // lane layout.  Through global loads that is 72 dependent L2 round trips per block (measured: 27 us per block); so each C wave
          // copies them ONCE per block by LDS-DMA into 1-KiB pieces of the ring slot this step has consumed -- pieces pr + 4 k, which only
          // THIS wave's DMA of the next step writes again: k = 0, 1 the staging slab below, k = 2.. the vectors -- and reads fragments from LDS.
          unsigned char* const stgA = smem + cslot * MF_ENTRY + pr * 1024;
          MF_ESTAMP_DECL;
          MF_ESTAMP(0);
          if (t > T) {
            // The range's LAST time step has no barrier of its own (nothing is published behind it), so a C wave that is ahead would write
            // its vectors over W2 pieces 24 + pr and 28 + pr of THIS entry while a slower C wave has yet to read them: fragment pr / pr + 4 of
            // one step of the block went wrong in the OTHER waves, 1-4 % of the stream-K launches whose range ends in a one- or two-step
            // tail -- the waves leave the hand-over poll in front of it at different times (tools/lab/resid_ln_soak.py; found by a one-off
            // failure of test_mlp_fused_resid_ln[32896]).  One more barrier for all eight waves: the P waves run theirs before they return.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
          }
          {
            const unsigned ldsA = lds0 + cslot * MF_ENTRY + pr * 1024;
            mf_piece(reinterpret_cast<const unsigned char*>(rl.b2), lane16, ldsA + 2 * 4096);
            mf_piece(reinterpret_cast<const unsigned char*>(rl.g), lane16, ldsA + 4 * 4096);
            mf_piece(reinterpret_cast<const unsigned char*>(rl.b), lane16, ldsA + 6 * 4096);
            if (lane_e < 32) {          // floats 256..383
              mf_piece(reinterpret_cast<const unsigned char*>(rl.b2) + 1024, lane16, ldsA + 3 * 4096);
              mf_piece(reinterpret_cast<const unsigned char*>(rl.g) + 1024, lane16, ldsA + 5 * 4096);
              mf_piece(reinterpret_cast<const unsigned char*>(rl.b) + 1024, lane16, ldsA + 7 * 4096);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          MF_ESTAMP(1);
          const unsigned char* const tab = stgA + fq * 16;
          // fragment i of vector v (v = 2: fc2 bias, 4: weight, 6: bias): floats 16 i + 4 fq .. + 3
#define MF_TAB(v, i) (*reinterpret_cast<const f32x4*>(tab + ((v) + ((i) >> 4)) * 4096 + ((i) & 15) * 64))
#define MF_REP24(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23)
          float s0 = 0.f, s1 = 0.f;
          f32x4 nb2 = MF_TAB(2, 0);
          // (the sched_barriers keep hipcc from hoisting all 24 fragment reads to the top: 96 registers beside the accumulator's 192.  The
          // bias is added on the fly in each pass, NOT into the accumulator: 192 VALU-redefined registers between the step loop and the
          // norm cost 158 spilled registers.  And this file is compiled with -fno-slp-vectorize: the SLP vectoriser re-packs these scalar
          // chains into v_pk_* operations on register pairs it first has to assemble -- 86 spills, and every scratch reload waits on
          // vmcnt(0) behind the stores in flight: 44,000 cycles per block)
#define MF_RL_P1(i)                                                                                   \
  {                                                                                                   \
    const f32x4 bb = nb2;                                                                             \
    if ((i) + 1 < MF_NI) nb2 = MF_TAB(2, ((i) + 1) % MF_NI);                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    const f32x4 u0 = acc[i][0] + bb, u1 = acc[i][1] + bb;                                             \
    s0 += (u0[0] + u0[1]) + (u0[2] + u0[3]);                                                          \
    s1 += (u1[0] + u1[1]) + (u1[2] + u1[3]);                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
          MF_REP24(MF_RL_P1)
#undef MF_RL_P1
          MF_ESTAMP(2);
          s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16);
          s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32);
          const float mean0 = s0 / (float)MF_D, mean1 = s1 / (float)MF_D;
          float q0 = 0.f, q1 = 0.f;
          nb2 = MF_TAB(2, 0);
#define MF_RL_P2(i)                                                                                   \
  {                                                                                                   \
    const f32x4 bb = nb2;                                                                             \
    if ((i) + 1 < MF_NI) nb2 = MF_TAB(2, ((i) + 1) % MF_NI);                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
      const float d0 = (acc[i][0][e] + bb[e]) - mean0, d1 = (acc[i][1][e] + bb[e]) - mean1;           \
      q0 = __builtin_fmaf(d0, d0, q0);                                                                \
      q1 = __builtin_fmaf(d1, d1, q1);                                                                \
    }                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
          MF_REP24(MF_RL_P2)
#undef MF_RL_P2
          MF_ESTAMP(3);
          q0 += __shfl_xor(q0, 16); q1 += __shfl_xor(q1, 16);
          q0 += __shfl_xor(q0, 32); q1 += __shfl_xor(q1, 32);
          const float rstd0 = rsqrtf(q0 / (float)MF_D + rl.eps), rstd1 = rsqrtf(q1 / (float)MF_D + rl.eps);
          // norm1 of the next block, as bf16: two 16-column fragments x both row groups = a 32-row x 32-column slab (2 KiB: pieces k = 0, 1) go
          // through LDS so that a store instruction covers 64-byte row segments (two consecutive slabs complete the lines in L2).
          {
            const int r4 = lane_e >> 2, ch4 = lane_e & 3;
            unsigned vxn = ((unsigned)(blk * MF_ROWS + pr * 32 + r4) * MF_D + 8u * ch4) * 2u;
            asm volatile("" : "+v"(vxn));
            const unsigned char* rdp = stgA + r4 * 64 + ((ch4 ^ ((r4 >> 1) & 3)) << 4);
            unsigned vx0 = x_row_off(blk), vx1 = x_row_off(blk) + (unsigned)(16 * MF_D * 4);
            asm volatile("" : "+v"(vx0), "+v"(vx1));
            // (no read-ahead of the next fragment's weight / bias here: with the 192 accumulator registers, the statistics and the slab in flight
            // there is no room for a second pair -- hipcc then parks accumulator fragments in scratch THROUGH THE STEP LOOP)
#define MF_RL_P3(i)                                                                                                                  \
  {                                                                                                                                  \
    const f32x4 g = MF_TAB(4, i), bb = MF_TAB(6, i), b2f = MF_TAB(2, i);                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                                  \
      const float mean = j ? mean1 : mean0, rstd = j ? rstd1 : rstd0;                                                                \
      const f32x4 v = acc[i][j] + b2f;                                                                                               \
      /* the stream row goes back in place (64-byte row segments per instruction, two instructions complete a line in L2); offsets in */ \
      /* VGPRs / immediates, never an SGPR soffset on a 16-byte store (see below) */                                                 \
      MF_XSTORE(__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), xrsrc, (j ? vx1 : vx0) + (unsigned)((i) * 64), 0, 0)); \
      u32x2 pk2;                                                                                                                     \
      pk2[0] = pack_bf16x2((v[0] - mean) * rstd * g[0] + bb[0], (v[1] - mean) * rstd * g[1] + bb[1]);                                \
      pk2[1] = pack_bf16x2((v[2] - mean) * rstd * g[2] + bb[2], (v[3] - mean) * rstd * g[3] + bb[3]);                                \
      *reinterpret_cast<u32x2*>(stgA + j * 4096 + frow * 64 + (((2 * ((i) & 1) + (fq >> 1)) ^ ((frow >> 1) & 3)) << 4) + ((fq & 1) << 3)) = pk2; \
    }                                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                               \
    if ((i) & 1) {                                                                                                                   \
      const u32x4 l0 = *reinterpret_cast<const u32x4*>(rdp);                                                                         \
      MF_NSTORE(__builtin_amdgcn_raw_buffer_store_b128(l0, orsrc, vxn + (unsigned)(((i) >> 1) * 64), 0, 0));                         \
      __builtin_amdgcn_sched_barrier(0);                                                                                             \
      const u32x4 l1 = *reinterpret_cast<const u32x4*>(rdp + 4096);                                                                  \
      MF_NSTORE(__builtin_amdgcn_raw_buffer_store_b128(l1, orsrc, vxn + (unsigned)(((i) >> 1) * 64 + 16 * MF_D * 2), 0, 0));         \
      __builtin_amdgcn_sched_barrier(0);                                                                                             \
    }                                                                                                                                \
  }
            MF_ESTAMP(4);
            MF_REP24(MF_RL_P3)
            MF_ESTAMP(5);
            MF_ESTAMP_DUMP;
#undef MF_RL_P3
#undef MF_REP24
#undef MF_TAB
          }
        }
      }
      const unsigned char* rd = stg + rrow * 128 + ((rch ^ rrow) << 4);
      auto stage = [&](int c, int j) __attribute__((always_inline)) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = 4 * c + ii;
          const f32x4 v = acc[i][j];
          u32x2 pk2;
          pk2[0] = pack_bf16x2(v[0], v[1]);
          pk2[1] = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<u32x2*>(stg + frow * 128 + (((2 * ii + (fq >> 1)) ^ (frow & 7)) << 4) + (((fq & 1) ^ (frow >> 3)) << 3)) = pk2;
        }
      };
      auto read_back = [&](u32x4 (&ln)[2]) __attribute__((always_inline)) {
        ln[0] = *reinterpret_cast<const u32x4*>(rd);
        const u32x4 tt = *reinterpret_cast<const u32x4*>(rd + 1024);
        ln[1] = u32x4{tt[2], tt[3], tt[0], tt[1]};
      };
      // rows beyond M need no predicate: num_records = M * 768 bytes, so their offsets fail the descriptor's bounds check and the store is dropped.
      // The (row group, column chunk) part of the offset is added in a VGPR, NOT passed as an SGPR soffset: with a register soffset hipcc
      // (ROCm 7.2) omits the wait state between a 16-byte buffer store and a VALU write of its data registers, and on gfx950 that write then
      // races the store's operand read -- single dwords of single lanes of the stored line came out as the NEXT line's (tools/mlp_lab.py
      // --stress: 193 of 200 launches differed from the two-GEMM pair, always in the same rows of a slab; profiles/r05_mlp_lab.md).
      auto store = [&](int c, int j, const u32x4 (&ln)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
          __builtin_amdgcn_raw_buffer_store_b128(ln[r], orsrc, voff_out + (unsigned)(((16 * j + 8 * r) * MF_D + 64 * c) * 2), 0, 0);
      };
      u32x4 lnA[2], lnB[2];
      if (!epi) {
#pragma unroll
        for (int i = 0; i < MF_NI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), sk_out, voff_sk + (unsigned)((i * 2 + j) * 1024), 0,
                                                   16 /* sc1: write-through */);
        // The counter is raised once the stores are known to be done.  Normally that is deferred to the next step's DMA wait; when this
        // workgroup's own tail comes next it is done HERE, before the poll below: a workgroup that waited for its predecessor before
        // publishing would chain the launch's workgroups one behind the other (ranges barely longer than a block: head, then tail, nothing
        // between), and a tail of one step would leave no later step to raise the counter in.
        if (load_next) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(sk_cnt_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          publish_pending = true;
        }
      }
      if (load_next) {
        // the previous workgroup's four C waves have published their parts (it ran that segment first: normally long ago); every wave that
        // reads polls for itself.  The predecessor was dispatched BEFORE this workgroup and publishes after its own first segment without
        // waiting for anybody, so the wait is bounded by dispatch skew (also when the workgroups run in several rounds: a CU mask, a
        // partitioned device); the spin is bounded all the same (poll_max: seconds), and a poll that runs out is RECORDED
        bool published = false;
        for (int spin = 0; spin < poll_max; ++spin) {
          if (__hip_atomic_load(sk_cnt_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 4u) { published = true; break; }
          __builtin_amdgcn_s_sleep(8);
        }
        if (!published && lane == 0) {
          // the launch goes on (on whatever the slot holds) and SAYS so: the error record behind the counters -- a magic word and the
          // workgroup -- is what tr_mlp_fused_status / tr_vit_forward_status turn into TR_ERR_LAUNCH at the caller's next status check.
          // (A trap here would take the whole process down, also under a CU mask or GPU sharing where the wait is merely long.)
          unsigned* err = reinterpret_cast<unsigned*>(scratch + (size_t)G * MF_SK_SLOT);
          int who = bid + 1;
          asm volatile("" : "+s"(who));        // formed HERE: as a loop invariant it is parked in a VGPR through the step loop, which has none to spare (one spill)
          __hip_atomic_store(err + 1, (unsigned)who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(err, MF_SK_ERR_MAGIC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("" ::: "memory");
      }
      if (!RL && epi) { stage(0, 0); read_back(lnA); }
#pragma unroll
      for (int c = 0; c < MF_NI / 4; ++c) {
        if (!RL && epi) {
          // pass (c, 0) is staged and being read back into lnA on entry
          stage(c, 1); store(c, 0, lnA); read_back(lnB);
          if (c + 1 < MF_NI / 4) stage(c + 1, 0);
          store(c, 1, lnB);
          if (c + 1 < MF_NI / 4) read_back(lnA);
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[4 * c + ii][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(nsrc, voff_new, (4 * c + ii) * si + j * sj,
                                                                                                 16 /* sc1: past this CU's L1 (the hand-over read) */));
      }
      if constexpr (RL) {
        // the next step's first weight fragments and hidden fragments are fetched AGAIN here: the copies requested above are dead on this path,
        // which frees their 24 registers for the norm's parameters and statistics (ring slot and hidden buffer stay valid through the next step)
        MF_READW(wA, slot_next, 0);
        const unsigned char* hsrc = hb + ((t + 1) & 1) * MF_HBUF + lane16;
        h0 = *reinterpret_cast<const bf16x8*>(hsrc);
        h1 = *reinterpret_cast<const bf16x8*>(hsrc + 1024);
      }
    }
    MF_STAMP(3);
    MF_STAMP_DUMP(1, t);
    ++t;
    cslot = nslot;
  };
  {
    MfCur cur = mf_first(q, NS);                // flat loop, opaque flags: see the P loop
    for (int tt = 0; tt < T; ++tt) {
      const MfCur nxt = mf_next(q, cur, NS);
      const bool last = mf_last_of_segment(q, cur, NS);
      int flags = __builtin_amdgcn_readfirstlane((last ? (cur.seg == -1 ? 2 : 1) : 0) | ((last && tt + 1 < T && nxt.seg == q.nfull) ? 4 : 0));
      asm volatile("" : "+s"(flags));
      c_step(flags & 3, (flags & 4) != 0, mf_block(q, cur), mf_block(q, nxt));
      cur = nxt;
    }
  }
#undef MF_READW
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (publish_pending && lane == 0) __hip_atomic_fetch_add(sk_cnt_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (a head is never a range's last segment)
}

}  // namespace

extern "C" int tr_mlp_fused_supported(int D, int Hd) { return (D == MF_D && Hd % 32 == 0 && Hd >= 64) ? 1 : 0; }

// Workgroups of a launch = compute units of the CURRENT device (one persistent 512-thread workgroup per CU: all of its LDS, half its
// registers), read once per device -- not a literal 256: a partitioned or smaller device gets its own grid, scratch size and thresholds.
// TR_MLP_FUSED_GRID=n (lab / tests): another grid, e.g. to run the stream-K hand-over in several rounds of workgroups.
static int mf_grid() {
  static const int forced = [] { const char* e = getenv("TR_MLP_FUSED_GRID"); return e ? atoi(e) : 0; }();
  if (forced > 0) return forced;
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int g = cached[dev].load(std::memory_order_relaxed);
  if (g == 0) {
    int n = 0;
    g = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    cached[dev].store(g, std::memory_order_relaxed);
  }
  return g;
}

static std::atomic<int> g_mlp_fused_mode{-1};
static std::atomic<int> g_mlp_sk_min_blocks{0};       // lab (tr_set_mlp_fused(mode >= 2)): fewest blocks for which auto takes the stream-K launch (0: grid + 1)
static std::atomic<int> g_mlp_poll_max{1 << 24};      // hand-over poll bound (iterations of an 8-tick sleep: seconds); tests shorten it
extern "C" int tr_set_mlp_fused(int mode) {
  if (mode >= 2) { g_mlp_sk_min_blocks.store(mode); mode = -1; }
  return g_mlp_fused_mode.exchange(mode < 0 ? -1 : (mode > 0 ? 1 : 0));
}
// the executor's question (tr_vit.hip): run the fused launch for M rows?  A block is a chain of Hd / 32 sequential steps.  With the stream-K
// scratch and more blocks than workgroups the steps are dealt evenly (no tail round): always; else the launch is ONE round however few blocks
// it holds: where they fill at least three quarters of the chip.
// concurrent: the forward runs beside others (tr_vit_config.concurrent): an underfilled round costs nothing then -- always.
int tr_mlp_fused_wanted(int M, int D, int Hd, int have_scratch, int concurrent) {
  if (!tr_mlp_fused_supported(D, Hd)) return 0;
  const int mode = g_mlp_fused_mode.load(std::memory_order_relaxed);
  if (mode >= 0) return mode;
  if (concurrent) return 1;
  const int nblk = (M + MF_ROWS - 1) / MF_ROWS, G = mf_grid();
  if (nblk > G) {
    const int lo = g_mlp_sk_min_blocks.load(std::memory_order_relaxed);
    if (have_scratch) return nblk >= (lo > 0 ? lo : G + 1);
    const int rounds = (nblk + G - 1) / G;
    return 4 * nblk >= 3 * G * rounds;
  }
  return 4 * nblk >= 3 * G;
}

// default OFF: measured in the model (tools/lab/mlp_model_ab.sh) the fused tail LOSES 4 % of the headline forward against the fused Mlp + the
// LayerNorm launch -- its epilogue holds the workgroup for ~22,000 cycles per block (profiles/r05_mlp_lab.md)
static std::atomic<int> g_mlp_resid_ln{0};
extern "C" int tr_set_mlp_resid_ln(int on) { return g_mlp_resid_ln.exchange(on ? 1 : 0); }
int tr_mlp_resid_ln_enabled() { return g_mlp_resid_ln.load(std::memory_order_relaxed); }

// norm2 inside the fused Mlp launch (tr_mlp_fused_ln_bf16) where the eval executor would run a lazy norm2 followed by the fused Mlp.
// 1 (default): where the launch is ONE round of whole blocks -- measured (tools/lab/mlp_ln_ab.py, profiles/r06_lab.md): 88.5 vs 93.0 us at
// 24,832 rows, but under the stream-K schedule every workgroup normalises every block it touches (two or three for 1.1-1.5 blocks of
// work) and the ~10 us a 128-row prologue holds the workgroup cost more than the LayerNorm launch saved (131 vs 121 us at 35,328 rows);
// 2 (lab): wherever the fused Mlp runs; 0: never.
static std::atomic<int> g_mlp_ln{1};
extern "C" int tr_set_mlp_ln(int mode) { return g_mlp_ln.exchange(mode < 0 ? 0 : (mode > 2 ? 2 : mode)); }
int tr_mlp_ln_wanted(int M, int D, int Hd, int have_scratch, int concurrent) {
  const int mode = g_mlp_ln.load(std::memory_order_relaxed);
  if (mode == 0 || !tr_mlp_fused_wanted(M, D, Hd, have_scratch, concurrent)) return 0;
  // one-round launches, and every launch of a forward that runs beside others: those run whole blocks (no stream-K), so each block is
  // normalised once, by the workgroup that computes it (profiles/r06_inflight_lab.md: +0.8 % with two forwards in flight)
  return mode == 2 || concurrent || (M + MF_ROWS - 1) / MF_ROWS <= mf_grid();
}

extern "C" int tr_set_mlp_poll_max(int iterations) { return g_mlp_poll_max.exchange(iterations >= 0 ? iterations : (1 << 24)); }      // 0 (tests): every hand-over is reported as abandoned

extern "C" size_t tr_mlp_pack_bytes(int D, int Hd) {
  if (D <= 0 || Hd <= 0 || D % 32 || Hd % 32) return 0;
  return ((size_t)(Hd / 32) * (size_t)(2 * (D / 32) + D / 16) + (size_t)(2 * (D / 16))) * 1024;       // the steps' fragments + the bias image
}

extern "C" size_t tr_mlp_fused_scratch_bytes(int D, int Hd) {
  return tr_mlp_fused_supported(D, Hd) ? (size_t)mf_grid() * (MF_SK_SLOT + (size_t)MF_SK_SETS * MF_SK_CNT) + MF_SK_ERR : 0;
}

extern "C" int tr_mlp_pack_bf16(const uint16_t* fc1_w, const uint16_t* fc2_w, const float* fc2_b, void* packed, int D, int Hd, tr_stream_t s) {
  TR_REQUIRE(fc1_w && fc2_w && fc2_b && packed, TR_ERR_NULL, "tr_mlp_pack_bf16: null pointer");
  TR_REQUIRE(D > 0 && Hd > 0 && D % 32 == 0 && Hd % 32 == 0, TR_ERR_SHAPE, "tr_mlp_pack_bf16: D=%d and Hd=%d must be multiples of 32", D, Hd);
  TR_REQUIRE(tr_aligned16(fc1_w) && tr_aligned16(fc2_w) && tr_aligned16(fc2_b) && tr_aligned16(packed), TR_ERR_ALIGN,
             "tr_mlp_pack_bf16: pointers must be 16-byte aligned");
  const int frags = (Hd / 32) * (2 * (D / 32) + D / 16) + 2 * (D / 16);
  hipLaunchKernelGGL(mlp_pack_kernel, dim3((frags + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(s), fc1_w, fc2_w, fc2_b, static_cast<u32x4*>(packed), D,
                     Hd);
  TR_CHECK_LAUNCH("tr_mlp_pack_bf16");
  return TR_OK;
}

// cset >= 0: the launch uses counter set cset, which the caller has zeroed (tr_mlp_fused_zero_counters: one memset for a whole forward);
// cset < 0: set 0, zeroed here by a memset node in front of the kernel
static int mlp_fused_launch(const char* who, const uint16_t* xn, const void* packed, const float* fc1_b, uint16_t* out, void* scratch,
                            size_t scratch_bytes, int M, int D, int Hd, const MfResid* rl, const MfNorm* nm, tr_stream_t s, int cset = -1) {
  TR_REQUIRE(cset < MF_SK_SETS, TR_ERR_CONFIG, "%s: counter set %d of %d", who, cset, MF_SK_SETS);
  TR_REQUIRE((xn || nm) && packed && fc1_b && out, TR_ERR_NULL, "%s: null pointer", who);
  TR_REQUIRE(M > 0 && tr_mlp_fused_supported(D, Hd), TR_ERR_SHAPE, "%s: unsupported shape M=%d D=%d Hd=%d (D must be %d, Hd %% 32 == 0)", who, M, D, Hd,
             MF_D);
  TR_REQUIRE(tr_aligned16(xn) && tr_aligned16(packed) && tr_aligned16(fc1_b) && tr_aligned16(out) && tr_aligned16(scratch),
             TR_ERR_ALIGN, "%s: pointers must be 16-byte aligned", who);
  TR_REQUIRE(scratch == nullptr || scratch_bytes >= tr_mlp_fused_scratch_bytes(D, Hd), TR_ERR_SHAPE,
             "%s: scratch of %zu bytes, tr_mlp_fused_scratch_bytes says %zu (or pass NULL: whole-block schedule)", who, scratch_bytes,
             tr_mlp_fused_scratch_bytes(D, Hd));
  const size_t out_bytes = (size_t)M * D * 2;
  TR_REQUIRE(out_bytes < ((size_t)1 << (rl ? 30 : 31)), TR_ERR_SHAPE, "%s: %zu output bytes exceed the range of the 32-bit store offsets", who, out_bytes);
  if (rl)       // + the stream's read-modify-write and the norm's parameters; no separate fc2 output
    tr_prof_note("mlp_fused_kernel<resid_ln>", 4.0 * M * D * Hd, 2.0 * M * D + 8.0 * M * D + 2.0 * M * D + 4.0 * D * Hd);
  else if (nm)  // the fp32 stream row + the bf16 pending residual in, bf16 out, the packed weights
    tr_prof_note("mlp_fused_kernel<ln>", 4.0 * M * D * Hd, 6.0 * M * D + 2.0 * M * D + 4.0 * D * Hd);
  else
    tr_prof_note("mlp_fused_kernel", 4.0 * M * D * Hd, 4.0 * M * D + 4.0 * D * Hd);
  const int nblk = (M + MF_ROWS - 1) / MF_ROWS, grid = mf_grid();
  const int G = nblk < grid ? nblk : grid;
  hipStream_t st = static_cast<hipStream_t>(s);
  unsigned char* sk = (nblk > G) ? static_cast<unsigned char*>(scratch) : nullptr;
  if (sk != nullptr && cset < 0) {
    // the hand-over counters of this launch (one line per workgroup, behind the accumulator slots) start at zero: a memset node ahead of the
    // kernel node (graph-capturable; a counter that the last consumer reset would fail a first, poisoned launch).  The error record in
    // front of them is NOT touched: it stays until tr_mlp_fused_status reads it.
    hipError_t e = hipMemsetAsync(sk + (size_t)G * MF_SK_SLOT + MF_SK_ERR, 0, (size_t)G * MF_SK_CNT, st);
    TR_REQUIRE(e == hipSuccess, TR_ERR_LAUNCH, "%s: hipMemsetAsync: %s", who, hipGetErrorString(e));
  }
  const int set = cset < 0 ? 0 : cset;
  const int poll_max = g_mlp_poll_max.load(std::memory_order_relaxed);
  const MfResid no_rl{nullptr, nullptr, nullptr, nullptr, 0.f};
  const MfNorm no_nm{nullptr, nullptr, nullptr, nullptr, 0.f};
  if (rl)
    hipLaunchKernelGGL((mlp_fused_kernel<true, false>), dim3(G), dim3(512), 0, st, xn, static_cast<const unsigned char*>(packed), fc1_b, out, sk, M,
                       Hd / 32, (unsigned)out_bytes, poll_max, set, *rl, no_nm);
  else if (nm)
    hipLaunchKernelGGL((mlp_fused_kernel<false, true>), dim3(G), dim3(512), 0, st, xn, static_cast<const unsigned char*>(packed), fc1_b, out, sk, M,
                       Hd / 32, (unsigned)out_bytes, poll_max, set, no_rl, *nm);
  else
    hipLaunchKernelGGL((mlp_fused_kernel<false, false>), dim3(G), dim3(512), 0, st, xn, static_cast<const unsigned char*>(packed), fc1_b, out, sk, M,
                       Hd / 32, (unsigned)out_bytes, poll_max, set, no_rl, no_nm);
  TR_CHECK_LAUNCH(who);
  return TR_OK;
}

extern "C" int tr_mlp_fused_bf16(const uint16_t* xn, const void* packed, const float* fc1_b, uint16_t* out, void* scratch, size_t scratch_bytes, int M,
                                 int D, int Hd, tr_stream_t s) {
  TR_REQUIRE(xn, TR_ERR_NULL, "tr_mlp_fused_bf16: null pointer");
  return mlp_fused_launch("tr_mlp_fused_bf16", xn, packed, fc1_b, out, scratch, scratch_bytes, M, D, Hd, nullptr, nullptr, s);
}

// topk.py:95's `self.mlp(self.norm2(x))` in ONE launch:  out = fc2(gelu(fc1(LayerNorm(x + delta; g, b, eps))))  with x the fp32 stream and
// delta the pending bf16 residual of the attention branch (neither is written).  Bit-identical to tr_layernorm2_bf16(x, NULL, delta, NULL, ..)
// followed by tr_mlp_fused_bf16.
extern "C" int tr_mlp_fused_ln_bf16(const float* x, const uint16_t* delta, const float* g, const float* b, float eps, const void* packed,
                                    const float* fc1_b, uint16_t* out, void* scratch, size_t scratch_bytes, int M, int D, int Hd, tr_stream_t s) {
  TR_REQUIRE(x && delta && g && b, TR_ERR_NULL, "tr_mlp_fused_ln_bf16: null pointer");
  TR_REQUIRE(tr_aligned16(x) && tr_aligned16(delta) && tr_aligned16(g) && tr_aligned16(b), TR_ERR_ALIGN,
             "tr_mlp_fused_ln_bf16: pointers must be 16-byte aligned");
  const MfNorm nm{x, delta, g, b, eps};
  return mlp_fused_launch("tr_mlp_fused_ln_bf16", nullptr, packed, fc1_b, out, scratch, scratch_bytes, M, D, Hd, nullptr, &nm, s);
}

// The tail of a transformer block and the head of the next in one launch (topk.py:95 `x = x + self.mlp(self.norm2(x))`, then the next
// block's :87 `self.norm1(x)`):  x += fc2(gelu(fc1(xn))) + fc2_b  in place on the fp32 stream,  xn_next = LayerNorm(x; next_g, next_b, eps)
// as bf16.  x must already hold the attention branch's residual.  xn_next may not alias xn.
extern "C" int tr_mlp_fused_resid_ln_bf16(const uint16_t* xn, const void* packed, const float* fc1_b, const float* fc2_b, float* x,
                                          const float* next_g, const float* next_b, float eps, uint16_t* xn_next, void* scratch,
                                          size_t scratch_bytes, int M, int D, int Hd, tr_stream_t s) {
  TR_REQUIRE(xn && fc2_b && x && next_g && next_b, TR_ERR_NULL, "tr_mlp_fused_resid_ln_bf16: null pointer");
  TR_REQUIRE(tr_aligned16(fc2_b) && tr_aligned16(x) && tr_aligned16(next_g) && tr_aligned16(next_b), TR_ERR_ALIGN,
             "tr_mlp_fused_resid_ln_bf16: pointers must be 16-byte aligned");
  TR_REQUIRE(xn_next != xn, TR_ERR_CONFIG, "tr_mlp_fused_resid_ln_bf16: xn_next aliases xn");
  const MfResid rl{x, fc2_b, next_g, next_b, eps};
  return mlp_fused_launch("tr_mlp_fused_resid_ln_bf16", xn, packed, fc1_b, xn_next, scratch, scratch_bytes, M, D, Hd, &rl, nullptr, s);
}

// The status check of the stream-K hand-over: waits for the stream, reads the error record behind the scratch's counters and clears it.
// TR_OK, or TR_ERR_LAUNCH when a workgroup's poll for its predecessor's accumulator ran out in some launch since the last check (the
// outputs of that launch are not valid).  A scratch that never saw such a launch holds no magic word, whatever else it holds.
extern "C" int tr_mlp_fused_status(void* scratch, size_t scratch_bytes, int D, int Hd, tr_stream_t s) {
  TR_REQUIRE(scratch, TR_ERR_NULL, "tr_mlp_fused_status: null pointer");
  TR_REQUIRE(tr_mlp_fused_supported(D, Hd) && scratch_bytes >= tr_mlp_fused_scratch_bytes(D, Hd), TR_ERR_SHAPE,
             "tr_mlp_fused_status: scratch of %zu bytes, tr_mlp_fused_scratch_bytes says %zu", scratch_bytes, tr_mlp_fused_scratch_bytes(D, Hd));
  hipStream_t st = static_cast<hipStream_t>(s);
  unsigned char* rec = static_cast<unsigned char*>(scratch) + (size_t)mf_grid() * MF_SK_SLOT;
  unsigned host[2] = {0u, 0u};
  hipError_t e = hipMemcpyAsync(host, rec, sizeof(host), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  TR_REQUIRE(e == hipSuccess, TR_ERR_LAUNCH, "tr_mlp_fused_status: %s", hipGetErrorString(e));
  if (host[0] != MF_SK_ERR_MAGIC || host[1] == 0u || host[1] > 65536u) return TR_OK;
  e = hipMemsetAsync(rec, 0, MF_SK_ERR, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  TR_REQUIRE(false, TR_ERR_LAUNCH,
             "fused Mlp: workgroup %u waited in vain for its predecessor's accumulator (stream-K hand-over poll ran out): the outputs of that "
             "launch are invalid", host[1] - 1u);
  return TR_ERR_LAUNCH;
}

// lab (tools/lab/clock_probe.py): read and reset the in-kernel clock probe of a -DTR_DIAG_CLOCK build: out[3] = {shader cycles, 100-MHz ticks,
// launches} of mlp_fused_kernel's workgroup 8; all zero in a product build.
extern "C" int tr_mlp_clock_probe_read(unsigned long long* out) {
  TR_REQUIRE(out, TR_ERR_NULL, "tr_mlp_clock_probe_read: null pointer");
  const unsigned long long zero[3] = {0ull, 0ull, 0ull};
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(tr_mlp_clock_probe), sizeof(zero));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(tr_mlp_clock_probe), zero, sizeof(zero));
  TR_REQUIRE(e == hipSuccess, TR_ERR_LAUNCH, "tr_mlp_clock_probe_read: %s", hipGetErrorString(e));
  return TR_OK;
}

// ---- the executor's forms (tr_vit.hip; C++ linkage, not part of the C ABI): ONE memset in front of a forward zeroes the counter sets of all
// its fused-Mlp launches (launch i uses set i) instead of a memset node in front of every launch -- five nodes of ~5 us fewer in the
// headline forward.  nsets <= MF_SK_SETS (= TR_MAX_DEPTH).
int tr_mlp_fused_zero_counters(void* scratch, size_t scratch_bytes, int D, int Hd, int nsets, tr_stream_t s) {
  TR_REQUIRE(scratch && scratch_bytes >= tr_mlp_fused_scratch_bytes(D, Hd) && nsets >= 1 && nsets <= MF_SK_SETS, TR_ERR_SHAPE,
             "tr_mlp_fused_zero_counters: scratch of %zu bytes / %d sets", scratch_bytes, nsets);
  const int G = mf_grid();
  hipError_t e = hipMemsetAsync(static_cast<unsigned char*>(scratch) + (size_t)G * MF_SK_SLOT + MF_SK_ERR, 0, (size_t)nsets * G * MF_SK_CNT,
                                static_cast<hipStream_t>(s));
  TR_REQUIRE(e == hipSuccess, TR_ERR_LAUNCH, "tr_mlp_fused_zero_counters: hipMemsetAsync: %s", hipGetErrorString(e));
  return TR_OK;
}
int tr_mlp_fused_bf16_set(const uint16_t* xn, const void* packed, const float* fc1_b, uint16_t* out, void* scratch, size_t scratch_bytes, int M, int D,
                          int Hd, int cset, tr_stream_t s) {
  TR_REQUIRE(xn, TR_ERR_NULL, "tr_mlp_fused_bf16: null pointer");
  return mlp_fused_launch("tr_mlp_fused_bf16", xn, packed, fc1_b, out, scratch, scratch_bytes, M, D, Hd, nullptr, nullptr, s, cset);
}
int tr_mlp_fused_ln_bf16_set(const float* x, const uint16_t* delta, const float* g, const float* b, float eps, const void* packed, const float* fc1_b,
                             uint16_t* out, void* scratch, size_t scratch_bytes, int M, int D, int Hd, int cset, tr_stream_t s) {
  TR_REQUIRE(x && delta && g && b, TR_ERR_NULL, "tr_mlp_fused_ln_bf16: null pointer");
  const MfNorm nm{x, delta, g, b, eps};
  return mlp_fused_launch("tr_mlp_fused_ln_bf16", nullptr, packed, fc1_b, out, scratch, scratch_bytes, M, D, Hd, nullptr, &nm, s, cset);
}
int tr_mlp_fused_resid_ln_bf16_set(const uint16_t* xn, const void* packed, const float* fc1_b, const float* fc2_b, float* x, const float* next_g,
                                   const float* next_b, float eps, uint16_t* xn_next, void* scratch, size_t scratch_bytes, int M, int D, int Hd,
                                   int cset, tr_stream_t s) {
  TR_REQUIRE(xn && fc2_b && x && next_g && next_b && xn_next != xn, TR_ERR_NULL, "tr_mlp_fused_resid_ln_bf16: null or aliased pointer");
  const MfResid rl{x, fc2_b, next_g, next_b, eps};
  return mlp_fused_launch("tr_mlp_fused_resid_ln_bf16", xn, packed, fc1_b, xn_next, scratch, scratch_bytes, M, D, Hd, &rl, nullptr, s, cset);
}
