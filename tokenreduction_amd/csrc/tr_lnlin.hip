// LayerNorm + Linear(384 -> N) in ONE launch for gfx950:  the start of a transformer block as the reference writes it
// (topk.py:86-87 `self.attn(self.norm1(x))`, :44 `qkv = self.qkv(x)`), with the block's pending residual adds folded in:
//
//     v = (x [+ d1]) [+ d2]          fp32 stream row + the previous block's pending bf16 residuals, the reference's order (topk.py:87, :95)
//     x_out = v                       written once (out of place), if anything was pending
//     out = bf16( LayerNorm(v; g, b, eps) . W^T + bias )
//
// Replaces tr_layernorm[2]_bf16 (12-14 B per element through HBM at the memory roof, 18 % of the headline forward) + tr_gemm_bf16 and is
// BIT-IDENTICAL to that pair: the norm is layernorm_half_kernel's arithmetic in its lane layout (its statements, contractions spelled out),
// the product is gemm_bf16_pc's (same MFMA, accumulators that start at the bias, K in the same 32-deep steps in the same order).
//
// Structure (one persistent 768-thread workgroup per CU, 3 waves per SIMD, 168 VGPRs; D = 384 only):
//   * 8 MFMA waves = 4 row slices of 32 rows x 2 feature halves.  Like the fused Mlp's fc1 waves (tr_mlp_fused.hip) a wave keeps ITS
//     32 x 384 normalised rows in registers as the MFMA B operand (96 VGPRs) and computes out^T = W_step . xn^T for 32 output features per
//     step, W rows fed in the order that makes a lane's eight results eight CONSECUTIVE features of one token: finished outputs, no
//     accumulator across steps, no epilogue tile -- bias is the accumulators' start value, two 16-byte stores per step.
//   * 4 LN waves normalise the rows of the workgroup's NEXT 128-row block while the MFMA waves multiply the current one -- half a wave
//     per row, coalesced 512-byte loads, the stream row written back once --, 16 rows per step, loads two steps ahead (four row pairs in
//     flight per wave).  The normalised bf16 rows go through a workgroup-private 2 x 96 KiB slot in global memory (L2-resident) to the
//     MFMA waves, which refill their B registers IN PLACE behind the last MFMAs of a block.  The MFMA waves feed the weight ring (LDS-DMA,
//     six pieces each per step).
//   * Weights: fragment-major packed copy (tr_lnlin_pack_bf16), 48-KiB entries = the two feature halves' 24 fragments of one step,
//     3-slot LDS ring, one s_barrier per step placed before the step's last k-window.
//   * Schedule: the launch's nblk * (N / 64) steps are cut into equal contiguous ranges, one per workgroup.  A step's outputs are final,
//     so a block that straddles two workgroups needs no hand-over -- only its LayerNorm is computed by both (the stream row is written by the
//     workgroup that owns the block's first step: exactly one writer per row).
#include "tr_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int LL_D = 384;
constexpr int LL_KS = LL_D / 32;                 // 12 k-steps of 32
constexpr int LL_FRH = 2 * LL_KS;                // 24 fragments of one 32-feature step
constexpr int LL_ENTRY = 2 * LL_FRH * 1024;      // 48 KiB: both feature halves of a step
constexpr int LL_NSLOT = 3;
constexpr int LL_BIAS_MAX = 3072;                // output features whose bias fits behind the ring (12 KiB) ...
constexpr int LL_GB_OFF = LL_NSLOT * LL_ENTRY + LL_BIAS_MAX * 4;      // ... in front of the LayerNorm weight and bias (2 x 1.5 KiB)
constexpr int LL_LDS = LL_NSLOT * LL_ENTRY + 16384;                // 147,456 + 16,384 = 163,840 B: all of the CU's LDS
constexpr int LL_ROWS = 128;
constexpr int LL_SLOT_BYTES = LL_ROWS * LL_D * 2;      // one block of normalised rows: 96 KiB

#ifdef TR_ABLATE_NO_MFMA
#define LL_MFMA(a, b, c) ([&] { asm volatile("" ::"v"(a), "v"(b)); return c; }())
#else
#define LL_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif

// diagnostic build only (-DTR_DIAG_STAMPS, tools/lab/lnlin_lab.py --stamps): s_memtime stamps of workgroup LL_DIAG_WG, MFMA wave 0 and LN wave 8,
// into the 64 KiB behind the scratch's row slots: [role][step 0..62 | 63: prologue][4]
#ifndef LL_DIAG_WG
#define LL_DIAG_WG 8
#endif
#ifdef TR_DIAG_STAMPS
#define LL_STAMP_DECL unsigned long long ts_[4] = {0, 0, 0, 0}
#define LL_STAMP(k) ts_[k] = __builtin_amdgcn_s_memtime()
#define LL_STAMP_DUMP(role, step)                                                                                      \
  do {                                                                                                                 \
    if (bid == LL_DIAG_WG && lane == 0 && (step) < 64) {                                                               \
      unsigned long long* st_ = reinterpret_cast<unsigned long long*>(a.scratch + (size_t)G * (2 * LL_SLOT_BYTES)) + ((role) * 64 + (step)) * 4; \
      st_[0] = ts_[0]; st_[1] = ts_[1]; st_[2] = ts_[2]; st_[3] = ts_[3];                                              \
    }                                                                                                                  \
  } while (0)
#else
#define LL_STAMP_DECL do { } while (0)
#define LL_STAMP(k) do { } while (0)
#define LL_STAMP_DUMP(role, step) do { } while (0)
#endif

__device__ __forceinline__ void ll_piece(const unsigned char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %[ld]\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o], %[b]"
      :
      : [o] "v"(voff), [b] "s"(sbase), [ld] "s"(lds_dst)
      : "memory", "m0");
}
// the MFMA waves' B fragments from the workgroup's slot: sc1 = past this CU's L1 (the slot's lines are rewritten every other block)
template <int IMM>
__device__ __forceinline__ bf16x8 ll_load_x(const unsigned char* sbase, unsigned voff) {
  bf16x8 v;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 sc1" : "=v"(v) : "v"(voff), "s"(sbase), "i"(IMM) : "memory");
  return v;
}
template <int IMM>
__device__ __forceinline__ void ll_reload_x(bf16x8& v, const unsigned char* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 sc1" : "+v"(v) : "v"(voff), "s"(sbase), "i"(IMM) : "memory");
}

#define LL_TIE_X(x)                                                                                                                   \
  "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), \
      "+v"(x[11])

// Fragment-major packing of W [N, 384] (nn.Linear layout, bf16).  pk[entry e][half h][fragment f][lane][8 bf16], NE = N / 64 entries:
//   step s = e + NE * h  (half 0 serves features [0, N/2), half 1 the rest: a wave's consecutive steps write consecutive 64-byte segments),
//   fragment f: tile t2 = f / 12, k-step ks = f % 12;  lane (r = l & 15, q = l >> 4) holds W[32 s + 8 (r >> 2) + 4 t2 + (r & 3)][32 ks + 8 q ..]
__global__ __launch_bounds__(256) void lnlin_pack_kernel(const uint16_t* __restrict__ W, u32x4* __restrict__ pk, int N) {
  const int NE = N / 64;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);          // (entry, half, fragment)
  if (g >= NE * 2 * LL_FRH) return;
  const int lane = threadIdx.x & 63;
  const int e = g / (2 * LL_FRH), h = (g / LL_FRH) & 1, f = g % LL_FRH;
  const int s = e + NE * h, t2 = f / LL_KS, ks = f % LL_KS;
  const int r = lane & 15, q = lane >> 4;
  pk[(size_t)g * 64 + lane] = *reinterpret_cast<const u32x4*>(W + (size_t)(32 * s + 8 * (r >> 2) + 4 * t2 + (r & 3)) * LL_D + 32 * ks + 8 * q);
}

struct LlArgs {
  const float* x;          // [M, 384] fp32 stream in
  const uint16_t* d1;      // [M, 384] bf16 pending residuals (ND of them)
  const uint16_t* d2;
  float* x_out;            // [M, 384] fp32 stream out (ND > 0), != x
  const float* g;          // LayerNorm weight, bias
  const float* b;
  const unsigned char* pk; // packed W
  const float* bias;       // [N]
  uint16_t* out;           // [M, N] bf16
  unsigned char* scratch;  // gridDim.x * 2 slots of 96 KiB
  int M, N;
  float eps;
};

// ND: pending residuals (0: x is normalised as it is and not written; 1: x + d1; 2: (x + d1) + d2, both written to x_out)
template <int ND>
__global__ __launch_bounds__(768, 3) void lnlin_kernel(const LlArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[LL_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x, bid = blockIdx.x;
  const int M = a.M, N = a.N, NE = N >> 6;
  const int nblk = (M + LL_ROWS - 1) / LL_ROWS;
  const long long U = (long long)nblk * NE;
  const int u0 = __builtin_amdgcn_readfirstlane((int)(U * bid / G)), u1 = __builtin_amdgcn_readfirstlane((int)(U * (bid + 1) / G));
  const int T = u1 - u0;
  if (T <= 0) return;
  const int b_first = __builtin_amdgcn_readfirstlane(u0 / NE), e0 = u0 - b_first * NE;      // (integer division runs on the vector ALU: once, here)
  const int b_last = __builtin_amdgcn_readfirstlane((u1 - 1) / NE);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned lane16 = (unsigned)lane * 16u;
  unsigned char* const myslots = a.scratch + (size_t)bid * (2 * LL_SLOT_BYTES);

  // ------------------------------------------------------------------------------------------------ LayerNorm of a row pair per wave
  // A block's 128 rows = 64 row pairs; a pair = one row per half-wave in layernorm_half_kernel<3>'s lane layout (lane `sub` of the half holds
  // the float4 chunks sub, sub + 32, sub + 64 of its row) with its arithmetic.  These loads are the COMPILER's (plain nontemporal loads,
  // its own vmcnt waits): an earlier version issued them as asm into persistent buffers and hipcc, not knowing the data was still in
  // flight, copied the buffers between register allocations right behind the loads (phi copies at a loop back edge, live-range splits
  // at 168 registers) -- garbage.  The waves that run this code in the main loop issue no other asm vector-memory instruction.
  const int sub = lane & 31, h2 = lane >> 5;
  struct LnBuf {
    f32x4 v[3];
    u32x2 p[3], q[3];
  };
  // pair pi (0..63) of block blk: rows 2 pi + h2, clamped to the last row
  auto ln_load = [&](int blk, int pi) __attribute__((always_inline)) {
    LnBuf in;
    const int row = min(blk * LL_ROWS + 2 * pi + h2, M - 1);
    const f32x4* xr = reinterpret_cast<const f32x4*>(a.x + (size_t)row * LL_D) + sub;
#pragma unroll
    for (int c = 0; c < 3; ++c) in.v[c] = __builtin_nontemporal_load(xr + 32 * c);
    if (ND >= 1) {
      const u32x2* dr = reinterpret_cast<const u32x2*>(a.d1 + (size_t)row * LL_D) + sub;
#pragma unroll
      for (int c = 0; c < 3; ++c) in.p[c] = __builtin_nontemporal_load(dr + 32 * c);
    }
    if (ND >= 2) {
      const u32x2* er = reinterpret_cast<const u32x2*>(a.d2 + (size_t)row * LL_D) + sub;
#pragma unroll
      for (int c = 0; c < 3; ++c) in.q[c] = __builtin_nontemporal_load(er + 32 * c);
    }
    return in;
  };
  // stream rows leave through a descriptor whose bound drops rows >= M -- and EVERY row for a workgroup that does not own the block's first
  // step (bound 0): the store instructions are issued either way, so the vmcnt bookkeeping does not depend on it
  auto x_rsrc = [&](bool owner) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.x_out), 0, (ND > 0 && owner) ? M * LL_D * 4 : 0, 0x00020000);
  };
  // the lane's LayerNorm parameters (chunks sub, sub + 32, sub + 64) are read from an LDS image per pair: as registers they are 24 too many
  // beside four load buffers, and as global loads inside ln_finish they would queue behind the row loads in flight
  const f32x4* const g_lds = reinterpret_cast<const f32x4*>(smem + LL_GB_OFF) + sub;
  const f32x4* const b_lds = g_lds + LL_D / 4;
  auto ln_finish = [&](const LnBuf& in, int blk, int pi, unsigned char* slot, const __amdgpu_buffer_rsrc_t xr) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(slot, 0, LL_SLOT_BYTES, 0x00020000);
    f32x4 v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = in.v[c];
    if (ND >= 2) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v[c][0] += __uint_as_float(in.p[c][0] << 16); v[c][1] += __uint_as_float(in.p[c][0] & 0xffff0000u);
        v[c][2] += __uint_as_float(in.p[c][1] << 16); v[c][3] += __uint_as_float(in.p[c][1] & 0xffff0000u);
      }
    }
    if (ND >= 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const u32x2 d = ND >= 2 ? in.q[c] : in.p[c];
        v[c][0] += __uint_as_float(d[0] << 16); v[c][1] += __uint_as_float(d[0] & 0xffff0000u);
        v[c][2] += __uint_as_float(d[1] << 16); v[c][3] += __uint_as_float(d[1] & 0xffff0000u);
      }
      const unsigned xo = ((unsigned)(blk * LL_ROWS + 2 * pi + h2) * LL_D + 4u * sub) * 4u;
#pragma unroll
      for (int c = 0; c < 3; ++c) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[c]), xr, xo + 512u * c, 0, 2 /* nt */);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)LL_D;
    float qs = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float t0 = v[c][0] - mean, t1 = v[c][1] - mean, t2 = v[c][2] - mean, t3 = v[c][3] - mean;
      v[c][0] = t0; v[c][1] = t1; v[c][2] = t2; v[c][3] = t3;
      qs += __builtin_fmaf(t0, t0, t1 * t1) + __builtin_fmaf(t2, t2, t3 * t3);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) qs += __shfl_xor(qs, o, 64);
    const float rstd = rsqrtf(qs / (float)LL_D + a.eps);
    const unsigned yo = ((unsigned)(2 * pi + h2) * LL_D + 4u * sub) * 2u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f32x4 gg = g_lds[32 * c], bb = b_lds[32 * c];
      u32x2 o;
      o[0] = pack_bf16x2(__builtin_fmaf(rstd * v[c][0], gg[0], bb[0]), __builtin_fmaf(rstd * v[c][1], gg[1], bb[1]));
      o[1] = pack_bf16x2(__builtin_fmaf(rstd * v[c][2], gg[2], bb[2]), __builtin_fmaf(rstd * v[c][3], gg[3], bb[3]));
      __builtin_amdgcn_raw_buffer_store_b64(o, yr, yo + 256u * c, 0, 0);
    }
  };
  // ---- the weight ring: time step t (unit u0 + t = block, entry e) consumes ring slot t % 3; the eight MFMA waves issue the 48 pieces of
  // an entry, six each (pieces wave + 8 k)
  int ld_e = e0, ld_slot = 0;                                            // entry of the next time step to load, its ring slot (MFMA waves)
  auto issue_piece = [&](int k) __attribute__((always_inline)) {
    const int f = wave + 8 * k;
    ll_piece(a.pk + (size_t)ld_e * LL_ENTRY + f * 1024, lane16, lds0 + (unsigned)ld_slot * LL_ENTRY + f * 1024);
  };
  auto advance_entry = [&]() __attribute__((always_inline)) {
    ld_e = (ld_e + 1 == NE) ? 0 : ld_e + 1;
    ld_slot = (ld_slot + 1 == LL_NSLOT) ? 0 : ld_slot + 1;
  };

  // ================================================================================================= prologue (all waves)
  LL_STAMP_DECL;
  LL_STAMP(0);
  // the bias vector goes to LDS once (behind the ring): an MFMA wave reads its eight values per step from there right before it needs
  // them -- as registers loaded a step ahead they were the eight registers too many (spills of asm-loaded registers are not survivable)
  float* const bias_lds = reinterpret_cast<float*>(smem + LL_NSLOT * LL_ENTRY);
  for (int i = tid; i < (N >> 2); i += 768) reinterpret_cast<f32x4*>(bias_lds)[i] = reinterpret_cast<const f32x4*>(a.bias)[i];
  if (tid < LL_D / 4) reinterpret_cast<f32x4*>(smem + LL_GB_OFF)[tid] = reinterpret_cast<const f32x4*>(a.g)[tid];
  else if (tid < LL_D / 2) reinterpret_cast<f32x4*>(smem + LL_GB_OFF)[tid] = reinterpret_cast<const f32x4*>(a.b)[tid - LL_D / 4];
  __syncthreads();                            // the LDS images (bias, LayerNorm parameters) are complete
  LL_STAMP(1);
  if (wave < 8) {                                                          // entries 0 and 1 fly while the first block is normalised
#pragma unroll
    for (int k = 0; k < 6; ++k) issue_piece(k);
    advance_entry();
    if (T > 1) {
#pragma unroll
      for (int k = 0; k < 6; ++k) issue_piece(k);
      advance_entry();
    }
  }
  {
    // the first block's rows by all twelve waves: pairs wave, wave + 12, .. of 64 (six rounds, two buffers) -- and, for a first block with
    // ONE step in this range, the second block too: its rows are needed at the end of that very step
    const int npro = (b_last > b_first && e0 + 1 == NE) ? 2 : 1;
    for (int pb = 0; pb < npro; ++pb) {
      const int blk = b_first + pb;
      unsigned char* slot = myslots + (size_t)pb * LL_SLOT_BYTES;
      const __amdgpu_buffer_rsrc_t xr = x_rsrc(pb > 0 || e0 == 0);
      LnBuf A = ln_load(blk, wave), B = ln_load(blk, wave + 12);
#pragma unroll
      for (int r = 0; r < 6; r += 2) {
        __builtin_amdgcn_sched_barrier(0);
        ln_finish(A, blk, wave + 12 * r, slot, xr);
        if (r + 2 < 6) A = ln_load(blk, wave + 12 * (r + 2));
        __builtin_amdgcn_sched_barrier(0);
        if (wave + 12 * (r + 1) < 64) ln_finish(B, blk, wave + 12 * (r + 1), slot, xr);
        if (r + 3 < 6) B = ln_load(blk, min(wave + 12 * (r + 3), 63));      // (waves 4..11, last round: pair 63 again, not used)
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  LL_STAMP(2);
  __builtin_amdgcn_s_barrier();               // entries 0 and 1 have landed, the first block's rows are in slot 0, the bias is in LDS
  asm volatile("" ::: "memory");
  LL_STAMP(3);
  if (wave == 0) LL_STAMP_DUMP(0, 63);
  if (wave == 8) LL_STAMP_DUMP(1, 63);

  if (wave >= 8) {
    // =============================================================================================== LN waves: the next block's rows
    // A block's 64 pairs: this wave takes pairs 4 i + lw, i = 0..15, four at a time.  LN step j = 0..4 of a block: buffer k = 0..3 is
    // finished as pair i = 4 (j - 1) + k (j >= 1) and then takes the loads of pair 4 j + k (j <= 3) -- four pairs (12 KB) of loads in flight
    // per wave, 48 KB per CU.  An LN step runs every SECOND time step, so the loads have two steps to arrive; whatever is left when the
    // refill is due goes back to back.
    const int lw = wave - 8;
    const __amdgpu_buffer_rsrc_t xr_own = x_rsrc(true);
    LnBuf Q0 = ln_load(b_first, lw), Q1 = Q0, Q2 = Q0, Q3 = Q0;      // (any defined value: never finished)
    int ln_blk = -1, ln_j = 5, phase = 0;
    int blk = b_first, e = e0;
    for (int t = 0; t < T; ++t) {
      const bool first_of_blk = (t == 0) || e == 0;
      const bool has_next = blk < b_last;
      // the step in which the MFMA waves refill their registers with the next block's rows is the block's last step (entry NE - 1; a block
      // with a successor in the range is in the range up to its end); the rows must be complete (and drained) at the barrier BEFORE that
      // step: at the end of this step if the next one is that step
      const bool deadline = has_next && e + 2 == NE;
      const bool one_step_blk = has_next && first_of_blk && e + 1 == NE;      // (only t == 0: the prologue did its successor)
      if (has_next && first_of_blk && !one_step_blk) { ln_blk = blk + 1; ln_j = 0; phase = 0; }
      LL_STAMP(0);
      int nsteps = (ln_blk == blk + 1 && ln_j < 5 && phase == 0) ? 1 : 0;
      if (deadline && ln_blk == blk + 1) nsteps = 5 - ln_j;
#ifdef LL_ABL_NO_LN
      nsteps = 0;                               // lab: the LN waves only keep the barriers company (outputs of later blocks are wrong)
#endif
      phase ^= 1;
      for (int k = 0; k < nsteps; ++k) {
        unsigned char* slot = myslots + (size_t)((ln_blk - b_first) & 1) * LL_SLOT_BYTES;
        const int j = ln_j;
#define LL_LN_SLOT(Q, k_)                                                                 \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    if (j >= 1) ln_finish(Q, ln_blk, 4 * (4 * (j - 1) + (k_)) + lw, slot, xr_own);         \
    if (j <= 3) Q = ln_load(ln_blk, 4 * (4 * j + (k_)) + lw);                              \
  } while (0)
        LL_LN_SLOT(Q0, 0);
        LL_LN_SLOT(Q1, 1);
        LL_LN_SLOT(Q2, 2);
        LL_LN_SLOT(Q3, 3);
#undef LL_LN_SLOT
        ++ln_j;
      }
      LL_STAMP(1);
      if (deadline) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the rows are in the slot (L2) before the barrier that releases the refill
      LL_STAMP(2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      LL_STAMP(3);
      if (wave == 8) LL_STAMP_DUMP(1, t);
      if (++e == NE) { e = 0; ++blk; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ================================================================================================= MFMA waves
  const int rs = wave & 3, fh = wave >> 2;
  const int frow = lane & 15, fq = lane >> 4;
  bf16x8 x0[LL_KS], x1[LL_KS];
  // a lane's B fragments of the block in a slot: rows 32 rs + frow (+ 16), columns 32 ks + 8 fq ..
  const unsigned xoff0 = ((unsigned)(32 * rs + frow) * LL_D + 8u * fq) * 2u, xoff1 = xoff0 + 16u * LL_D * 2u;
#define LL_LOAD_X(ks)                           \
  x0[ks] = ll_load_x<(ks) * 64>(myslots, xoff0); \
  x1[ks] = ll_load_x<(ks) * 64>(myslots, xoff1)
  LL_LOAD_X(0); LL_LOAD_X(1); LL_LOAD_X(2); LL_LOAD_X(3); LL_LOAD_X(4); LL_LOAD_X(5);
  LL_LOAD_X(6); LL_LOAD_X(7); LL_LOAD_X(8); LL_LOAD_X(9); LL_LOAD_X(10); LL_LOAD_X(11);
#undef LL_LOAD_X
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.out), 0, (int)((size_t)M * N * 2), 0x00020000);
  // bias of the lane's eight features of step s: bias[32 s + 8 fq + 0..3] (tile 0), + 4..7 (tile 1)
  const float* const bias_lane = bias_lds + 8 * fq;
  f32x4 bn0 = *reinterpret_cast<const f32x4*>(bias_lane + 32 * (e0 + NE * fh)), bn1 = *reinterpret_cast<const f32x4*>(bias_lane + 32 * (e0 + NE * fh) + 4);
  asm volatile("s_waitcnt vmcnt(0)" : LL_TIE_X(x0), LL_TIE_X(x1)::"memory");

  bf16x8 w[2][2];                               // fragments of the k-step being multiplied and of the next one
  const unsigned char* const wbase = smem + fh * (LL_FRH * 1024) + lane16;      // this half's fragments of a slot
#define LL_READW(buf, slotp, ks)                                                      \
  buf[0] = *reinterpret_cast<const bf16x8*>((slotp) + (ks) * 1024);                   \
  buf[1] = *reinterpret_cast<const bf16x8*>((slotp) + (LL_KS + (ks)) * 1024)
  LL_READW(w[0], wbase, 0);
  int cslot = 0;
  int blk = b_first, e = e0;
  for (int t = 0; t < T; ++t) {
    const bool more = t + 1 < T;
    const int e_n = (e + 1 == NE) ? 0 : e + 1, blk_n = (e + 1 == NE) ? blk + 1 : blk;
    int flags = __builtin_amdgcn_readfirstlane(((more && blk_n != blk) ? 1 : 0) | ((t + 2 < T) ? 2 : 0));
    asm volatile("" : "+s"(flags));             // opaque: one copy of the step body
    const bool reload = (flags & 1) != 0;       // the block's last step: the x registers are refilled IN PLACE behind their last MFMA
    const bool dma = (flags & 2) != 0;          // entry t + 2 exists: its pieces go out behind the first windows' MFMAs
    const unsigned char* nslotx = myslots + (size_t)((blk_n - b_first) & 1) * LL_SLOT_BYTES;
    const unsigned char* slot = wbase + cslot * LL_ENTRY;
    const int nslot = (cslot + 1 == LL_NSLOT) ? 0 : cslot + 1;
    const unsigned char* slot_next = wbase + nslot * LL_ENTRY;
    f32x4 a00 = bn0, a01 = bn0, a10 = bn1, a11 = bn1;
    LL_STAMP(0);
#ifdef LL_ABL_NO_DMA
#define LL_PIECE(k) asm volatile("global_load_dword %0, %1, %2" : "=v"(dummy_) : "v"(0u), "s"(a.pk) : "memory")      /* lab: a 4-byte load in place of the 1-KiB piece (same vmcnt) */
    unsigned dummy_;
#else
#define LL_PIECE(k) issue_piece(k)
#endif
#define LL_WIN(ks)                                                                                   \
  do {                                                                                               \
    if ((ks) + 1 < LL_KS) { LL_READW(w[((ks) + 1) & 1], slot, (ks) + 1); }                           \
    else { LL_READW(w[((ks) + 1) & 1], slot_next, 0); }                                              \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    a00 = LL_MFMA(w[(ks) & 1][0], x0[ks], a00);                                                      \
    a01 = LL_MFMA(w[(ks) & 1][0], x1[ks], a01);                                                      \
    a10 = LL_MFMA(w[(ks) & 1][1], x0[ks], a10);                                                      \
    a11 = LL_MFMA(w[(ks) & 1][1], x1[ks], a11);                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    if ((ks) < 6 && dma) LL_PIECE(ks);                                                               \
    if (reload) { ll_reload_x<(ks) * 64>(x0[ks], nslotx, xoff0); ll_reload_x<(ks) * 64>(x1[ks], nslotx, xoff1); } \
  } while (0)
    LL_WIN(0); LL_WIN(1); LL_WIN(2); LL_WIN(3); LL_WIN(4); LL_WIN(5);
    if (dma) advance_entry();
    LL_WIN(6); LL_WIN(7); LL_WIN(8); LL_WIN(9); LL_WIN(10);
    // Entry t + 1 (its pieces went out a step ago) has landed once only what is younger remains: the previous step's two stores, this
    // step's six pieces -- and, in a refill step, the 22 refill loads so far.  Every fragment of entry t is in registers (requested up to
    // k-step 11): the barrier releases its ring slot and publishes entry t + 1, whose first fragments the last window requests.
    // (no pieces went out in this step: the same without them)
    LL_STAMP(1);
    if (dma) {
      if (reload) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      if (reload) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    LL_STAMP(2);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    LL_STAMP(3);
    if (wave == 0) LL_STAMP_DUMP(0, t);
    LL_WIN(11);
#undef LL_WIN
    // the next step's start values (LDS: read here, needed behind the stores below)
    bn0 = *reinterpret_cast<const f32x4*>(bias_lane + 32 * (e_n + NE * fh));
    bn1 = *reinterpret_cast<const f32x4*>(bias_lane + 32 * (e_n + NE * fh) + 4);
    // finished outputs: tile 0 holds the lane's features 8 fq + 0..3 of the step, tile 1 + 4..7
    {
      const int s = e + NE * fh;
      const unsigned vo = ((unsigned)(blk * LL_ROWS + 32 * rs + frow) * (unsigned)N + 32u * (unsigned)s + 8u * fq) * 2u;
      const u32x4 o0 = {pack_bf16x2(a00[0], a00[1]), pack_bf16x2(a00[2], a00[3]), pack_bf16x2(a10[0], a10[1]), pack_bf16x2(a10[2], a10[3])};
      const u32x4 o1 = {pack_bf16x2(a01[0], a01[1]), pack_bf16x2(a01[2], a01[3]), pack_bf16x2(a11[0], a11[1]), pack_bf16x2(a11[2], a11[3])};
#ifdef LL_ABL_NO_STORE
      asm volatile("" ::"v"(o0), "v"(o1), "v"(vo));      // lab: no output stores (the vmcnt counts below then over-wait: ablation timing only)
      __builtin_amdgcn_raw_buffer_store_b128(o0, __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.out), 0, 0, 0x00020000), vo, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(o1, __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.out), 0, 0, 0x00020000), vo, 0, 0);
#else
      __builtin_amdgcn_raw_buffer_store_b128(o0, orsrc, vo, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(o1, orsrc, vo + (unsigned)(16 * N * 2), 0, 0);
#endif
    }
    // a refill's loads have landed before the next step uses them (the two stores, youngest, stay in flight); otherwise nothing is waited
    // for here: ONE tied statement on every path (a tied statement per branch makes hipcc copy the 96 x registers between allocations)
    if (reload) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(32)" : LL_TIE_X(x0), LL_TIE_X(x1)::"memory");
    cslot = nslot;
    e = e_n;
    blk = blk_n;
  }
#undef LL_READW
}

}  // namespace

extern "C" int tr_lnlin_supported(int D, int N) { return (D == LL_D && N >= 128 && N % 64 == 0 && N <= LL_BIAS_MAX) ? 1 : 0; }      // (N / 64 >= 2 steps per block)

static int ll_grid() {
  static const int forced = [] { const char* e = getenv("TR_LNLIN_GRID"); return e ? atoi(e) : 0; }();
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

extern "C" size_t tr_lnlin_pack_bytes(int D, int N) { return tr_lnlin_supported(D, N) ? (size_t)N * D * 2 : 0; }
extern "C" size_t tr_lnlin_scratch_bytes(int D, int N) { return tr_lnlin_supported(D, N) ? (size_t)ll_grid() * 2 * LL_SLOT_BYTES + 65536 : 0; }      // (+ 64 KiB: the diagnostic build's stamps)

extern "C" int tr_lnlin_pack_bf16(const uint16_t* W, void* packed, int D, int N, tr_stream_t s) {
  TR_REQUIRE(W && packed, TR_ERR_NULL, "tr_lnlin_pack_bf16: null pointer");
  TR_REQUIRE(tr_lnlin_supported(D, N), TR_ERR_SHAPE, "tr_lnlin_pack_bf16: unsupported shape D=%d N=%d (D must be %d, N %% 64 == 0)", D, N, LL_D);
  TR_REQUIRE(tr_aligned16(W) && tr_aligned16(packed), TR_ERR_ALIGN, "tr_lnlin_pack_bf16: pointers must be 16-byte aligned");
  const int frags = (N / 64) * 2 * LL_FRH;
  hipLaunchKernelGGL(lnlin_pack_kernel, dim3((frags + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(s), W, static_cast<u32x4*>(packed), N);
  TR_CHECK_LAUNCH("tr_lnlin_pack_bf16");
  return TR_OK;
}

extern "C" int tr_lnlin_bf16(const float* x, const uint16_t* d1, const uint16_t* d2, float* x_out, const float* g, const float* b, float eps,
                             const void* packed, const float* bias, uint16_t* out, void* scratch, size_t scratch_bytes, int M, int D, int N,
                             tr_stream_t s) {
  TR_REQUIRE(x && g && b && packed && bias && out && scratch, TR_ERR_NULL, "tr_lnlin_bf16: null pointer");
  TR_REQUIRE(M > 0 && tr_lnlin_supported(D, N), TR_ERR_SHAPE, "tr_lnlin_bf16: unsupported shape M=%d D=%d N=%d (D must be %d, N %% 64 == 0)", M, D,
             N, LL_D);
  TR_REQUIRE(d1 != nullptr || d2 == nullptr, TR_ERR_NULL, "tr_lnlin_bf16: d2 without d1");
  TR_REQUIRE((d1 == nullptr) == (x_out == nullptr), TR_ERR_NULL,
             "tr_lnlin_bf16: x_out goes with the pending residuals (both or neither: without one the stream is not rewritten)");
  TR_REQUIRE(x_out != x, TR_ERR_CONFIG, "tr_lnlin_bf16: the stream is written out of place (a block's rows may be read by two workgroups)");
  TR_REQUIRE(tr_aligned16(x) && tr_aligned16(d1) && tr_aligned16(d2) && tr_aligned16(x_out) && tr_aligned16(g) && tr_aligned16(b) &&
                 tr_aligned16(packed) && tr_aligned16(bias) && tr_aligned16(out) && tr_aligned16(scratch),
             TR_ERR_ALIGN, "tr_lnlin_bf16: pointers must be 16-byte aligned");
  TR_REQUIRE(scratch_bytes >= tr_lnlin_scratch_bytes(D, N), TR_ERR_SHAPE, "tr_lnlin_bf16: scratch of %zu bytes, tr_lnlin_scratch_bytes says %zu",
             scratch_bytes, tr_lnlin_scratch_bytes(D, N));
  TR_REQUIRE((size_t)M * N * 2 < ((size_t)1 << 31) && (size_t)M * D * 4 < ((size_t)1 << 31), TR_ERR_SHAPE,
             "tr_lnlin_bf16: M=%d exceeds the range of the 32-bit offsets", M);
  TR_REQUIRE(N <= LL_BIAS_MAX, TR_ERR_SHAPE, "tr_lnlin_bf16: N=%d output features, at most %d (the bias vector's LDS image)", N, LL_BIAS_MAX);
  const int nd = d2 ? 2 : (d1 ? 1 : 0);
  // the LayerNorm launch it replaces moves (4 [+ 4 written]) + 2 nd + 2 bytes per element; the GEMM 2 (M K + N K + M N) -- here the
  // normalised rows stay on the chip: x in, residuals in, x out, W, out
  tr_prof_note("lnlin_kernel", 2.0 * M * N * D, (double)M * D * (4.0 + 2.0 * nd + (nd ? 4.0 : 0.0)) + 2.0 * N * D + 2.0 * M * N);
  const int nblk = (M + LL_ROWS - 1) / LL_ROWS;
  const long long U = (long long)nblk * (N / 64);
  const int grid = ll_grid();
  const int G = U < grid ? (int)U : grid;
  const LlArgs a{x, d1, d2, x_out, g, b, static_cast<const unsigned char*>(packed), bias, out, static_cast<unsigned char*>(scratch), M, N, eps};
  hipStream_t st = static_cast<hipStream_t>(s);
  if (nd == 2) hipLaunchKernelGGL(lnlin_kernel<2>, dim3(G), dim3(768), 0, st, a);
  else if (nd == 1) hipLaunchKernelGGL(lnlin_kernel<1>, dim3(G), dim3(768), 0, st, a);
  else hipLaunchKernelGGL(lnlin_kernel<0>, dim3(G), dim3(768), 0, st, a);
  TR_CHECK_LAUNCH("tr_lnlin_bf16");
  return TR_OK;
}
