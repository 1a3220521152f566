// bf16 Linear layers with a RESIDENT activation block, for K = 320 / 384 (the qkv, proj and fc1 GEMMs of DeiT-S: topk.py:44,52 and timm Mlp).
//
// Why (profiles/r04_gemm_lab.md): gemm_bf16_pc is bound by the CU's L2 -> LDS feed, ~29 cycles per 1-KiB LDS-DMA piece; a 256x128 tile
// feeds 48 pieces per K-step for 1024 matrix cycles and the matrix pipes idle a third of every step.  With K = 384 a block of 128 token
// rows is 96 KiB: it stays in LDS while the workgroup walks the column tiles of those rows, and only WEIGHT slabs stream (16 pieces per
// K-step for 512 matrix cycles): 8.7 instead of 14 vector-memory instructions per MFLOP, and the K-loop becomes matrix-bound.
//   * 8 waves: 4 MFMA waves (one per SIMD, 64x64 outputs each of a 128x128 tile, 256 VGPRs) + 4 service waves (all LDS-DMA, all stores).
//   * LDS, all 160 KiB: activation block 6 x 16 KiB (one slab per K-step), weight ring 3 x 16 KiB, two 8-KiB mailboxes.
//   * Two accumulator sets: while a tile accumulates in one, the tile before leaves from the other -- one 16-row slab per K-step,
//     rounded to bf16 into a mailbox (the staging layout of gemm_bf16_pc: conflict-free both ways).  The per-step barrier that
//     publishes the next weight slab also publishes the mailbox: the service waves read it in the next round, apply the GELU where
//     asked and store whole 128-byte lines.  No flag, no polling; the MFMA waves spend 12 instructions per step on the hand-over.
//   * Work is cut into (row block, column tile) units, rows outermost, and dealt to the 256 workgroups as equal contiguous runs: every
//     workgroup gets the same number of units to within one (no tail round), a run that starts inside a row block loads that block
//     itself.  When a run moves to the next row block the new slabs replace the old ones one K-step behind the last tile's reads.
//   * vmcnt retires in order: a service wave counts every vector-memory instruction it issues and remembers the count behind the
//     weight slab of each step; the wait in front of a barrier is the exact difference.
// Numerics: accumulation order = gemm_bf16_pc's (bias, then k ascending in steps of 32): TR_EPI_BF16 is bit-identical to it.
// TR_EPI_GELU_BF16 applies the GELU fit to the bf16-ROUNDED pre-activation -- the definition of the training forward
// (tr_gemm_gelu_keep_bf16) and of tr_gelu_bf16 -- so eval and training agree bit for bit on fc1.
#include "../../tokenreduction_amd/csrc/tr_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int AR_BM = 128, AR_BN = 128, AR_BK = 64;
constexpr int AR_SLAB = 128 * 128;          // 16 KiB: 128 rows x 64 bf16
constexpr int AR_MAXK = 6;                  // K-steps a resident block can hold
constexpr int AR_WRING = AR_MAXK * AR_SLAB; // byte offset of the weight ring (3 slabs)
constexpr int AR_MBOX = AR_WRING + 3 * AR_SLAB;   // two mailboxes of 8 KiB: [MFMA wave][16 rows][128 B]
constexpr int AR_LDS = AR_MBOX + 2 * 8192;  // 163840: the whole LDS

__device__ __forceinline__ int ar_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void ar_piece(const void* sbase, unsigned voff, unsigned lds_dst) {
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

// s_waitcnt vmcnt(N) with the largest N <= n out of the few values a round can leave behind (the immediate must be a constant; waiting
// for MORE than necessary is always correct).  In a steady round n is 4 (the next weight slab), +2 with a served chunk, +4 with a replaced
// activation slab: the common values come first -- a 33-way switch compiled to a tree of ~30 scalar branches, a few hundred cycles per
// round of a wave whose whole round should take 500.
__device__ __forceinline__ void ar_wait_vm(int n) {
  if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifdef TR_AR_STAMPS
__device__ unsigned long long ar_stamps[2][128][4];
#define AR_STAMP(who, step, k)                                                                                     \
  do {                                                                                                             \
    if (blockIdx.x == 8 && (step) < 128 && (lane) == 0) ar_stamps[who][step][k] = __builtin_amdgcn_s_memtime();    \
  } while (0)
#else
#define AR_STAMP(who, step, k) do { } while (0)
#endif

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_ar(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                       const float* __restrict__ bias, uint16_t* __restrict__ outp, int M, int N, int K, int nRb,
                                                       int nNt, unsigned out_bytes) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[AR_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = K / AR_BK;                       // 5 or 6 (launcher)
  const long U = (long)nRb * nNt;                 // units, row block outermost
  const int u0 = (int)(U * blockIdx.x / gridDim.x), u1 = (int)(U * (blockIdx.x + 1) / gridDim.x);
  const int nu = u1 - u0;
  if (nu <= 0) return;
  const int S = nu * nk;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  if (wave >= 4) {
    // ============================================================ service wave
    const int lw = wave - 4;
    const int l3 = lane >> 3, pc = lane & 7;
    // a slab is 128 rows = 16 pieces of 8 rows; this wave moves rows lw*32 + 8p + l3 (p = 0..3) of every slab, activation or weight.
    // Per lane and piece: the byte offset of (its row, its swizzled 16-byte chunk) inside the operand, for the CURRENT row block /
    // column tile; a K-step adds 128 bytes.  Recomputed only when the row block / column tile changes -- a round of this wave is
    // 4-8 DMA instructions and must not cost more than the 512 matrix cycles it runs beside (first version: integer divisions and
    // address products per piece, ~600 cycles of bookkeeping per round: slower than the kernel it replaces).
    unsigned wrow[4], arow[4];
    const unsigned Kb = (unsigned)K * 2u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = lw * 32 + 8 * p + l3;
      wrow[p] = (unsigned)r * Kb + (unsigned)(pc ^ ((r >> 1) & 7)) * 16u;       // + column tile * 128 rows * Kb
    }
    auto set_arow = [&](int rb) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int r = lw * 32 + 8 * p + l3;
        arow[p] = (unsigned)min(rb * AR_BM + r, M - 1) * Kb + (unsigned)(pc ^ ((r >> 1) & 7)) * 16u;
      }
    };
    const unsigned dst_rows = lds0 + lw * (32 * 128);
    int issued = 0;                // vector-memory instructions of this wave so far (pieces + stores)
    auto a_slab = [&](int kt) __attribute__((always_inline)) {       // activation slab kt of the row block `arow` points at
      const unsigned koff = (unsigned)kt * 128u;
#pragma unroll
      for (int p = 0; p < 4; ++p) ar_piece(A, arow[p] + koff, dst_rows + kt * AR_SLAB + p * 1024);
      issued += 4;
    };
    auto w_slab = [&](unsigned ctoff, int kt, int slot) __attribute__((always_inline)) {   // weight slab (column tile at byte ctoff, K-step kt)
      const unsigned off = ctoff + (unsigned)kt * 128u;
#pragma unroll
      for (int p = 0; p < 4; ++p) ar_piece(W, wrow[p] + off, dst_rows + AR_WRING + slot * AR_SLAB + p * 1024);
      issued += 4;
    };
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)out_bytes, 0x00020000);
    // mailbox chunk s of the unit at (row block rb, column tile ct): 16 rows x 64 columns of MFMA wave lw -> global memory (two
    // full-line stores per lane)
    const int rrow = lane >> 3, rch = lane & 7;
    const unsigned char* const rd0 = smem + AR_MBOX + lw * 2048 + rrow * 128 + ((rch ^ rrow) << 4);
    const unsigned lane_out = ((unsigned)((lw >> 1) * 64 + rrow) * (unsigned)N + (unsigned)((lw & 1) * 64 + rch * 8)) * 2u;
    auto serve = [&](int rb, int ct, int s) __attribute__((always_inline)) {
      const unsigned char* rd = rd0 + (s & 1) * 8192;
      u32x4 ln[2];
      ln[0] = *reinterpret_cast<const u32x4*>(rd);
      const u32x4 t = *reinterpret_cast<const u32x4*>(rd + 1024);       // rows 8..15: halves swapped
      ln[1] = u32x4{t[2], t[3], t[0], t[1]};
      const int m = rb * AR_BM + (lw >> 1) * 64 + s * 16 + rrow;
      const unsigned off0 = lane_out + ((unsigned)(rb * AR_BM + s * 16) * (unsigned)N + (unsigned)(ct * AR_BN)) * 2u;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        u32x4 v = ln[r];
        if (EPI == TR_EPI_GELU_BF16) {
#ifndef TR_ABLATE_NO_GELU
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2 x = {__builtin_bit_cast(float, v[e] << 16), __builtin_bit_cast(float, v[e] & 0xffff0000u)};
            const f32x2 g = gelu2(x);
            v[e] = pack_bf16x2(g[0], g[1]);
          }
#endif
        }
        bool ok = m + 8 * r < M;
#ifdef TR_ABLATE_NO_STORE
        ok = ok && (K == 0x7fffffff);
#endif
        // out-of-range rows get an offset beyond num_records: the buffer bounds check drops the store, the count stays exact
        __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ok ? off0 + (unsigned)r * 8u * (unsigned)N * 2u : 0x80000000u, 0, 0);
      }
      issued += 2;
    };

    // ---- prologue: the first row block (slab 0 first) and the first two weight slabs
    const int rb0 = u0 / nNt, ct0 = u0 - rb0 * nNt;
    const unsigned ct_step = (unsigned)AR_BN * Kb;       // bytes between column tiles of W
    int mark0, mark1;              // `issued` right behind the weight slab of steps b and b+1 (that of b+2 is issued in round b)
    set_arow(rb0);
    a_slab(0);
    w_slab((unsigned)ct0 * ct_step, 0, 0);
    mark0 = issued;
    a_slab(1);
    w_slab((unsigned)ct0 * ct_step, 1, 1);  // step 1 is (u0, K-step 1): nk >= 5
    mark1 = issued;
    for (int kt = 2; kt < nk; ++kt) a_slab(kt);
    // the step whose weight slab the next round issues (b + 2): unit index, column tile (as a byte offset into W), K-step, ring slot
    int w_u = u0, w_ct = ct0, w_kt = 2, w_slot = 2;
    unsigned w_ctoff = (unsigned)ct0 * ct_step;
    // step b - 1 (the step the MFMA waves have just finished reading when round b starts): unit, its row block / column tile, K-step
    int p_u = u0, p_rb = rb0, p_ct = ct0, p_kt = -1;
    // the unit BEFORE p_u (whose slabs are handed over during p_u's first four K-steps)
    int q_rb = rb0, q_ct = ct0;
    bool arow_next = false;        // arow already points at the row block after p_rb
    for (int b = 0; b < S; ++b) {
      ar_wait_vm(issued - mark0);
      if (lw == 0) AR_STAMP(1, b, 0);
      __builtin_amdgcn_s_barrier();          // B_b: step b's slabs have landed; the MFMA waves are done reading step b-1
      asm volatile("" ::: "memory");
      if (lw == 0) AR_STAMP(1, b, 1);
#ifndef TR_ABLATE_NO_EPI
      // the chunk the MFMA waves handed over during step b-1 (K-steps 0..3 of every unit but the first): slab p_kt of the unit before
      if (p_kt >= 0 && p_kt < 4 && p_u > u0) serve(q_rb, q_ct, p_kt);
#endif
      // step b-1 was (p_u, p_kt): if the run leaves that row block after this unit, its slab p_kt is free for the next block's
      if (p_kt >= 0 && p_ct == nNt - 1 && p_u + 1 < u1) {
        if (!arow_next) { set_arow(p_rb + 1); arow_next = true; }
        a_slab(p_kt);
      }
      if (w_u < u1) w_slab(w_ctoff, w_kt, w_slot);
      mark0 = mark1;
      mark1 = issued;                // behind the weight slab of step b+2 (or behind nothing more, at the tail)
      if (lw == 0) AR_STAMP(1, b, 2);
      w_slot = (w_slot == 2) ? 0 : w_slot + 1;
      if (++w_kt == nk) {
        w_kt = 0;
        ++w_u;
        if (++w_ct == nNt) { w_ct = 0; w_ctoff = 0; } else { w_ctoff += ct_step; }
      }
      if (p_kt < 0) {
        p_kt = 0;
      } else if (++p_kt == nk) {
        p_kt = 0;
        ++p_u;
        q_rb = p_rb; q_ct = p_ct;
        if (++p_ct == nNt) { p_ct = 0; ++p_rb; arow_next = false; }
      }
    }
#ifndef TR_ABLATE_NO_EPI
    // drain: the last unit's four slabs, one per barrier (after the loop p_* is the run's last step, i.e. its last unit)
    for (int d = 0; d < 4; ++d) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      serve(p_rb, p_ct, d);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ================================================================ MFMA wave: 64 x 64 of the 128 x 128 tile
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  bf16x8 wA[4], aA[4], wC[4], aC[4];
#define AR_READ(WF, AF, wslot, kt_, ks)                                                                                  \
  do {                                                                                                                   \
    const unsigned char* sa_ = smem + (kt_) * AR_SLAB;                                                                   \
    const unsigned char* sw_ = smem + AR_WRING + (wslot) * AR_SLAB;                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                                   \
      WF[i_] = *reinterpret_cast<const bf16x8*>(sw_ + ar_swz(wn * 64 + i_ * 16 + frow, 4 * (ks) + fq));                  \
      AF[i_] = *reinterpret_cast<const bf16x8*>(sa_ + ar_swz(wm * 64 + i_ * 16 + frow, 4 * (ks) + fq));                  \
    }                                                                                                                    \
  } while (0)
#ifdef TR_ABLATE_NO_MFMA
#define AR_GROUP(ACC, WF, AF, i_) asm volatile("" ::"v"(WF[i_]), "v"(AF[i_]))
#else
#define AR_GROUP(ACC, WF, AF, i_)                                                                                        \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                       \
      ACC[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[i_], AF[j_], ACC[i_][j_], 0, 0, 0)
#endif

  f32x4 accP[4][4], accQ[4][4];
  // a unit's accumulators START at the bias of its columns (this wave's 64), loaded straight into the set that is free: the set a unit
  // leaves from is empty after its fourth slab, so the bias of the unit after next... of the NEXT unit goes there during K-step 4
  auto load_bias = [&](f32x4 (&acc)[4][4], int u) __attribute__((always_inline)) {
    const int rb = u / nNt, ct = u - rb * nNt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias + ct * AR_BN + wn * 64 + i * 16 + 4 * fq);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = b;
    }
  };
  // slab j of a finished tile -> this wave's 2 KiB of mailbox (j & 1), bf16; the staging layout of gemm_bf16_pc: 16-byte chunk XOR row&7,
  // rows 8..15 take the other 8-byte half
  auto stage = [&](f32x4 (&old)[4][4], const int j) __attribute__((always_inline)) {
#ifndef TR_ABLATE_NO_EPI
    unsigned char* stg = smem + AR_MBOX + (j & 1) * 8192 + wave * 2048;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x2 pk;
      pk[0] = pack_bf16x2(old[i][j][0], old[i][j][1]);
      pk[1] = pack_bf16x2(old[i][j][2], old[i][j][3]);
      *reinterpret_cast<u32x2*>(stg + frow * 128 + (((2 * i + (fq >> 1)) ^ (frow & 7)) << 4) + (((fq & 1) ^ (frow >> 3)) << 3)) = pk;
    }
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(old[i][j]));
#endif
  };

  int gs = 0, wslot = 0;
  // one unit (nk K-steps) accumulating into `cur` while the unit before leaves from `old` (HAVE_OLD: not the run's first unit)
  auto run_unit = [&](f32x4 (&cur)[4][4], f32x4 (&old)[4][4], const bool have_old, int u) __attribute__((always_inline)) {
    for (int kt = 0; kt < nk; ++kt) {
      if (wave == 0) AR_STAMP(0, gs, 0);
      const int nslot = (wslot == 2) ? 0 : wslot + 1;
      const int nkt = (kt + 1 == nk) ? 0 : kt + 1;
      AR_READ(wC, aC, wslot, kt, 1);
      AR_GROUP(cur, wA, aA, 0);
      AR_GROUP(cur, wA, aA, 1);
      AR_GROUP(cur, wA, aA, 2);
      AR_GROUP(cur, wA, aA, 3);
      if (have_old && kt < 4) {
        switch (kt) {                                  // static slab index: the accumulators are registers
          case 0: stage(old, 0); break;
          case 1: stage(old, 1); break;
          case 2: stage(old, 2); break;
          default: stage(old, 3); break;
        }
      }
      if (kt == 4 && u + 1 < u1) load_bias(old, u + 1);       // `old` is empty by now; the next unit accumulates there
      if (wave == 0) AR_STAMP(0, gs, 1);
      if (gs + 1 < S) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own reads of step gs done, own mailbox writes visible after the barrier
        __builtin_amdgcn_s_barrier();                        // B_{gs+1}
        asm volatile("" ::: "memory");
      }
      if (wave == 0) AR_STAMP(0, gs, 2);
      AR_READ(wA, aA, nslot, nkt, 0);                        // after the last step: stale slabs, unused
      __builtin_amdgcn_sched_barrier(0);
      AR_GROUP(cur, wC, aC, 0);
      AR_GROUP(cur, wC, aC, 1);
      AR_GROUP(cur, wC, aC, 2);
      AR_GROUP(cur, wC, aC, 3);
      wslot = nslot;
      ++gs;
    }
  };

  load_bias(accP, u0);
  __builtin_amdgcn_s_barrier();            // B_0
  asm volatile("" ::: "memory");
  AR_READ(wA, aA, 0, 0, 0);
  int u = u0;
  run_unit(accP, accQ, false, u);
  ++u;
  while (u < u1) {
    run_unit(accQ, accP, true, u);
    ++u;
    if (u >= u1) break;
    run_unit(accP, accQ, true, u);
    ++u;
  }
  // drain: the last unit's accumulators are in accP when the run has an odd number of units, else in accQ
  if ((nu & 1) != 0) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      stage(accP, d);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef TR_ABLATE_NO_EPI
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
    }
  } else {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      stage(accQ, d);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef TR_ABLATE_NO_EPI
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
    }
  }
#undef AR_READ
#undef AR_GROUP
}

}  // namespace

// Is (M, N, K) a shape the resident-activation kernel takes?  K = 320 or 384 (a 128-row block of 5-6 K-steps fills the 96 KiB it has,
// and the hand-over of a tile needs four K-steps of the next one), N a multiple of 128.
extern "C" int tr_gemm_ar_supported(int M, int N, int K) {
  return M > 0 && N > 0 && N % AR_BN == 0 && K % AR_BK == 0 && K / AR_BK >= 5 && K / AR_BK <= AR_MAXK;
}

// tr_gemm_bf16 for TR_EPI_BF16 / TR_EPI_GELU_BF16 on those shapes; exported so that tests and the lab can call it directly.
extern "C" int tr_gemm_bf16_ar(const uint16_t* A, const uint16_t* W, const float* bias, uint16_t* out, int M, int N, int K, int epilogue,
                               tr_stream_t s) {
  TR_REQUIRE(A && W && bias && out, TR_ERR_NULL, "tr_gemm_bf16_ar: null pointer");
  TR_REQUIRE(epilogue == TR_EPI_BF16 || epilogue == TR_EPI_GELU_BF16, TR_ERR_SHAPE, "tr_gemm_bf16_ar: bf16 epilogues only (got %d)", epilogue);
  TR_REQUIRE(tr_gemm_ar_supported(M, N, K), TR_ERR_SHAPE, "tr_gemm_bf16_ar: need K = 320 or 384 and N %% 128 == 0 (M=%d N=%d K=%d)", M, N, K);
  TR_REQUIRE(tr_aligned16(A) && tr_aligned16(W) && tr_aligned16(bias) && (reinterpret_cast<uintptr_t>(out) & 127u) == 0, TR_ERR_ALIGN,
             "tr_gemm_bf16_ar: operands must be 16-byte aligned, the output 128-byte aligned");
  const size_t out_bytes = (size_t)M * N * 2;
  TR_REQUIRE(out_bytes < ((size_t)1 << 31) && (size_t)M * K * 2 < ((size_t)1 << 32) && (size_t)N * K * 2 < ((size_t)1 << 32), TR_ERR_SHAPE,
             "tr_gemm_bf16_ar: operands / outputs beyond the 32-bit offset range");
  tr_prof_note(epilogue == TR_EPI_BF16 ? "gemm_bf16_ar<EPI_BF16>" : "gemm_bf16_ar<EPI_GELU_BF16>", 2.0 * M * N * K,
               2.0 * ((double)M * K + (double)N * K) + 2.0 * M * N);
  const int nRb = (M + AR_BM - 1) / AR_BM, nNt = N / AR_BN;
  hipStream_t st = static_cast<hipStream_t>(s);
  if (epilogue == TR_EPI_BF16)
    hipLaunchKernelGGL(gemm_bf16_ar<TR_EPI_BF16>, dim3(256), dim3(512), 0, st, A, W, bias, out, M, N, K, nRb, nNt, (unsigned)out_bytes);
  else
    hipLaunchKernelGGL(gemm_bf16_ar<TR_EPI_GELU_BF16>, dim3(256), dim3(512), 0, st, A, W, bias, out, M, N, K, nRb, nNt, (unsigned)out_bytes);
  TR_CHECK_LAUNCH("tr_gemm_bf16_ar");
  return TR_OK;
}
