// bf16 Linear layers on MFMA for gfx950:  out = epilogue(A[M,K] * W[N,K]^T + bias[N]).
//
// Replaces the nn.Linear calls of the reference's block (SURVEY.md 8a rows a1,a3,a4,a5):
//   qkv      topk.py:44     (TR_EPI_BF16)          proj     topk.py:52 + residual :87 (TR_EPI_RESID_F32)
//   mlp.fc1  timm Mlp + GELU (TR_EPI_GELU_BF16)    mlp.fc2  + residual topk.py:95     (TR_EPI_RESID_F32)
//   head     topk.py:203    (TR_EPI_F32)           patch_embed.proj topk.py:181-186   (TR_EPI_PATCH_F32)
//
// Structure (cdna_hip_programming.md section 5, "Pipelining across barriers", T1-T4):
//   * ONE persistent 512-thread workgroup per CU walks a list of 256x128 output tiles; 8 waves as 4(M) x 2(N), each
//     wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16.  W is the MFMA "A" operand and the activation the "B" operand, so a
//     lane's accumulator holds 4 CONSECUTIVE output columns -> 8/16-byte epilogue stores.
//   * Operands go global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, no VGPR staging) into a 3-slot ring of 48-KiB
//     K-steps (BK = 64; 144 KiB of the 160 KiB LDS).  Two K-steps (96 KiB) are always in flight behind a COUNTED
//     `s_waitcnt vmcnt(N)`, one raw `s_barrier` per K-step, and the ring runs ACROSS tile boundaries (K is only 384
//     for three of the four block GEMMs = 6 K-steps per tile, so a per-tile pipeline fill would dominate).
//   * LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled by (row>>1)&7 -- conflict-free ds_read_b128
//     fragment reads (tools/lds_sim.py).  LDS-DMA writes lane-linearly, so the swizzle is applied to the per-lane
//     SOURCE address and again on the read (guide rule 21).
//   * vmcnt is one in-order counter for loads, stores and LDS-DMA on gfx950.  Everything that touches it in the K-loop
//     is therefore hand-counted: the DMA and the epilogue's bias/residual loads are inline asm (invisible to hipcc's
//     own waitcnt insertion, which would otherwise drain the ring with vmcnt(0) once per tile), the loads are issued
//     BEFORE the next K-step's DMA so waiting for them leaves that DMA in flight, and the 16 epilogue stores per lane
//     are branch-free buffer stores (out-of-range lanes are dropped by the descriptor's bounds check), so the number
//     of stores younger than a DMA group is exact and can be added to the allowed count.
//     Ablation history (what each change bought): profiles/r01_gemm_lab.md.
//   * Tile order: tile(i) = i*G + (bid%8)*(G/8) + bid/8: the 32 workgroups that share an XCD (blocks b, b+8, ...) work
//     on 32 consecutive tiles (n fastest) at a time, so an activation row panel is fetched into that XCD's L2 once.
#include "tr_common.h"

// {shader cycles, 100-MHz ticks, launches} of gemm_bf16_pc workgroup 8 since the last tr_clock_probe_read: written by -DTR_DIAG_CLOCK builds only
__device__ unsigned long long tr_gemm_clock_probe[3];

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int BK = 64;
constexpr int PBM = 256, PBN = 128;
constexpr int P_STAGE_BYTES = (PBM + PBN) * 128;  // 48 KiB: A rows then W rows, 128 B (64 bf16) per row
constexpr int P_NSTAGE = 3;
constexpr int EPI_DGELU = 17;         // internal: bf16 output times gelu'(outp2[same element]) -- fc2's data gradient with the GELU backward folded in (tr_gemm_dgelu_bf16)
constexpr int EPI_GELU_KEEP = 16;     // internal: TR_EPI_GELU_BF16 plus the pre-activation as a second bf16 output (tr_gemm_gelu_keep_bf16)
// LDS-DMA pieces per wave and K-step in gemm_bf16_persistent: 4 of A, 2 of W = 6 (what its vmcnt immediates count)

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// six 1-KiB LDS-DMA pieces per wave and K-step: 4 of A (rows w*32 + 8j .. +7), 2 of W (rows w*16 + 8j .. +7).
// Source = scalar base + per-lane 32-bit byte offset (saddr form: half the address VGPRs of 64-bit pointers).
// M0 carries the wave-uniform LDS destination; it is compiler-reserved, so it is saved and restored in the statement.
__device__ __forceinline__ void issue_stage(const uint16_t* A, const uint16_t* W, unsigned oa0, unsigned oa1, unsigned oa2,
                                            unsigned oa3, unsigned ow0, unsigned ow1, unsigned lds_a, unsigned lds_w) {
#ifndef TR_ABLATE_NO_DMA
  unsigned keep;
  asm volatile(
      "s_mov_b32 %[keep], m0\n\t"
      "s_mov_b32 m0, %[la]\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[a0], %[A]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[a1], %[A]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[a2], %[A]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[a3], %[A]\n\t"
      "s_mov_b32 m0, %[lw]\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[w0], %[W]\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[w1], %[W]\n\t"
      "s_mov_b32 m0, %[keep]"
      : [keep] "=&s"(keep)
      : [a0] "v"(oa0), [a1] "v"(oa1), [a2] "v"(oa2), [a3] "v"(oa3), [w0] "v"(ow0), [w1] "v"(ow1), [A] "s"(A), [W] "s"(W),
        [la] "s"(lds_a), [lw] "s"(lds_w)
      : "memory", "scc");
#endif
}

// one 1-KiB piece; used to spread a K-step's six pieces between its MFMA groups.  A wave issues in order, and the CU's
// address path takes ~30 cycles per piece (48 pieces per K-step): issued back to back at the top of the step they
// back-pressure every wave for about as long as the step's MFMAs take, so DMA and matrix work serialise
// (profiles/r01_gemm_lab.md: 25.6 us DMA-only + 26.1 us MFMA-only = 49 us together on the qkv shape).
__device__ __forceinline__ void issue_piece(const uint16_t* sbase, unsigned voff, unsigned lds_dst) {
#ifndef TR_ABLATE_NO_DMA
  // M0 is clobbered, not saved/restored (2 scalar instructions less per piece, 12 per K-step and wave): nothing else in this
  // kernel uses M0 -- gfx9 LDS instructions do not read it -- and the bulk issue_stage() sets it itself.
  asm volatile(
      "s_mov_b32 m0, %[ld]\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[o], %[b]"
      :
      : [o] "v"(voff), [b] "s"(sbase), [ld] "s"(lds_dst)
      : "memory", "m0");
#endif
}

// mid-step wait when the previous K-step's epilogue stores (ST per lane) are younger than the DMA group waited for:
// allowed outstanding = 4 pieces of this step + ST stores
template <int ST>
__device__ __forceinline__ void wait_mid() {
  if (ST == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
}

// loads hipcc must not count (see header): the destination is valid only after the matching EPI_WAIT statement.
__device__ __forceinline__ f32x4 asm_load16(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// scalar base + per-lane byte offset + immediate (one offset VGPR serves the four 64-byte-spaced column groups of a row)
template <int IMM>
__device__ __forceinline__ f32x4 asm_load16_so(const float* sbase, unsigned voff) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "i"(IMM) : "memory");
  return v;
}

#define EPI_WAIT_B(N, b) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])::"memory")
#define EPI_WAIT_R(N, r)                                                                                                 \
  asm volatile("s_waitcnt vmcnt(" #N ")"                                                                                 \
               : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]), "+v"(r[0][3]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[1][2]),  \
                 "+v"(r[1][3]), "+v"(r[2][0]), "+v"(r[2][1]), "+v"(r[2][2]), "+v"(r[2][3]), "+v"(r[3][0]), "+v"(r[3][1]),  \
                 "+v"(r[3][2]), "+v"(r[3][3])::"memory")

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_persistent(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                               const float* __restrict__ bias, void* __restrict__ outp,
                                                               const float* __restrict__ aux, int aux_i, int M, int N, int K,
                                                               int nMt, int nNt, unsigned out_bytes) {
  constexpr bool HAS_RV = (EPI == TR_EPI_RESID_F32 || EPI == TR_EPI_PATCH_F32);
  constexpr bool OUT_BF16 = (EPI == TR_EPI_BF16 || EPI == TR_EPI_GELU_BF16);
  constexpr int STORES_PER_TILE = OUT_BF16 ? 8 : 16;   // buffer stores per lane in one epilogue (all 16-byte, whole 128-B lines)
  // 144 KiB ring + 8 x 2 KiB wave-private epilogue staging = the whole 160 KiB LDS
  __shared__ __attribute__((aligned(16))) unsigned char smem[P_NSTAGE * P_STAGE_BYTES + 8 * 2048];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;          // 4 x 2 waves, 64 x 64 outputs each
  const int frow = lane & 15, fq = lane >> 4;
  const int nk = K / BK;

  const int G = gridDim.x, bid = blockIdx.x;
  const int T = nMt * nNt;
  const int toff = (bid & 7) * (G >> 3) + (bid >> 3);
  if (toff >= T) return;
  const int my_tiles = (T - toff + G - 1) / G;
  const int S = my_tiles * nk;                      // flattened K-steps of this workgroup

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned lds_a_w = lds0 + wave * 4096;              // this wave's A pieces inside a stage
  const unsigned lds_w_w = lds0 + PBM * 128 + wave * 2048;  // this wave's W pieces
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)out_bytes, 0x00020000);

  // per-lane source mapping of one DMA piece (8 rows x 8 chunks): LDS position (row, chunk pc) must hold LOGICAL chunk
  // pc ^ ((row>>1)&7)
  const int l3 = lane >> 3, pc = lane & 7;
  int ca[4], cw[2], ra[4], rw[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ra[j] = wave * 32 + j * 8 + l3;
    ca[j] = (pc ^ ((ra[j] >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    rw[j] = wave * 16 + j * 8 + l3;
    cw[j] = (pc ^ ((rw[j] >> 1) & 7)) * 8;
  }

  // ---- load-side cursor (runs P_NSTAGE-1 K-steps ahead of the compute cursor)
  int l_tile = toff, l_kt = 0, l_slot = 0, l_step = 0;
  unsigned oa0, oa1, oa2, oa3, ow0, ow1;   // per-lane byte offsets into A / W of the next K-step to load
#define SET_TILE_PTRS(tile)                                                   \
  do {                                                                        \
    const int tm0_ = ((tile) / nNt) * PBM, tn0_ = ((tile) % nNt) * PBN;       \
    oa0 = ((unsigned)min(tm0_ + ra[0], M - 1) * K + ca[0]) * 2u;              \
    oa1 = ((unsigned)min(tm0_ + ra[1], M - 1) * K + ca[1]) * 2u;              \
    oa2 = ((unsigned)min(tm0_ + ra[2], M - 1) * K + ca[2]) * 2u;              \
    oa3 = ((unsigned)min(tm0_ + ra[3], M - 1) * K + ca[3]) * 2u;              \
    ow0 = ((unsigned)min(tn0_ + rw[0], N - 1) * K + cw[0]) * 2u;              \
    ow1 = ((unsigned)min(tn0_ + rw[1], N - 1) * K + cw[1]) * 2u;              \
  } while (0)
// Every K-step issues exactly DMA_PER_STEP pieces: the next K-step to load, or -- once this workgroup has nothing left to
// load -- dummy pieces (lane-invariant source A[0..7]) into ring slot l_slot, which is free by then (it belonged to
// K-step S-3).  So every counted wait below is ONE asm statement with ONE immediate.  (Two statements on an if/else made
// hipcc copy the asm-loaded registers on one path BEFORE the wait -- garbage; cdna guide section 5.7 item 1.)
#define ADVANCE_LOAD_CURSOR()                                                                                           \
  do {                                                                                                                  \
    if (l_step < S) {                                                                                                   \
      ++l_step;                                                                                                         \
      l_slot = (l_slot == P_NSTAGE - 1) ? 0 : l_slot + 1;                                                               \
      if (++l_kt == nk) {                                                                                               \
        l_kt = 0;                                                                                                       \
        l_tile += G;                                                                                                    \
        if (l_step < S) SET_TILE_PTRS(l_tile);                                                                          \
      } else {                                                                                                          \
        oa0 += 2 * BK; oa1 += 2 * BK; oa2 += 2 * BK; oa3 += 2 * BK; ow0 += 2 * BK; ow1 += 2 * BK;                        \
      }                                                                                                                 \
    }                                                                                                                   \
  } while (0)
#define ISSUE_NEXT()                                                                                                    \
  do {                                                                                                                  \
    const bool real_ = l_step < S;                                                                                      \
    issue_stage(A, real_ ? W : A, real_ ? oa0 : 0u, real_ ? oa1 : 0u, real_ ? oa2 : 0u, real_ ? oa3 : 0u,                \
                real_ ? ow0 : 0u, real_ ? ow1 : 0u, lds_a_w + l_slot * P_STAGE_BYTES, lds_w_w + l_slot * P_STAGE_BYTES); \
    ADVANCE_LOAD_CURSOR();                                                                                              \
  } while (0)

  SET_TILE_PTRS(l_tile);
#pragma unroll
  for (int p = 0; p < P_NSTAGE - 1; ++p) ISSUE_NEXT();

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Fragment registers: the first 32-deep half of a K-step (ks = 0) is read one half-step AHEAD, right after the barrier that
  // publishes its slot (which sits in the MIDDLE of the previous K-step); the second half (ks = 1) is read at the top of
  // its own K-step and used after the mid-step barrier.  So every LDS read has 16 MFMAs (x2 waves per SIMD) to hide under,
  // instead of all eight waves reading at once behind a top-of-step barrier with the matrix pipe idle (in-kernel stamps:
  // ~500 of ~1700 cycles per K-step).
  bf16x8 wA[4], aA[4], wC[4], aC[4];
#define READ_FRAGS(WF, AF, slot, ks)                                                                          \
  do {                                                                                                        \
    const unsigned char* sa_ = smem + (slot) * P_STAGE_BYTES;                                                 \
    const unsigned char* sw_ = sa_ + PBM * 128;                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
      WF[i_] = *reinterpret_cast<const bf16x8*>(sw_ + swz(wn * 64 + i_ * 16 + frow, 4 * (ks) + fq));          \
      AF[i_] = *reinterpret_cast<const bf16x8*>(sa_ + swz(wm * 64 + i_ * 16 + frow, 4 * (ks) + fq));          \
    }                                                                                                         \
  } while (0)
#ifdef TR_ABLATE_NO_MFMA
#define MFMA_GROUP(WF, AF, i_) asm volatile("" ::"v"(WF[i_]), "v"(AF[i_]))
#else
#define MFMA_GROUP(WF, AF, i_)                                                                               \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                            \
      acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[i_], AF[j_], acc[i_][j_], 0, 0, 0)
#endif

  int c_tile = toff, c_kt = 0, c_slot = 0;
  int st_prev = 0;        // epilogue stores issued at the END of the previous K-step (younger than every DMA piece before them)

  // K-step 0: its DMA group is the older of the two just issued
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  READ_FRAGS(wA, aA, 0, 0);

// One K-step.  (W0,A0): this step's ks=0 fragments, already (being) read; (W0N,A0N): the next step's, read at the mid barrier.
#define GEMM_STEP(W0, A0, W0N, A0N)                                                                                          \
  do {                                                                                                                       \
    const bool tile_end = (c_kt == nk - 1);                                                                                  \
    const bool tile_begin = (c_kt == 0);                                                                                     \
    f32x4 bv[4];     /* bias of this tile's columns (last K-step only) */                                                    \
    f32x4 rv[4][4];  /* residual / pos_embed rows (first K-step only)   */                                                   \
    STAMP(0);                                                                                                                \
    READ_FRAGS(wC, aC, c_slot, 1);                                                                                           \
    /* this K-step's six DMA pieces (the K-step two ahead, into the slot K-step g-1 used; dummies in the tail) */            \
    const bool real = l_step < S;                                                                                            \
    const uint16_t* srcW = real ? W : A;                                                                                     \
    const unsigned q0 = real ? oa0 : 0u, q1 = real ? oa1 : 0u, q2 = real ? oa2 : 0u, q3 = real ? oa3 : 0u;                   \
    const unsigned q4 = real ? ow0 : 0u, q5 = real ? ow1 : 0u;                                                               \
    const unsigned dA = lds_a_w + l_slot * P_STAGE_BYTES, dW = lds_w_w + l_slot * P_STAGE_BYTES;                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                       \
    MFMA_GROUP(W0, A0, 0); issue_piece(A, q0, dA);        __builtin_amdgcn_sched_barrier(0);                                 \
    MFMA_GROUP(W0, A0, 1); issue_piece(A, q1, dA + 1024); __builtin_amdgcn_sched_barrier(0);                                 \
    MFMA_GROUP(W0, A0, 2); issue_piece(A, q2, dA + 2048); __builtin_amdgcn_sched_barrier(0);                                 \
    MFMA_GROUP(W0, A0, 3); issue_piece(A, q3, dA + 3072); __builtin_amdgcn_sched_barrier(0);                                 \
    STAMP(1);                                                                                                                \
    const int next_slot = (c_slot == P_NSTAGE - 1) ? 0 : c_slot + 1;                                                         \
    if (g + 1 < S) {                                                                                                         \
      /* DMA group g+1 (issued during K-step g-1) has landed once only the ops issued after it remain: the 4 pieces above */ \
      /* and the previous K-step's epilogue stores, if it ended a tile */                                                    \
      if (st_prev) wait_mid<STORES_PER_TILE>(); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                        \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  /* own reads of slot g are done: it may be refilled after this */ \
      __builtin_amdgcn_s_barrier();                                                                                          \
      asm volatile("" ::: "memory");                                                                                         \
      STAMP(2);                                                                                                              \
    }                                                                                                                        \
    /* unconditional on purpose (after the last K-step it reads a stale slot, unused): a conditional assignment would keep */ \
    /* BOTH ks=0 register sets live around the whole loop */                                                                 \
    READ_FRAGS(W0N, A0N, next_slot, 0);                                                                                      \
    /* epilogue-side loads: issued HERE, i.e. older than pieces 4,5 below, so waiting for them at the end of the step */     \
    /* is vmcnt(2) and they never enter a DMA count (they are older than this group's last piece) */                         \
    if (HAS_RV && tile_begin) {                                                                                              \
      const int m0 = (c_tile / nNt) * PBM, n0 = (c_tile % nNt) * PBN;                                                        \
      /* N % 64 == 0 here (launcher): a wave's 64 columns are all valid or all invalid -> one clamped base + immediates */   \
      const unsigned cbase = (unsigned)min(n0 + wn * 64, N - 64) + 4 * fq;                                                   \
      const float* rbase = (EPI == TR_EPI_PATCH_F32) ? aux : reinterpret_cast<const float*>(outp);                           \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                        \
        const int mc = min(m0 + wm * 64 + j * 16 + frow, M - 1);                                                             \
        const unsigned rrow = (EPI == TR_EPI_PATCH_F32) ? (unsigned)(1 + mc % aux_i) : (unsigned)mc;                         \
        const unsigned voff = (rrow * (unsigned)N + cbase) * 4u;                                                             \
        rv[j][0] = asm_load16_so<0>(rbase, voff);                                                                            \
        rv[j][1] = asm_load16_so<64>(rbase, voff);                                                                           \
        rv[j][2] = asm_load16_so<128>(rbase, voff);                                                                          \
        rv[j][3] = asm_load16_so<192>(rbase, voff);                                                                          \
      }                                                                                                                      \
    }                                                                                                                        \
    if (tile_end) {                                                                                                          \
      const int n0 = (c_tile % nNt) * PBN;                                                                                   \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) bv[i] = asm_load16(bias + min(n0 + wn * 64 + i * 16 + 4 * fq, N - 4));   \
    }                                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                       \
    MFMA_GROUP(wC, aC, 0); issue_piece(srcW, q4, dW);        __builtin_amdgcn_sched_barrier(0);                              \
    MFMA_GROUP(wC, aC, 1); issue_piece(srcW, q5, dW + 1024); __builtin_amdgcn_sched_barrier(0);                              \
    MFMA_GROUP(wC, aC, 2);                                                                                                   \
    MFMA_GROUP(wC, aC, 3);                                                                                                   \
    STAMP(3);                                                                                                                \
    ADVANCE_LOAD_CURSOR();                                                                                                   \
    c_slot = next_slot;                                                                                                      \
    st_prev = 0;                                                                                                             \
    if (HAS_RV && tile_begin) {                                                                                              \
      EPI_WAIT_R(2, rv);                                                                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[i][j] += rv[j][i];                                                 \
    }                                                                                                                        \
    if (!tile_end) {                                                                                                         \
      ++c_kt;                                                                                                                \
    } else {                                                                                                                 \
      EPI_WAIT_B(2, bv);                                                                                                     \
      epilogue_store(bv);                                                                                                    \
      st_prev = 1;                                                                                                           \
      c_kt = 0;                                                                                                              \
      c_tile += G;                                                                                                           \
    }                                                                                                                        \
    STAMP(4);                                                                                                                \
  } while (0)

#ifdef TR_DIAG_STAMPS
  unsigned long long ts_[5];
#define STAMP(k) ts_[k] = __builtin_amdgcn_s_memtime()
#else
#define STAMP(k) do { } while (0)
#endif

  // The accumulator layout (lane = row, 4 consecutive columns) would store 16 rows x 32 B per instruction; the store path
  // prices an instruction by the LINES it touches (stamps: 16 such stores per lane = ~7.8k cycles per tile, half the tile).
  // So each 16-row slab goes through 2 KiB of wave-private LDS (16-byte chunk index XOR row&7: conflict-free for the
  // b128 traffic, 2-way for the bf16 b64 writes) and leaves as 16-byte stores covering whole 128-byte lines, 8 rows each.
  auto epilogue_store = [&](f32x4 (&bv)[4]) __attribute__((always_inline)) {
#ifdef TR_ABLATE_NO_EPI
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(acc[i][j]), "v"(bv[i])); acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#else
    unsigned char* stg = smem + P_NSTAGE * P_STAGE_BYTES + wave * 2048;
    const int m0 = (c_tile / nNt) * PBM, n0 = (c_tile % nNt) * PBN;
    const int rrow = lane >> 3, rch = lane & 7;             // read-back mapping: 8 rows x 8 chunks per instruction
    constexpr int NPASS = OUT_BF16 ? 1 : 2;                 // a slab row is 64 bf16 = 128 B, or 2 x (32 fp32 = 128 B)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int ip = 0; ip < NPASS; ++ip) {
#pragma unroll
        for (int ii = 0; ii < (OUT_BF16 ? 4 : 2); ++ii) {
          const int i = OUT_BF16 ? ii : 2 * ip + ii;
          float v0 = acc[i][j][0] + bv[i][0], v1 = acc[i][j][1] + bv[i][1], v2 = acc[i][j][2] + bv[i][2], v3 = acc[i][j][3] + bv[i][3];
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (OUT_BF16) {
#ifndef TR_ABLATE_NO_GELU
            if (EPI == TR_EPI_GELU_BF16) {
              const f32x2 g01 = gelu2(f32x2{v0, v1}), g23 = gelu2(f32x2{v2, v3});
              v0 = g01[0]; v1 = g01[1]; v2 = g23[0]; v3 = g23[1];
            }
#endif
            u32x2 pk;
            pk[0] = pack_bf16x2(v0, v1);
            pk[1] = pack_bf16x2(v2, v3);
            *reinterpret_cast<u32x2*>(stg + frow * 128 + (((2 * i + (fq >> 1)) ^ (frow & 7)) << 4) + (fq & 1) * 8) = pk;
          } else {
            *reinterpret_cast<f32x4*>(stg + frow * 128 + (((4 * ii + fq) ^ (frow & 7)) << 4)) = f32x4{v0, v1, v2, v3};
          }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int row = r * 8 + rrow;
          const u32x4 pk = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((rch ^ (row & 7)) << 4));
          const int m = m0 + wm * 64 + j * 16 + row;
          const int n = n0 + wn * 64 + (OUT_BF16 ? rch * 8 : ip * 32 + rch * 4);
          size_t orow = (size_t)m;
          if (EPI == TR_EPI_PATCH_F32) orow = orow + orow / aux_i + 1;   // row (b,p) -> b*(P+1) + 1 + p
          bool ok = (m < M) && (n < N);
#ifdef TR_ABLATE_NO_STORE
          ok = ok && (aux_i == 0x7fffffff);
#endif
          // out-of-range lanes get an offset beyond num_records: the buffer bounds check drops their store
          const unsigned off = ok ? (unsigned)((orow * N + n) * (OUT_BF16 ? 2 : 4)) : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b128(pk, orsrc, off, 0, 0);
        }
      }
    }
#endif
  };

#ifdef TR_DIAG_STAMPS
#define DUMP_STAMPS()                                                                                         \
  do {                                                                                                        \
    if (aux != nullptr && bid == 8 && g < 64 && lane == 0 && (wave == 0 || wave == 7)) {                      \
      unsigned long long* st = reinterpret_cast<unsigned long long*>(const_cast<float*>(aux)) + ((wave ? 64 : 0) + g) * 5; \
      st[0] = ts_[0]; st[1] = ts_[1]; st[2] = ts_[2]; st[3] = ts_[3]; st[4] = ts_[4];                        \
    }                                                                                                         \
  } while (0)
#else
#define DUMP_STAMPS() do { } while (0)
#endif

  for (int g = 0; g < S; ++g) {
    // the ks=0 registers are dead after the first four MFMA groups, so the read-ahead at the mid barrier refills them in place
    GEMM_STEP(wA, aA, wA, aA);
    DUMP_STAMPS();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // dummy DMA into this workgroup's LDS must not outlive it
#undef GEMM_STEP
#undef READ_FRAGS
#undef MFMA_GROUP
#undef STAMP
#undef DUMP_STAMPS
#undef SET_TILE_PTRS
#undef ISSUE_NEXT
#undef ADVANCE_LOAD_CURSOR
}


// ------------------------------------------------------------------------------------------------------------------
// Producer/consumer variant for the bf16-output epilogues (qkv, proj, fc1+GELU, fc2 -- every per-block GEMM).
//
// Same tiles, ring, swizzle and epilogue as gemm_bf16_persistent above, but the DMA is issued by FOUR DEDICATED LOADER
// WAVES (waves 8-11, one per SIMD) and the eight MFMA waves never touch the vector-memory pipe inside the K-loop.
// Why: a wave issues in order, and the CU's address path needs ~27 cycles per 1-KiB piece -- 48 pieces per K-step =
// 1300-1440 cycles, more than the step's 1024 MFMA cycles.  With the pieces inside the MFMA waves' streams (even
// one piece per 4-MFMA group) every wave keeps stalling on a full address queue with its MFMAs queued behind the stall
// (profiles/r01_gemm_lab.md: main loop 40.4 us with DMA vs 25.4 us without, DMA alone 25.6 us).  A loader wave can sit
// in that stall all day.  Three waves per SIMD -> 168 VGPRs per wave, which the bf16 epilogues fit (no residual registers).
// Barrier protocol (one s_barrier per K-step, all 12 waves): at barrier B_g the loaders have waited for DMA group g
// (vmcnt(12): group g+1 may still fly) and the MFMA waves have finished READING slot (g-1)%3 (lgkmcnt(0));
// after it the loaders refill that slot with group g+2 and the MFMA waves read slot g%3.
#ifndef TR_PC_LOADERS
#define TR_PC_LOADERS 4     // loader waves per workgroup: 4 (one per SIMD) or 2 (lab: two SIMDs keep 176 registers per lane free for another launch's waves)
#endif
constexpr int PC_LW = TR_PC_LOADERS, PC_SL = 4 / PC_LW;      // loader waves; 64-row slices of the tile per loader wave
constexpr int PC_THREADS = 512 + 64 * PC_LW;
template <int EPI>
__global__ __launch_bounds__(PC_THREADS, 3) void gemm_bf16_pc(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                       const float* __restrict__ bias, uint16_t* __restrict__ outp,
                                                       uint16_t* __restrict__ outp2, int M, int N, int K, int nMt, int nNt,
                                                       unsigned out_bytes) {
  // EPI_GELU_KEEP (training forward of fc1): outp2 receives the pre-activation, outp its GELU -- one pass instead of a GEMM that
  // writes the pre-activation and an elementwise kernel that re-reads it (0.37 ms of a 12.5 ms DeiT-S step)
  // EPI_DGELU (backward of fc2 -> GELU): outp2 is READ -- the kept pre-activation, laid out like the output; every finished 16-byte output
  // line is multiplied by gelu'(pre) before it is stored, instead of an elementwise pass over [M, hidden] afterwards
  static_assert(EPI == TR_EPI_BF16 || EPI == TR_EPI_GELU_BF16 || EPI == EPI_GELU_KEEP || EPI == EPI_DGELU, "bf16-output epilogues only");
  __shared__ __attribute__((aligned(16))) unsigned char smem[P_NSTAGE * P_STAGE_BYTES + 8 * 2048];
#ifdef TR_LAB_STAGGER   // lab only: every other workgroup of an XCD starts late, so that the epilogues (all of a tile's HBM traffic) of one half of the chip fall into the K-loops of the other
  if ((blockIdx.x >> 3) & 1)
    for (int i = 0; i < TR_LAB_STAGGER; ++i) __builtin_amdgcn_s_sleep(64);
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = K / BK;
  const int G = gridDim.x, bid = blockIdx.x;
  const int T = nMt * nNt;
  const int toff = (bid & 7) * (G >> 3) + (bid >> 3);
  // Tail round.  T tiles on G workgroups = F full rounds + R = T - F*G tiles that would occupy R workgroups for a whole tile time
  // while G - R idle (591 tiles of the N = 384 GEMMs at B = 256: 2.31 rounds paid as 3).  When R <= G/2 the tail tiles are cut into
  // two 128-row HALF tiles, one per workgroup (2R <= G): same ring, same barrier protocol, the loaders fetch 128 instead of 256
  // activation rows (8 instead of 12 pieces per wave and K-step) and only the MFMA waves of rows 0..127 (wm < 2, one per SIMD) work.
#ifdef TR_ABLATE_NO_W_DMA
  const bool split = false;
#else
  const int F = T / G, R = T - F * G;
  const bool split = R > 0 && 2 * R <= G;
#endif
  const int my_full = split ? T / G : (toff < T ? (T - toff + G - 1) / G : 0);
  const bool has_half = split && toff < 2 * (T - (T / G) * G);
  const int units = my_full + (has_half ? 1 : 0);
  if (units == 0) return;
  const int half_tile = (T / G) * G + (toff >> 1), half_sel = toff & 1;
  const int S_full = my_full * nk, S = units * nk;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  if (wave >= 8) {
    // ================================ loader wave: 12 pieces per K-step and slice (A rows lw*64.., W rows lw*32..), PC_SL slices
    const int lw0 = wave - 8;
    const int l3 = lane >> 3, pc = lane & 7;
    // LDS position (row, chunk pc) must hold LOGICAL chunk pc ^ ((row>>1)&7); rows lw*64 + 8j + l3 and lw*32 + 8j + l3:
    // (row>>1)&7 = (4j + (l3>>1)) & 7 for both (lw*64, lw*32 are multiples of 16)
    unsigned o[PC_SL][12];
    int l_unit = 0, l_kt = 0, l_slot = 0, l_step = 0;
    auto set_unit = [&](int u) __attribute__((always_inline)) {
      const bool half = u >= my_full;
      const int tile = half ? half_tile : toff + u * G;
      const int tm0 = (tile / nNt) * PBM + (half ? 128 * half_sel : 0), tn0 = (tile % nNt) * PBN;
      const int rpw = half ? 32 : 64;       // activation rows per slice
#pragma unroll
      for (int q = 0; q < PC_SL; ++q) {
        const int lw = lw0 + q * PC_LW;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = (pc ^ ((4 * j + (l3 >> 1)) & 7)) * 8;
          o[q][j] = ((unsigned)min(tm0 + lw * rpw + 8 * j + l3, M - 1) * K + c) * 2u;     // j >= 4 unused by a half tile
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = (pc ^ ((4 * j + (l3 >> 1)) & 7)) * 8;
          o[q][8 + j] = ((unsigned)min(tn0 + lw * 32 + 8 * j + l3, N - 1) * K + c) * 2u;
        }
      }
    };
    auto issue_group = [&]() __attribute__((always_inline)) {
      const bool real = l_step < S;
      const bool hmode = has_half && l_step >= S_full;     // the half tile is a workgroup's LAST unit; the dummy groups behind it keep its size
      const uint16_t* sW = real ? W : A;
#pragma unroll
      for (int q = 0; q < PC_SL; ++q) {
        const int lw = lw0 + q * PC_LW;
        const unsigned da = lds0 + l_slot * P_STAGE_BYTES + lw * (hmode ? 4096 : 8192);
        const unsigned dw = lds0 + l_slot * P_STAGE_BYTES + PBM * 128 + lw * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_piece(A, real ? o[q][j] : 0u, da + j * 1024);
        if (!hmode) {
#pragma unroll
          for (int j = 4; j < 8; ++j) issue_piece(A, real ? o[q][j] : 0u, da + j * 1024);
        }
#ifndef TR_ABLATE_NO_W_DMA    // lab only: what would the K-loop do if the weight panel stayed in LDS (feed 32 KB instead of 48 KB per K-step)?
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_piece(sW, real ? o[q][8 + j] : 0u, dw + j * 1024);
#endif
      }
      if (real) {
        ++l_step;
        l_slot = (l_slot == P_NSTAGE - 1) ? 0 : l_slot + 1;
        if (++l_kt == nk) {
          l_kt = 0;
          ++l_unit;
          if (l_step < S) set_unit(l_unit);
        } else {
#pragma unroll
          for (int q = 0; q < PC_SL; ++q)
#pragma unroll
            for (int j = 0; j < 12; ++j) o[q][j] += 2 * BK;
        }
      }
    };
    set_unit(0);
    issue_group();      // group 0 -> slot 0
    issue_group();      // group 1 -> slot 1   (dummies if S < 2)
    for (int g = 0; g < S; ++g) {
      // group g landed; group g+1 -- real or dummy, 12 pieces per slice or a half tile's 8 -- may still fly
#ifdef TR_ABLATE_NO_W_DMA
      if (PC_SL == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#else
      if (has_half && g + 1 >= S_full) {
        if (PC_SL == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      } else {
        if (PC_SL == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      }
#endif
      __builtin_amdgcn_s_barrier();                        // B_g: the MFMA waves are done reading slot (g-1)%3 == (g+2)%3
      if (g + 1 < S) issue_group();                        // group g+2 -> that slot (dummy pieces once nothing is left to load)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ================================ MFMA wave
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc2 = __builtin_amdgcn_make_buffer_rsrc(EPI == EPI_GELU_KEEP || EPI == EPI_DGELU ? outp2 : outp, 0, (int)out_bytes, 0x00020000);
  // The accumulators START at the bias of the tile's columns (instead of zero + a bias add in the epilogue: 64 VALU adds per lane
  // and tile less, in the one phase where the matrix pipe idles); the next tile's bias is fetched under the last K-step.
  auto load_bias = [&](f32x4 (&bv)[4], int tile) __attribute__((always_inline)) {
    const int n0 = (tile % nNt) * PBN;
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4*>(bias + min(n0 + wn * 64 + i * 16 + 4 * fq, N - 4));
  };
  f32x4 acc[4][4];
  {
    f32x4 b0[4];
    if (EPI == EPI_DGELU) {                 // a data gradient has no bias
#pragma unroll
      for (int i = 0; i < 4; ++i) b0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      load_bias(b0, my_full > 0 ? toff : half_tile);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = b0[i];
  }
  bf16x8 wA[4], aA[4], wC[4], aC[4];
#define READ_FRAGS(WF, AF, slot, ks)                                                                          \
  do {                                                                                                        \
    const unsigned char* sa_ = smem + (slot) * P_STAGE_BYTES;                                                 \
    const unsigned char* sw_ = sa_ + PBM * 128;                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
      WF[i_] = *reinterpret_cast<const bf16x8*>(sw_ + swz(wn * 64 + i_ * 16 + frow, 4 * (ks) + fq));          \
      AF[i_] = *reinterpret_cast<const bf16x8*>(sa_ + swz(wm * 64 + i_ * 16 + frow, 4 * (ks) + fq));          \
    }                                                                                                         \
  } while (0)
#ifdef TR_ABLATE_NO_MFMA
#define MFMA_GROUP(WF, AF, i_) asm volatile("" ::"v"(WF[i_]), "v"(AF[i_]))
#else
#define MFMA_GROUP(WF, AF, i_)                                                                               \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                            \
      acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[i_], AF[j_], acc[i_][j_], 0, 0, 0)
#endif

  int c_tile = my_full > 0 ? toff : half_tile, c_kt = 0, c_slot = 0;
  int gs = 0;                                // K-steps done, over all units of this workgroup
#ifdef TR_DIAG_CLOCK
  const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  __builtin_amdgcn_s_barrier();            // B_0
  asm volatile("" ::: "memory");
  // One K-step (+ the epilogue after a tile's last one).  HALF: the unit is this workgroup's 128-row half tile (its last unit).
  // The loop is instantiated twice (full tiles, then the half tile) so that the full-tile path keeps its register allocation.
  auto k_step = [&](const bool HALF) __attribute__((always_inline)) {
    // the half-tile copy of the loop recomputes the lane id (2 instructions) instead of keeping lane-derived values alive across the
    // full-tile loop (the allocator spilled one of them to scratch otherwise)
    const int lane_h = HALF ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : lane;
    const int frow = lane_h & 15, fq = lane_h >> 4;
    const bool tile_end = (c_kt == nk - 1);
    READ_FRAGS(wC, aC, c_slot, 1);
    f32x4 bv[4];                           // bias of the NEXT tile (the tail workgroups reload their last tile's: unused)
    u32x4 pr[8];                           // EPI_DGELU: pre-activation lines of THIS tile
    int n_tile = c_tile;
    if (tile_end) {
      if (!HALF) n_tile = (gs + 1 < S_full) ? c_tile + G : (has_half ? half_tile : c_tile);
      asm volatile("" : "+s"(n_tile));      // keeps the address arithmetic of the bias loads HERE (hoisted, it cost 8 VGPRs for a whole K-step)
      if (EPI == EPI_DGELU) {
        // no bias: its registers (and 16 more) hold THIS tile's eight pre-activation lines per lane instead, requested a whole K-step
        // before the epilogue multiplies them in -- issued from inside the epilogue, every slab waited out the full memory latency
        const int m0 = (c_tile / nNt) * PBM + (HALF ? 128 * half_sel : 0) + wm * 64 + (lane_h >> 3);
        const int n0 = (c_tile % nNt) * PBN + wn * 64 + (lane_h & 7) * 8;
        const unsigned o0 = ((unsigned)m0 * (unsigned)N + (unsigned)n0) * 2u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {       // slabs 0 and 1 now (the bias registers); slabs 2 and 3 once the last MFMAs have freed their fragments
          const bool ok = (m0 + 8 * q < M) && (n0 < N);
          pr[q] = __builtin_amdgcn_raw_buffer_load_b128(orsrc2, ok ? o0 + (unsigned)q * 16u * (unsigned)N : 0x80000000u, 0, 0);   // out of range: zeros
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        load_bias(bv, n_tile);
      }
    }
    MFMA_GROUP(wA, aA, 0);
    MFMA_GROUP(wA, aA, 1);
    MFMA_GROUP(wA, aA, 2);
    MFMA_GROUP(wA, aA, 3);
    const int next_slot = (c_slot == P_NSTAGE - 1) ? 0 : c_slot + 1;
    if (gs + 1 < S) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own reads of slot g%3 are done: the loaders may refill it after B_{g+2}
      __builtin_amdgcn_s_barrier();                        // B_{g+1}: group g+1 has landed
      asm volatile("" ::: "memory");
    }
    READ_FRAGS(wA, aA, next_slot, 0);      // unconditional: after the last K-step it reads a stale slot, unused
    __builtin_amdgcn_sched_barrier(0);
    MFMA_GROUP(wC, aC, 0);
    MFMA_GROUP(wC, aC, 1);
    MFMA_GROUP(wC, aC, 2);
    MFMA_GROUP(wC, aC, 3);
    c_slot = next_slot;
    ++gs;
    if (!tile_end) {
      ++c_kt;
      return;
    }
    // ---- epilogue (same LDS-staged full-line stores as gemm_bf16_persistent)
#ifdef TR_ABLATE_NO_EPI
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(acc[i][j])); acc[i][j] = bv[i]; }
#else
    {
      // Software-pipelined over the four 16-row slabs: slab j is staged (bf16 -> wave-private LDS) while slab j-1's read-back is
      // still in flight, and stored after that.  LDS operations of one wave execute in order, so re-using the one 2-KiB stage
      // is safe (the read-back of j-1 is queued before the writes of j), and only one LDS round trip per tile is exposed
      // instead of four.
      unsigned char* stg = smem + P_NSTAGE * P_STAGE_BYTES + wave * 2048;
      const int rrow = lane_h >> 3, rch = lane_h & 7;
      const int m_first = (c_tile / nNt) * PBM + (HALF ? 128 * half_sel : 0) + wm * 64 + rrow;   // row of (slab 0, half 0); +8 per half-slab
      const int n = (c_tile % nNt) * PBN + wn * 64 + rch * 8;
      const unsigned off_first = ((unsigned)m_first * (unsigned)N + (unsigned)n) * 2u;      // < 2 GiB (launcher)
      const unsigned off_step = 16u * (unsigned)N;                                          // 8 rows of bf16
      const unsigned char* rd = stg + rrow * 128 + ((rch ^ rrow) << 4);                     // (row & 7) == rrow for both halves
      // PRE: stage the pre-activation itself and leave the accumulators alone (EPI_GELU_KEEP's first pass over a slab)
      auto stage_ = [&](int j, const bool PRE) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
          if (!PRE) acc[i][j] = bv[i];
          if (!PRE && (EPI == TR_EPI_GELU_BF16 || EPI == EPI_GELU_KEEP)) {
            if (EPI == EPI_GELU_KEEP) {        // GELU of the ROUNDED pre-activation: what the backward differentiates (tr_gelu_bwd_bf16 reads it)
              const unsigned r01 = pack_bf16x2(v0, v1), r23 = pack_bf16x2(v2, v3);
              v0 = __builtin_bit_cast(float, r01 << 16); v1 = __builtin_bit_cast(float, r01 & 0xffff0000u);
              v2 = __builtin_bit_cast(float, r23 << 16); v3 = __builtin_bit_cast(float, r23 & 0xffff0000u);
            }
            const f32x2 g01 = gelu2(f32x2{v0, v1}), g23 = gelu2(f32x2{v2, v3});
            v0 = g01[0]; v1 = g01[1]; v2 = g23[0]; v3 = g23[1];
          }
          u32x2 pk;
          pk[0] = pack_bf16x2(v0, v1);
          pk[1] = pack_bf16x2(v2, v3);
          // 16-byte chunk XOR row&7 spreads the rows over the banks; rows r and r+8 share a chunk, so they take OPPOSITE 8-byte
          // halves of it (tools/lds_sim.py: 4 instead of 8 cycles per ds_write_b64) -- undone for free in read_back
          *reinterpret_cast<u32x2*>(stg + frow * 128 + (((2 * i + (fq >> 1)) ^ (frow & 7)) << 4) + (((fq & 1) ^ (frow >> 3)) << 3)) = pk;
        }
      };
      auto stage = [&](int j) __attribute__((always_inline)) { stage_(j, false); };
      auto read_back = [&](u32x4 (&ln)[2]) __attribute__((always_inline)) {
        ln[0] = *reinterpret_cast<const u32x4*>(rd);
        const u32x4 t = *reinterpret_cast<const u32x4*>(rd + 1024);       // rows 8..15: halves swapped
        ln[1] = u32x4{t[2], t[3], t[0], t[1]};
      };
      auto store_ = [&](int j, const u32x4 (&ln)[2], const bool SECOND) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          bool ok = (m_first + 8 * (2 * j + r) < M) && (n < N);
#ifdef TR_ABLATE_NO_STORE
          ok = ok && (K == 0x7fffffff);
#endif
          // out-of-range lanes get an offset beyond num_records: the buffer bounds check drops their store
          const unsigned off = ok ? off_first + (unsigned)(2 * j + r) * off_step : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b128(ln[r], SECOND ? orsrc2 : orsrc, off, 0, 0);
        }
      };
      auto store = [&](int j, const u32x4 (&ln)[2]) __attribute__((always_inline)) { store_(j, ln, false); };
      u32x4 lnA[2], lnB[2];
      if (EPI == EPI_DGELU) {
        auto scale = [&](u32x4 (&ln)[2], int j) __attribute__((always_inline)) {
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const unsigned int d[4] = {ln[r][0], ln[r][1], ln[r][2], ln[r][3]};
            const unsigned int x[4] = {pr[2 * j + r][0], pr[2 * j + r][1], pr[2 * j + r][2], pr[2 * j + r][3]};
            unsigned int o[4];
            dgelu_line(d, x, o);
            ln[r] = u32x4{o[0], o[1], o[2], o[3]};
          }
        };
#pragma unroll
        for (int q = 4; q < 8; ++q) {
          const bool ok = (m_first + 8 * q < M) && (n < N);
          pr[q] = __builtin_amdgcn_raw_buffer_load_b128(orsrc2, ok ? off_first + (unsigned)q * off_step : 0x80000000u, 0, 0);
        }
        stage(0); read_back(lnA);
        stage(1); scale(lnA, 0); store(0, lnA); read_back(lnB);
        stage(2); scale(lnB, 1); store(1, lnB); read_back(lnA);
        stage(3); scale(lnA, 2); store(2, lnA); read_back(lnB);
        scale(lnB, 3); store(3, lnB);
      } else if (EPI == EPI_GELU_KEEP) {
        // every slab passes the stage twice: pre-activation (to outp2), then its GELU (to outp); same one-ahead pipelining
        stage_(0, true); read_back(lnA);
        stage_(0, false); store_(0, lnA, true); read_back(lnB);
#pragma unroll
        for (int j = 1; j < 4; ++j) {
          stage_(j, true); store_(j - 1, lnB, false); read_back(lnA);
          stage_(j, false); store_(j, lnA, true); read_back(lnB);
        }
        store_(3, lnB, false);
      } else {
        stage(0); read_back(lnA);
        stage(1); store(0, lnA); read_back(lnB);
        stage(2); store(1, lnB); read_back(lnA);
        stage(3); store(2, lnA); read_back(lnB);
        store(3, lnB);
      }
    }
#endif
    c_kt = 0;
    c_tile = n_tile;
  };
  READ_FRAGS(wA, aA, 0, 0);
  for (int g = 0; g < S_full; ++g) k_step(false);
  if (has_half) {
    if (wm < 2) {
      for (int g = 0; g < nk; ++g) k_step(true);
    } else {
      // rows 128..255 do not exist in a half tile: these waves only keep the barrier count (one per K-step, none after the last)
      for (int g = 0; g < nk; ++g) {
        if (gs + 1 < S) __builtin_amdgcn_s_barrier();
        ++gs;
      }
    }
  }
#ifdef TR_DIAG_CLOCK
  // diagnostic build only (tools/lab/clock_probe.py): shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz of workgroup 8's K-loops, summed
  // over the launches since the last read -- into a device symbol: no output buffer is touched, so the build runs inside the model
  if (bid == 8 && tid == 0) {
    atomicAdd(&tr_gemm_clock_probe[0], __builtin_amdgcn_s_memtime() - ck0);
    atomicAdd(&tr_gemm_clock_probe[1], __builtin_amdgcn_s_memrealtime() - rt0);
    atomicAdd(&tr_gemm_clock_probe[2], 1ull);
  }
#endif
#undef READ_FRAGS
#undef MFMA_GROUP
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
  size_t out_rows = (size_t)M;
  if (epilogue == TR_EPI_PATCH_F32) {
    TR_REQUIRE(aux && aux_i > 0 && M % aux_i == 0 && tr_aligned16(aux), TR_ERR_SHAPE,
               "tr_gemm_bf16: PATCH epilogue needs pos_embed and P | M");
    out_rows = (size_t)(M / aux_i) * (aux_i + 1);
  }
  const size_t esz = (epilogue == TR_EPI_BF16 || epilogue == TR_EPI_GELU_BF16) ? 2 : 4;
  const size_t out_bytes = out_rows * (size_t)N * esz;
  TR_REQUIRE(out_bytes < ((size_t)1 << 31), TR_ERR_SHAPE, "tr_gemm_bf16: output of %zu bytes exceeds the 2 GiB range of the store offsets",
             out_bytes);
  TR_REQUIRE((size_t)M * K * 2 < ((size_t)1 << 32) && (size_t)N * K * 2 < ((size_t)1 << 32), TR_ERR_SHAPE,
             "tr_gemm_bf16: operands must be smaller than 4 GiB (32-bit DMA offsets)");
  if (esz == 2) TR_REQUIRE(N % 8 == 0, TR_ERR_SHAPE, "tr_gemm_bf16: bf16 outputs need N %% 8 == 0 (got %d)", N);
  if (epilogue == TR_EPI_RESID_F32 || epilogue == TR_EPI_PATCH_F32)
    TR_REQUIRE(N % 64 == 0, TR_ERR_SHAPE, "tr_gemm_bf16: residual/patch epilogues need N %% 64 == 0 (got %d)", N);
  hipStream_t st = static_cast<hipStream_t>(s);
  {
    static const char* const kname[5] = {"gemm_bf16_pc<EPI_BF16>", "gemm_bf16_pc<EPI_GELU_BF16>", "gemm_bf16_persistent<EPI_RESID_F32>",
                                         "gemm_bf16_persistent<EPI_F32>", "gemm_bf16_persistent<EPI_PATCH_F32>"};
    if (epilogue >= 0 && epilogue < 5)
      tr_prof_note(kname[epilogue], 2.0 * M * N * K, 2.0 * ((double)M * K + (double)N * K) + (double)esz * M * N);
  }
  const int nMt = (M + PBM - 1) / PBM, nNt = (N + PBN - 1) / PBN;
  // one persistent workgroup per CU (all 160 KiB of LDS each); 256 CUs on MI355X.  A multiple of 8 keeps the XCD grouping.
  dim3 grid(256), block(512);
  if (epilogue == TR_EPI_BF16 || epilogue == TR_EPI_GELU_BF16) {
    // the per-block GEMMs: 8 MFMA waves + 4 loader waves
    if (epilogue == TR_EPI_BF16)
      hipLaunchKernelGGL(gemm_bf16_pc<TR_EPI_BF16>, grid, dim3(PC_THREADS), 0, st, A, W, bias, static_cast<uint16_t*>(out),
                         static_cast<uint16_t*>(nullptr), M, N, K, nMt, nNt, (unsigned)out_bytes);
    else
      hipLaunchKernelGGL(gemm_bf16_pc<TR_EPI_GELU_BF16>, grid, dim3(PC_THREADS), 0, st, A, W, bias, static_cast<uint16_t*>(out),
                         static_cast<uint16_t*>(nullptr), M, N, K, nMt, nNt, (unsigned)out_bytes);
    TR_CHECK_LAUNCH("tr_gemm_bf16");
    return TR_OK;
  }
#define TR_LAUNCH(E)                                                                                                  \
  hipLaunchKernelGGL(gemm_bf16_persistent<E>, grid, block, 0, st, A, W, bias, out, aux, aux_i, M, N, K, nMt, nNt, \
                     (unsigned)out_bytes)
  switch (epilogue) {
    case TR_EPI_BF16: TR_LAUNCH(TR_EPI_BF16); break;
    case TR_EPI_GELU_BF16: TR_LAUNCH(TR_EPI_GELU_BF16); break;
    case TR_EPI_RESID_F32: TR_LAUNCH(TR_EPI_RESID_F32); break;
    case TR_EPI_F32: TR_LAUNCH(TR_EPI_F32); break;
    case TR_EPI_PATCH_F32: TR_LAUNCH(TR_EPI_PATCH_F32); break;
    default: TR_REQUIRE(false, TR_ERR_SHAPE, "tr_gemm_bf16: unknown epilogue %d", epilogue);
  }
#undef TR_LAUNCH
  TR_CHECK_LAUNCH("tr_gemm_bf16");
  return TR_OK;
}

// Backward of fc2 -> GELU (timm Mlp): out bf16 [M,N] = bf16(A W^T) * gelu'(pre), pre bf16 [M,N] the pre-activation the training forward
// kept -- bitwise what tr_gemm_bf16(TR_EPI_BF16, zero bias) followed by tr_gelu_bwd_bf16 produces, in one launch.
extern "C" int tr_gemm_dgelu_bf16(const uint16_t* A, const uint16_t* W, const uint16_t* pre, uint16_t* out, int M, int N, int K, tr_stream_t s) {
  TR_REQUIRE(A && W && pre && out, TR_ERR_NULL, "tr_gemm_dgelu_bf16: null pointer");
  TR_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0 && N % 8 == 0, TR_ERR_SHAPE, "tr_gemm_dgelu_bf16: need K %% %d == 0, N %% 8 == 0 (M=%d N=%d K=%d)", BK, M, N,
             K);
  TR_REQUIRE(tr_aligned16(A) && tr_aligned16(W) && tr_aligned16(pre) && tr_aligned16(out), TR_ERR_ALIGN,
             "tr_gemm_dgelu_bf16: pointers must be 16-byte aligned");
  const size_t out_bytes = (size_t)M * N * 2;
  TR_REQUIRE(out_bytes < ((size_t)1 << 31) && (size_t)M * K * 2 < ((size_t)1 << 32) && (size_t)N * K * 2 < ((size_t)1 << 32), TR_ERR_SHAPE,
             "tr_gemm_dgelu_bf16: operands / outputs beyond the 32-bit offset range");
  tr_prof_note("gemm_bf16_pc<EPI_DGELU>", 2.0 * M * N * K, 2.0 * ((double)M * K + (double)N * K) + 4.0 * M * N);
  const int nMt = (M + PBM - 1) / PBM, nNt = (N + PBN - 1) / PBN;
  hipLaunchKernelGGL(gemm_bf16_pc<EPI_DGELU>, dim3(256), dim3(PC_THREADS), 0, static_cast<hipStream_t>(s), A, W, static_cast<const float*>(nullptr), out,
                     const_cast<uint16_t*>(pre), M, N, K,
                     nMt, nNt, (unsigned)out_bytes);
  TR_CHECK_LAUNCH("tr_gemm_dgelu_bf16");
  return TR_OK;
}

// fc1 of the TRAINING forward (timm Mlp): pre bf16 [M,N] = A W^T + bias (what the backward differentiates), h bf16 [M,N] = gelu(pre)
// with the fit of TR_EPI_GELU_BF16 applied to the ROUNDED pre-activation -- bitwise what tr_gemm_bf16(TR_EPI_BF16) followed by
// tr_gelu_bf16 produces, in one launch.
extern "C" int tr_gemm_gelu_keep_bf16(const uint16_t* A, const uint16_t* W, const float* bias, uint16_t* pre, uint16_t* h, int M, int N, int K,
                                      tr_stream_t s) {
  TR_REQUIRE(A && W && bias && pre && h, TR_ERR_NULL, "tr_gemm_gelu_keep_bf16: null pointer");
  TR_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0 && N % 8 == 0, TR_ERR_SHAPE, "tr_gemm_gelu_keep_bf16: need K %% %d == 0, N %% 8 == 0 (M=%d N=%d K=%d)", BK,
             M, N, K);
  TR_REQUIRE(tr_aligned16(A) && tr_aligned16(W) && tr_aligned16(bias) && tr_aligned16(pre) && tr_aligned16(h), TR_ERR_ALIGN,
             "tr_gemm_gelu_keep_bf16: pointers must be 16-byte aligned");
  const size_t out_bytes = (size_t)M * N * 2;
  TR_REQUIRE(out_bytes < ((size_t)1 << 31) && (size_t)M * K * 2 < ((size_t)1 << 32) && (size_t)N * K * 2 < ((size_t)1 << 32), TR_ERR_SHAPE,
             "tr_gemm_gelu_keep_bf16: operands / outputs beyond the 32-bit offset range");
  tr_prof_note("gemm_bf16_pc<EPI_GELU_KEEP>", 2.0 * M * N * K, 2.0 * ((double)M * K + (double)N * K) + 4.0 * M * N);
  const int nMt = (M + PBM - 1) / PBM, nNt = (N + PBN - 1) / PBN;
  hipLaunchKernelGGL(gemm_bf16_pc<EPI_GELU_KEEP>, dim3(256), dim3(PC_THREADS), 0, static_cast<hipStream_t>(s), A, W, bias, h, pre, M, N, K, nMt, nNt,
                     (unsigned)out_bytes);
  TR_CHECK_LAUNCH("tr_gemm_gelu_keep_bf16");
  return TR_OK;
}

// lab (tools/lab/clock_probe.py): read and reset the in-kernel clock probes of a -DTR_DIAG_CLOCK build.  which 0: gemm_bf16_pc.  out[3] = {shader
// cycles, 100-MHz ticks, launches}; all zero in a product build.
extern "C" int tr_gemm_clock_probe_read(unsigned long long* out) {
  TR_REQUIRE(out, TR_ERR_NULL, "tr_gemm_clock_probe_read: null pointer");
  const unsigned long long zero[3] = {0ull, 0ull, 0ull};
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(tr_gemm_clock_probe), sizeof(zero));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(tr_gemm_clock_probe), zero, sizeof(zero));
  TR_REQUIRE(e == hipSuccess, TR_ERR_LAUNCH, "tr_gemm_clock_probe_read: %s", hipGetErrorString(e));
  return TR_OK;
}
