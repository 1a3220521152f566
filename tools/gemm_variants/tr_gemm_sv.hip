// bf16 Linear layers on MFMA for gfx950, second generation: the per-block GEMMs whose output is bf16 (qkv topk.py:44, proj topk.py:52,
// mlp.fc1 + GELU and mlp.fc2 of timm's Mlp) when N is a multiple of 192 (every DeiT width: 192, 384, 768 and their 3x / 4x).
//
// What is different from gemm_bf16_pc (tr_gemm.hip, kept for the other shapes and the training epilogues), and why (profiles/r04_gemm_lab.md):
//   * Tile (32 JT) x 192 instead of 256 x 128, JT = 4..6 chosen per launch: no column waste on N = 384 / 1152 / 1536, and a row count the
//     launcher can pick so that the tile count fills the last round of the 256 persistent workgroups.
//   * The epilogue leaves the MFMA waves.  A wave's accumulators are final slab by slab (16 token rows) inside the tile's LAST half
//     K-step, because every half K-step walks the slabs in the same order; each slab goes, rounded to bf16, into a 12-KiB LDS "mailbox"
//     and the wave's next MFMA group starts the next tile in those registers (6 conversions + 3 LDS stores + 1 flag store + 3 LDS loads
//     of the next bias per slab -- the hand-over is bound by vector-instruction ISSUE, every instruction less counts).  The four SERVICE
//     waves (one per SIMD; they also issue all LDS-DMA) pull a mailbox into registers, release it, and push it out -- GELU where asked,
//     whole 128-byte lines -- a few lines per K-step over the FOLLOWING tile's K-loop, behind that step's DMA pieces.
//   * Two operand rings instead of one: activations 3 K-steps deep, weights 3 deep where the LDS allows (JT <= 5) and 2 deep at JT = 6
//     (a one-step window costs a landing wait of a few hundred cycles per step: in-kernel stamps, profiles/r04_gemm_lab.md).
//   * The bias reaches the MFMA waves through LDS (one DMA piece per tile) and is read straight into the accumulators: the MFMA waves
//     issue no vector-memory instruction at all.
//   * Hand-over flags are single-writer monotonic counters in LDS (plain stores, polled loads): LDS instructions of one wave execute in
//     order, so a flag written after the data is seen after the data; no atomics, no barrier beyond the one per K-step.
// Numerics: the accumulation order of an output element is the old kernel's (bias, then k ascending in steps of 32), so TR_EPI_BF16 is
// bit-identical to gemm_bf16_pc.  TR_EPI_GELU_BF16 applies the GELU fit to the bf16-ROUNDED pre-activation -- what the training forward
// (tr_gemm_gelu_keep_bf16) and tr_gelu_bf16 have always done, so eval and training now agree bit for bit on fc1.
#include "../../tokenreduction_amd/csrc/tr_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int SV_BN = 192, SV_BK = 64;
constexpr int SV_W_SLOT = SV_BN * 128;     // 24 KiB: 192 weight rows x 64 bf16
constexpr int SV_MB_ROW = 2 * SV_BN;       // a mailbox row: 192 bf16
constexpr int SV_MB_BUF = 32 * SV_MB_ROW;  // 12 KiB: one 16-row slab of each of the 8 MFMA waves
constexpr int SV_BIAS_BUF = 1024;          // 192 floats, padded to one DMA piece
constexpr int SV_LDS_MAX = 163840;

template <int JT>
struct SvLds {
  static constexpr int BM = 32 * JT;
  static constexpr int A_SLOT = BM * 128;
  static constexpr int WD = JT <= 5 ? 3 : 2;            // depth of the weight ring
  static constexpr int W_RING = 3 * A_SLOT;
  static constexpr int MBOX = W_RING + WD * SV_W_SLOT;
  // as many mailboxes as fit, at most one per slab: 3 / 2 / 3 for JT 4 / 5 / 6
  static constexpr int NMB_FIT = (SV_LDS_MAX - MBOX - 2 * SV_BIAS_BUF - 256) / SV_MB_BUF;
  static constexpr int NMB = NMB_FIT < JT ? NMB_FIT : JT;
  static constexpr int BIAS = MBOX + NMB * SV_MB_BUF;
  static constexpr int FLAGS = BIAS + 2 * SV_BIAS_BUF;   // bytes: ready[b][wave] at 32 b + 4 wave (b < 4), free[b] at 128 + 16 b
  static constexpr int TOTAL = FLAGS + 256;
  // slab j of a tile goes to mailbox j % NMB; uses of mailbox b per tile
  static constexpr int uses(int b) { return (JT - b + NMB - 1) / NMB; }
};

// one 1-KiB LDS-DMA piece: 64 lanes x 16 bytes, wave-uniform LDS destination in M0, per-lane source = scalar base + 32-bit offset
__device__ __forceinline__ void sv_piece(const void* sbase, unsigned voff, unsigned lds_dst) {
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

__device__ __forceinline__ unsigned sv_flag_load(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// wait until *p >= want, given a value `have` read earlier.  TR_SV_SPIN_LIMIT (lab builds only) bounds the wait, so that a protocol error
// ends in wrong results, not in a hung GPU
__device__ __forceinline__ void sv_flag_wait(const unsigned* p, unsigned want, unsigned have) {
#ifdef TR_SV_SPIN_LIMIT
  for (int spin = 0; spin < TR_SV_SPIN_LIMIT && have < want; ++spin) {
#else
  while (have < want) {
#endif
    __builtin_amdgcn_s_sleep(1);
    have = sv_flag_load(p);
  }
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}
// Every flag word has ONE writer and only ever grows, so a plain LDS store publishes it (no atomics).  LDS instructions of one wave
// execute in order: everything the wave wrote (or read) before is done when the store is seen; the signal fences only stop the
// COMPILER from moving those accesses across it (a release would make it wait for lgkmcnt(0)).
__device__ __forceinline__ void sv_flag_set(unsigned* p, unsigned v, int lane) {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate must be a constant: one statement per value)
__device__ __forceinline__ void sv_wait_vm(int n) {
  switch (n) {
#define SV_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    SV_VM(4) SV_VM(5) SV_VM(6) SV_VM(7) SV_VM(8) SV_VM(9) SV_VM(10) SV_VM(11) SV_VM(12) SV_VM(13) SV_VM(14) SV_VM(15) SV_VM(16) SV_VM(17)
    SV_VM(18) SV_VM(19) SV_VM(20) SV_VM(21) SV_VM(22) SV_VM(23) SV_VM(24) SV_VM(25) SV_VM(26) SV_VM(27) SV_VM(28) SV_VM(29) SV_VM(30)
    SV_VM(31) SV_VM(32) SV_VM(33) SV_VM(34) SV_VM(35) SV_VM(36)
#undef SV_VM
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

#ifdef TR_SV_STAMPS
// lab only: s_memtime stamps of workgroup 8 -- [0]: MFMA wave 0, [1]: service wave 0 -- eight per K-step, the first 128 steps
__device__ unsigned long long sv_stamps[2][128][8];
#define SV_STAMP(who, step, k)                                                                                     \
  do {                                                                                                             \
    if (blockIdx.x == 8 && (step) < 128 && (lane) == 0) sv_stamps[who][step][k] = __builtin_amdgcn_s_memtime();    \
  } while (0)
#else
#define SV_STAMP(who, step, k) do { } while (0)
#endif

// EPI: TR_EPI_BF16 or TR_EPI_GELU_BF16.  JT: 16-row slabs per MFMA wave; the tile is 32 JT x 192.
template <int EPI, int JT>
__global__ __launch_bounds__(768, 3) void gemm_bf16_sv(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                       const float* __restrict__ bias, uint16_t* __restrict__ outp, int M, int N, int K, int nMt,
                                                       int nNt, unsigned out_bytes) {
  using L = SvLds<JT>;
  constexpr int BM = L::BM, NMB = L::NMB, WD = L::WD;
  static_assert(L::TOTAL <= SV_LDS_MAX && NMB >= 2 && NMB <= 4, "LDS budget");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[L::TOTAL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = K / SV_BK;
  const int G = gridDim.x, bid = blockIdx.x;
  const int T = nMt * nNt;
  // the 32 workgroups that share an XCD (blocks b, b+8, ...) take 32 consecutive tiles (n fastest) at a time: an activation row panel
  // is fetched into that XCD's L2 once
  const int toff = (bid & 7) * (G >> 3) + (bid >> 3);
  const int my_tiles = toff < T ? (T - toff + G - 1) / G : 0;
  const int S = my_tiles * nk;                       // this workgroup's K-steps, over all its tiles
  unsigned* const flags = reinterpret_cast<unsigned*>(smem + L::FLAGS);
  if (tid < 64) flags[tid] = 0u;
  __syncthreads();
  if (my_tiles == 0) return;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  if (wave >= 8) {
    // ============================================================ service wave: all LDS-DMA + the epilogue's stores
    // Barrier B_b publishes K-step b.  After B_b (a "round") this wave issues the activations of step b+2 (ring of 3) and the weights of
    // step b+2 (ring of 3) or b+1 (ring of 2), then a few lines of the slabs it pulled at the last tile boundary.  Before B_b it waits for
    // the operands of step b: everything but the DMA that may still fly (JT activation pieces, + 6 weight pieces with the deep ring) and
    // the stores of the previous round, which are younger than its DMA -- an exact count, as vmcnt retires in order.
    const int lw = wave - 8;
    const int l3 = lane >> 3, pc = lane & 7;
    unsigned oa[JT], ow[6];
    int a_unit = 0, a_kt = 0, a_step = 0, w_unit = 0, w_kt = 0, w_step = 0;
    auto set_a = [&](int u) __attribute__((always_inline)) {
      const int tile = toff + u * G;
      const int tm0 = (tile / nNt) * BM;
#pragma unroll
      for (int p = 0; p < JT; ++p) {
        const int r = lw * 8 * JT + 8 * p + l3;
        const int c = pc ^ ((r >> 1) & 7);           // LDS position (row, chunk pc) holds LOGICAL chunk pc ^ ((row>>1)&7)
        oa[p] = ((unsigned)min(tm0 + r, M - 1) * (unsigned)K + (unsigned)c * 8u) * 2u;
      }
    };
    auto set_w = [&](int u) __attribute__((always_inline)) {
      const int tile = toff + u * G;
      const int tn0 = (tile % nNt) * SV_BN;
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const int r = lw * 48 + 8 * p + l3;
        const int c = pc ^ ((r >> 1) & 7);
        ow[p] = ((unsigned)(tn0 + r) * (unsigned)K + (unsigned)c * 8u) * 2u;
      }
    };
    // once nothing is left to load the pieces become dummies (lane-invariant source) into the slot that is free anyway: every counted
    // wait keeps its value
    auto issue_a = [&](int aslot) __attribute__((always_inline)) {
      const bool real = a_step < S;
      const unsigned dst = lds0 + aslot * L::A_SLOT + lw * (8 * JT * 128);
#pragma unroll
      for (int p = 0; p < JT; ++p) sv_piece(A, real ? oa[p] : 0u, dst + p * 1024);
      if (real) {
        ++a_step;
        if (++a_kt == nk) {
          a_kt = 0;
          ++a_unit;
          if (a_step < S) set_a(a_unit);
        } else {
#pragma unroll
          for (int p = 0; p < JT; ++p) oa[p] += 2 * SV_BK;
        }
      }
    };
    auto issue_w = [&](int wslot) __attribute__((always_inline)) {
      const bool real = w_step < S;
      const unsigned dst = lds0 + L::W_RING + wslot * SV_W_SLOT + lw * (48 * 128);
#pragma unroll
      for (int p = 0; p < 6; ++p) sv_piece(real ? (const void*)W : (const void*)A, real ? ow[p] : 0u, dst + p * 1024);
      if (real) {
        ++w_step;
        if (++w_kt == nk) {
          w_kt = 0;
          ++w_unit;
          if (w_step < S) set_w(w_unit);
        } else {
#pragma unroll
          for (int p = 0; p < 6; ++p) ow[p] += 2 * SV_BK;
        }
      }
    };
    auto bias_piece = [&](int u) __attribute__((always_inline)) {
      // 192 floats of the tile's columns -> LDS (lanes 48.. re-read lane 47's 16 bytes: the piece is always 1 KiB); service wave 0 only
      const int tile = toff + u * G;
      const unsigned off = (unsigned)((tile % nNt) * SV_BN) * 4u + (unsigned)min(lane, 47) * 16u;
      sv_piece(bias, off, lds0 + L::BIAS + (u & 1) * SV_BIAS_BUF);
    };

    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)out_bytes, 0x00020000);
    // One slab = one mailbox = 32 rows x 384 bytes, served by ONE wave (slab j of tile u by wave (j + u) & 3, so that up to four
    // mailboxes are drained side by side and the work evens out over two tiles): 12 x 16 bytes per lane.
    // A lane takes row 8t + (lane >> 3), 16-byte chunk (lane & 7) + 8l of it (t = 0..3, l = 0..2); position = chunk ^ (row & 7).
    const unsigned rd_off = (unsigned)(L::MBOX + l3 * SV_MB_ROW + ((pc ^ l3) << 4));
    u32x4 ln[12];                     // the slab being pushed out
    int pend_tile = 0, pend_u = 0, pend_j = 0;   // its tile (global index, and index within this workgroup) and slab
    bool second = false;              // this wave also serves slab pend_j + 4 of that tile: pulled when the first one has left (its
                                      // mailbox is not needed again before the next tile's hand-over, and 48 registers hold one slab)
    int ph = 4, ph_round = 0;         // push phases (3 lines each, 4 per slab): next, per round
    auto pull = [&](int u, int j) __attribute__((always_inline)) {     // mailbox -> registers, mailbox released
      const int b = j % NMB;     // j is wave-uniform, not a constant here
      const unsigned k = (unsigned)(u * ((JT - b + NMB - 1) / NMB) + j / NMB);
      {   // all eight MFMA waves have written use k of this mailbox: each keeps its own counter word
        const u32x4* rp = reinterpret_cast<const u32x4*>(flags + 8 * b);
#ifdef TR_SV_SPIN_LIMIT
        for (int spin = 0; spin < TR_SV_SPIN_LIMIT; ++spin) {
#else
        for (;;) {
#endif
          const u32x4 c0 = __builtin_nontemporal_load(rp), c1 = __builtin_nontemporal_load(rp + 1);
          const unsigned lo = min(min(min(c0[0], c0[1]), min(c0[2], c0[3])), min(min(c1[0], c1[1]), min(c1[2], c1[3])));
          if (__builtin_amdgcn_readfirstlane(lo) > k) break;
          __builtin_amdgcn_s_sleep(1);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
      }
      const unsigned char* src = smem + rd_off + b * SV_MB_BUF;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int l = 0; l < 3; ++l) ln[3 * t + l] = *reinterpret_cast<const u32x4*>(src + t * 8 * SV_MB_ROW + l * 128);
      sv_flag_set(flags + 32 + 4 * b, k + 1u, lane);  // in-order LDS: the reads above have been performed when the store is
    };
    // one push phase: the three lines of row group t of the slab in registers -> (GELU) -> global memory
    auto push_phase = [&](const int t) __attribute__((always_inline)) {
      const int m0 = (pend_tile / nNt) * BM + 16 * pend_j + l3;
      // mailbox row 8t + l3 is row 8(t&1) + l3 of the slab of wave row t>>1
      const int dm = (t >> 1) * (16 * JT) + (t & 1) * 8;
      const unsigned off0 = ((unsigned)(m0 + dm) * (unsigned)N + (unsigned)((pend_tile % nNt) * SV_BN)) * 2u + (unsigned)pc * 16u;
      bool ok = m0 + dm < M;
#ifdef TR_ABLATE_NO_STORE
      ok = ok && (K == 0x7fffffff);
#endif
#pragma unroll
      for (int l = 0; l < 3; ++l) {
        u32x4 v = ln[3 * t + l];
        if (t & 1) v = u32x4{v[2], v[3], v[0], v[1]};   // rows 8..15 of a slab were written with their 8-byte halves swapped
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
        // out-of-range rows get an offset beyond num_records: the buffer bounds check drops the store, the store COUNT stays exact
        __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, ok ? off0 + (unsigned)l * 128u : 0x80000000u, 0, 0);
      }
    };
    // up to n phases of what is pending (the second slab is pulled when the first has left); returns the number pushed
    auto push_some = [&](int n) __attribute__((always_inline)) -> int {
      int done = 0;
      while (done < n) {
        if (ph == 4) {
          if (!second) break;
          second = false;
          pend_j += 4;
          pull(pend_u, pend_j);
          ph = 0;
        }
        switch (ph) {
          case 0: push_phase(0); break;
          case 1: push_phase(1); break;
          case 2: push_phase(2); break;
          default: push_phase(3); break;
        }
        ++ph;
        ++done;
      }
      return done;
    };

    set_a(0);
    set_w(0);
    if (lw == 0) bias_piece(0);
    int aslot = 2, wslot = WD == 3 ? 2 : 1;     // targets of round 0: activations of step 2, weights of step 2 (deep ring) or 1
    if (WD == 3) {
      issue_w(0); issue_a(0); issue_w(1); issue_a(1);
    } else {
      issue_w(0); issue_a(0); issue_a(1);
    }
    constexpr int FLY = WD == 3 ? JT + 6 : JT;  // DMA pieces of the previous round that may still be in flight at a barrier
    int b_kt = 0, b_unit = 0;                   // step b = b_unit * nk + b_kt
    int stores = 0;                             // stores issued in the previous round (all younger than its DMA)
    for (int b = 0; b < S; ++b) {
      sv_wait_vm(FLY + stores);
      if (lw == 0) SV_STAMP(1, b, 0);
      __builtin_amdgcn_s_barrier();          // B_b
      asm volatile("" ::: "memory");
      if (lw == 0) SV_STAMP(1, b, 1);
      const bool boundary = (b_kt == 0) && b > 0;   // the MFMA waves are in the last half step of tile b_unit - 1 and hand it over
      if (b_kt == 0 && lw == 0 && b_unit + 1 < my_tiles) bias_piece(b_unit + 1);
      int r = 0;
      stores = 0;
#ifndef TR_ABLATE_NO_EPI
      if (boundary) {
        // whatever is still pending of the tile before must leave first: its registers are about to be refilled
        stores = 3 * push_some(8);
        r = (lw - (b_unit - 1)) & 3;
      }
      // the first slabs are ready within a few hundred cycles of the barrier, the later ones are not: their waves issue the DMA first
      const bool early = boundary && r < 2;
#else
      const bool early = false;
#endif
      if (!early) {
        issue_w(wslot);
        if (lw == 0) SV_STAMP(1, b, 2);
        issue_a(aslot);
      }
#ifndef TR_ABLATE_NO_EPI
      if (boundary) pull(b_unit - 1, r);
#endif
      if (early) {
        issue_w(wslot);
        if (lw == 0) SV_STAMP(1, b, 2);
        issue_a(aslot);
      }
      if (lw == 0) SV_STAMP(1, b, 3);
#ifndef TR_ABLATE_NO_EPI
      if (boundary) {
        pend_u = b_unit - 1;
        pend_tile = toff + pend_u * G;
        pend_j = r;
        second = r + 4 < JT;
        ph = 0;
        ph_round = ((second ? 8 : 4) + nk - 1) / nk;
      }
      stores += 3 * push_some(ph_round);
#endif
      aslot = (aslot == 2) ? 0 : aslot + 1;
      wslot = (wslot == WD - 1) ? 0 : wslot + 1;
      if (++b_kt == nk) { b_kt = 0; ++b_unit; }
    }
#ifndef TR_ABLATE_NO_EPI
    {   // the last tile: no DMA left to hide behind
      push_some(8);
      pend_u = my_tiles - 1;
      pend_tile = toff + pend_u * G;
      pend_j = (lw - pend_u) & 3;
      pull(pend_u, pend_j);
      second = pend_j + 4 < JT;
      ph = 0;
      push_some(8);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy DMA into this workgroup's LDS must not outlive it
    return;
  }

  // ================================================================ MFMA wave: 16 JT rows x 48 columns of the tile
  const int wm = wave >> 2, wn = wave & 3;
  const int frow = lane & 15, fq = lane >> 4;
  // fragment read offsets inside a slot: row*128 + ((chunk ^ ((row>>1)&7)) << 4), chunk = 4 ks + fq; the wave's row bases are multiples
  // of 16, so the swizzle term is the lane's own and the two k-halves differ by XOR 64
  const unsigned lo0 = (unsigned)(frow * 128 + ((fq ^ ((frow >> 1) & 7)) << 4));
  const unsigned char* const a_rd0 = smem + wm * (16 * JT * 128) + lo0;
  const unsigned char* const a_rd1 = smem + wm * (16 * JT * 128) + (lo0 ^ 64u);
  const unsigned char* const w_rd0 = smem + L::W_RING + wn * (48 * 128) + lo0;
  const unsigned char* const w_rd1 = smem + L::W_RING + wn * (48 * 128) + (lo0 ^ 64u);
  auto rdA = [&](int slot, const int ks, int j) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8*>((ks ? a_rd1 : a_rd0) + slot * L::A_SLOT + j * 2048);
  };
  auto rdW = [&](int slot, const int ks, int i) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8*>((ks ? w_rd1 : w_rd0) + slot * SV_W_SLOT + i * 2048);
  };

  f32x4 acc[3][JT];
  bf16x8 A0[JT], A1[JT], W0[3], W1[3];      // fragments of the two k-halves of a step
#ifdef TR_ABLATE_NO_MFMA
#define SV_MFMA(WF, AFRAG, C) ({ asm volatile("" ::"v"(WF), "v"(AFRAG)); (C); })
#else
#define SV_MFMA(WF, AFRAG, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16((WF), (AFRAG), (C), 0, 0, 0)
#endif

  int u = 0, gs = 0;
  int aslot = 0, wslot = 0;
  // the accumulators of slab j START at the bias of the tile's columns, read straight into them from the LDS copy the service waves keep
  const unsigned char* const bias_rd = smem + L::BIAS + (wn * 48 + fq * 4) * 4;
  auto init_slab = [&](const unsigned char* bz, int j) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i][j] = *reinterpret_cast<const f32x4*>(bz + i * 64);
  };

  // One K-step = two half steps of 3 x JT MFMAs, slab by slab; each half step first requests the fragments of the NEXT half step.
  //   first half : k 0..31 (W0, A0); requests k 32..63 of the same slots
  //   barrier B_{gs+1}: step gs+1 has landed, and this wave is done reading step gs
  //   second half: k 32..63 (W1, A1); requests k 0..31 of the NEXT step's slots
  //   LAST: the tile's accumulators leave slab by slab during the second half (slab j-1 under the MFMAs of slab j).
  auto k_step = [&](const bool LAST) __attribute__((always_inline)) {
    const int an = (aslot == 2) ? 0 : aslot + 1, wnx = (wslot == WD - 1) ? 0 : wslot + 1;
    if (wave == 0) SV_STAMP(0, gs, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i) W1[i] = rdW(wslot, 1, i);
#pragma unroll
    for (int j = 0; j < JT; ++j) A1[j] = rdA(aslot, 1, j);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < JT; ++j) {
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i][j] = SV_MFMA(W0[i], A0[j], acc[i][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (wave == 0) SV_STAMP(0, gs, 1);
    if (!LAST || gs + 1 < S) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own reads of step gs are done: its slots may be refilled after B_{gs+1}
      __builtin_amdgcn_s_barrier();                        // B_{gs+1}: step gs+1 has landed
      asm volatile("" ::: "memory");
    }
    if (wave == 0) SV_STAMP(0, gs, 2);
    if (!LAST) {
#pragma unroll
      for (int i = 0; i < 3; ++i) W0[i] = rdW(wnx, 0, i);
#pragma unroll
      for (int j = 0; j < JT; ++j) A0[j] = rdA(an, 0, j);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < JT; ++j) {
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i][j] = SV_MFMA(W1[i], A1[j], acc[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // ---- the tile's last half step: slab j-1 is handed over under the MFMAs of slab j.  Per slab and wave: 6 conversions, 3 LDS
      // stores, 1 flag store, 3 LDS loads (the next tile's bias, straight into the accumulators) -- the three 8-byte store addresses
      // and the bias address are computed once per tile, from an opaque copy of the lane id so that they are NOT hoisted out of the
      // K-loop (three registers the loop does not have).  The next tile's fragments are requested slab by slab BEHIND the MFMAs that
      // free their registers: a once-per-tile read latency is cheaper than a spill.
      // mailbox write: row wm*16 + frow, 8 bytes at logical 16-byte chunk wn*6 + 2i + (fq>>1), half fq&1; position = chunk ^ (row & 7),
      // and rows 8..15 take the OTHER 8-byte half (16 lanes of a ds_write_b64 then cover all 32 banks)
      unsigned l_ = (unsigned)lane;
      asm volatile("" : "+v"(l_));
      const unsigned fr_ = l_ & 15u, fq_ = l_ >> 4;
      unsigned char* mw[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
        mw[i] = smem + L::MBOX + (wm * 16 + fr_) * SV_MB_ROW + ((((unsigned)(wn * 6 + 2 * i) + (fq_ >> 1)) ^ (fr_ & 7u)) << 4) + (((fq_ & 1u) ^ (fr_ >> 3)) << 3);
      const unsigned char* const bz = smem + L::BIAS + ((u + 1) & 1) * SV_BIAS_BUF + (wn * 48 + fq_ * 4) * 4;
      unsigned* const my_ready = flags + wave;
      auto dump = [&](const int j, unsigned have) __attribute__((always_inline)) {
        constexpr int b = 0; (void)b;
        const int mb = j % NMB;
        const unsigned k = (unsigned)(u * L::uses(mb) + j / NMB);
#ifndef TR_ABLATE_NO_EPI
        // every earlier use of this mailbox must have been pulled: the tile before's (its second slabs are pulled a few rounds late) and,
        // for a re-use within the tile, slab j - NMB's
        sv_flag_wait(flags + 32 + 4 * mb, k, have);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          u32x2 pk;
          pk[0] = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
          pk[1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
          *reinterpret_cast<u32x2*>(mw[i] + mb * SV_MB_BUF) = pk;
        }
        sv_flag_set(my_ready + 8 * mb, k + 1u, lane);
#else
#pragma unroll
        for (int i = 0; i < 3; ++i) asm volatile("" ::"v"(acc[i][j]));
        (void)have; (void)k;
#endif
        init_slab(bz, j);
      };
      unsigned fr[NMB];
#pragma unroll
      for (int b = 0; b < NMB; ++b) fr[b] = sv_flag_load(flags + 32 + 4 * b);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        // a mailbox used a second time within the tile: look at its release counter one MFMA group ahead of the hand-over
        if (j >= 1 && j - 1 >= NMB) fr[(j - 1) % NMB] = sv_flag_load(flags + 32 + 4 * ((j - 1) % NMB));
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i][j] = SV_MFMA(W1[i], A1[j], acc[i][j]);
        A0[j] = rdA(an, 0, j);
        __builtin_amdgcn_sched_barrier(0);
        if (j >= 1) {
          dump(j - 1, fr[(j - 1) % NMB]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) W0[i] = rdW(wnx, 0, i);
      if (JT - 1 >= NMB) fr[(JT - 1) % NMB] = sv_flag_load(flags + 32 + 4 * ((JT - 1) % NMB));
      dump(JT - 1, fr[(JT - 1) % NMB]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wave == 0) SV_STAMP(0, gs, 3);
    aslot = an;
    wslot = wnx;
    ++gs;
  };

  __builtin_amdgcn_s_barrier();            // B_0
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 3; ++i) W0[i] = rdW(0, 0, i);
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    A0[j] = rdA(0, 0, j);
    init_slab(bias_rd, j);
  }
  for (u = 0; u < my_tiles; ++u) {
    for (int kt = 0; kt < nk - 1; ++kt) k_step(false);
    k_step(true);
  }
#undef SV_MFMA
}

template <int EPI, int JT>
int sv_launch(const uint16_t* A, const uint16_t* W, const float* bias, uint16_t* out, int M, int N, int K, unsigned out_bytes, hipStream_t st) {
  const int nMt = (M + 32 * JT - 1) / (32 * JT), nNt = N / SV_BN;
  hipLaunchKernelGGL((gemm_bf16_sv<EPI, JT>), dim3(256), dim3(768), 0, st, A, W, bias, out, M, N, K, nMt, nNt, out_bytes);
  return 0;
}

}  // namespace

// Rows per tile = 32 jt that finish the launch soonest on 256 persistent workgroups.  A tile's K-step costs what its LDS-DMA pieces cost
// (4 jt of activations + 24 of weights; the matrix work, 192 jt cycles per SIMD, stays below that for every jt <= 7) plus a fixed part
// for the step's barrier; the workgroup with the most tiles sets the time.  Calibrated on the DeiT-S / DeiT-B shapes (profiles/r04_gemm_lab.md).
extern "C" int tr_gemm_sv_pick_jt(int M, int N, int K) {
  (void)K;
  const int nNt = N / SV_BN;
  int best = 6;
  double best_cost = 1e300;
  for (int jt = 6; jt >= 4; --jt) {
    const long tiles = (long)((M + 32 * jt - 1) / (32 * jt)) * nNt;
    const long rounds = (tiles + 255) / 256;
    const double cost = (double)rounds * (4.0 * jt + 24.0 + 6.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = jt; }
  }
  return best;
}

// tr_gemm_bf16 for TR_EPI_BF16 / TR_EPI_GELU_BF16 with an explicit tile height (jt = 4..7, 0 = tr_gemm_sv_pick_jt): the entry point
// the dispatcher in tr_gemm_bf16 uses, exported so that tests and the lab can force every instantiation.
extern "C" int tr_gemm_bf16_sv(const uint16_t* A, const uint16_t* W, const float* bias, uint16_t* out, int M, int N, int K, int epilogue, int jt,
                               tr_stream_t s) {
  TR_REQUIRE(A && W && bias && out, TR_ERR_NULL, "tr_gemm_bf16_sv: null pointer");
  TR_REQUIRE(epilogue == TR_EPI_BF16 || epilogue == TR_EPI_GELU_BF16, TR_ERR_SHAPE, "tr_gemm_bf16_sv: bf16 epilogues only (got %d)", epilogue);
  TR_REQUIRE(M > 0 && N > 0 && K > 0 && K % SV_BK == 0 && N % SV_BN == 0, TR_ERR_SHAPE,
             "tr_gemm_bf16_sv: need K %% 64 == 0 and N %% 192 == 0 (M=%d N=%d K=%d)", M, N, K);
  TR_REQUIRE(jt == 0 || (jt >= 4 && jt <= 6), TR_ERR_SHAPE, "tr_gemm_bf16_sv: jt must be 0 or 4..6 (got %d)", jt);
  TR_REQUIRE(tr_aligned16(A) && tr_aligned16(W) && tr_aligned16(bias) && (reinterpret_cast<uintptr_t>(out) & 127u) == 0, TR_ERR_ALIGN,
             "tr_gemm_bf16_sv: operands must be 16-byte aligned, the output 128-byte aligned");
  const size_t out_bytes = (size_t)M * N * 2;
  TR_REQUIRE(out_bytes < ((size_t)1 << 31) && (size_t)M * K * 2 < ((size_t)1 << 32) && (size_t)N * K * 2 < ((size_t)1 << 32), TR_ERR_SHAPE,
             "tr_gemm_bf16_sv: operands / outputs beyond the 32-bit offset range");
  if (jt == 0) jt = tr_gemm_sv_pick_jt(M, N, K);
  tr_prof_note(epilogue == TR_EPI_BF16 ? "gemm_bf16_sv<EPI_BF16>" : "gemm_bf16_sv<EPI_GELU_BF16>", 2.0 * M * N * K,
               2.0 * ((double)M * K + (double)N * K) + 2.0 * M * N);
  hipStream_t st = static_cast<hipStream_t>(s);
  uint16_t* o = out;
#define SV_CASE(J)                                                                                                   \
  case J:                                                                                                            \
    if (epilogue == TR_EPI_BF16) sv_launch<TR_EPI_BF16, J>(A, W, bias, o, M, N, K, (unsigned)out_bytes, st);         \
    else sv_launch<TR_EPI_GELU_BF16, J>(A, W, bias, o, M, N, K, (unsigned)out_bytes, st);                            \
    break
  switch (jt) {
    SV_CASE(4);
    SV_CASE(5);
    SV_CASE(6);
  }
#undef SV_CASE
  TR_CHECK_LAUNCH("tr_gemm_bf16_sv");
  return TR_OK;
}
