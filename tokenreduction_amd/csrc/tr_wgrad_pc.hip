// Weight gradient dW[N,K] = dY[M,N]^T X[M,K] (nn.Linear, engine.py:76 loss.backward()) as a producer/consumer kernel: the LDS-DMA ring
// and barrier protocol of gemm_bf16_pc (tr_gemm.hip) with the token dimension as the reduction.
//
// Why a second weight-gradient kernel: wgrad_kernel (tr_backward.hip, 128 x 128 tile, 4 waves that load, stage and multiply) moves
// (128 + 128) x 64 x 2 B per 2 x 128 x 128 x 64 FLOP -- 64 FLOP per byte fed through the CU's 64 B/clk vector-memory path -- and its waves
// stall together at every slab barrier: 390-520 TFLOP/s on the DeiT-S shapes, 24 % of a training step.  Here:
//   * tile 192 x 192 (96 FLOP per byte fed; 192 divides every DeiT width: 192 / 384 / 768 and their x3, x4 multiples),
//   * four dedicated loader waves copy global -> LDS with global_load_lds_dwordx4 (no registers, no VALU), three 48-KiB stages in flight,
//   * eight MFMA waves (4 x 2, 48 x 96 outputs each) read both operands TRANSPOSED out of row-major [token][column] images with
//     ds_read_b64_tr_b16 and never touch the vector-memory pipe inside the loop,
//   * one persistent workgroup per CU walks its (tile, token-range) units; the fp32 partial of a unit is stored once per unit.
// A stage holds 64 tokens: six images [64 tokens][64 columns] bf16 (128-byte rows) -- three of dY (the tile's 192 dW rows), three of X.
// Image swizzle: 16-byte chunk ^ 2 * (bit1(row) | bit3(row) << 1): the 16 rows one transposed read touches ({0..3, 8..11} + 16 h per half
// wave) land on all 64 banks exactly once.  The bias gradient (column sums of dY) comes from the matrix pipe as well: one extra MFMA
// per dY fragment against an all-ones operand, in the units of the first column tile only.
// Rows past M (a ragged last slab) must contribute ZERO: their dY lanes fetch a zero line instead, their X lanes the last valid row.
#include "tr_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int QT = 192;                    // tile edge (dW rows and columns)
constexpr int QM = 64;                     // tokens per stage
constexpr int Q_IMG = QM * 128;            // one image: 8 KiB
constexpr int Q_STAGE = 6 * Q_IMG;         // 48 KiB
constexpr int Q_NSTAGE = 3;

__device__ __attribute__((aligned(64))) unsigned int g_zero_line[16];      // zero-initialised: the source of every out-of-range dY row

__device__ __forceinline__ int qswz(int row) { return 2 * (((row >> 1) & 1) | (((row >> 3) & 1) << 1)); }

__device__ __forceinline__ bf16x8 lds_tr_pair(const unsigned char* p0) {      // rows r and r + 4 (same swizzle: +512 bytes)
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 512));
  const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, c);
}

__device__ __forceinline__ void dma_piece(const unsigned char* src, unsigned lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %[ld]\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %[a], off"
      :
      : [a] "v"(src), [ld] "s"(lds_dst)
      : "memory", "m0");
}

// One launch serves up to FOUR Linear layers (the executor hands over fc2, fc1, proj and qkv of a block together): their units share the
// 256 workgroups, so each matrix is cut into a quarter as many token ranges -- every launch writes (and the reduce re-reads)
// 256 x 192 x 192 fp32 partials = 38 MB whatever the matrices are, and that traffic is now paid once per block instead of four times.
struct QProblem {
  const uint16_t* Y;      // dY [M, ldy]
  const uint16_t* X;      // X  [M, ldx]
  float* part;            // [S][N][K]
  float* bpart;           // [S][N] or nullptr
  long ldy, ldx;
  int M, N, K, nKt, tiles, S, sps, u0;      // u0: first unit of this problem; units u0 + split * tiles + tile (split-major: the workgroups of
                                            // one round share their dY / X slabs in L2), tile = nt * nKt + kt
  int yskip;                                // > 0: dY holds one extra leading row per `yskip` rows (the CLS row of [B, P + 1, D] when the
};                                          // layer saw only the P patch rows: PatchEmbed): row m of the product is dY row m + m / yskip + 1
struct QGroup {
  QProblem p[4];          // unused entries: u0 = INT_MAX
  int n, U;
};
// field-wise selects: indexing the kernel argument dynamically would copy it to scratch
// (a conditional expression on two struct fields is an LVALUE in C++: the compiler selected between addresses and, to be able to, kept a
// 360-byte private copy of the arguments in scratch.  qsel4 takes the four candidates BY VALUE -- four static scalar loads and selects.)
template <typename T>
__device__ __forceinline__ T qsel4(int pi, T a, T b, T c, T d) {
  return pi >= 2 ? (pi == 3 ? d : c) : (pi ? b : a);
}
#define QSEL(pi, f) qsel4(pi, q0.f, q1.f, q2.f, q3.f)
#define QPROB(uu) (((uu) >= q1.u0 ? 1 : 0) + ((uu) >= q2.u0 ? 1 : 0) + ((uu) >= q3.u0 ? 1 : 0))

template <bool BIAS>
__global__ __launch_bounds__(768, 3) void wgrad_pc_kernel(const QProblem q0, const QProblem q1, const QProblem q2, const QProblem q3, const int U) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[Q_NSTAGE * Q_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x, bid = blockIdx.x;
  const int toff = (bid & 7) * (G >> 3) + (bid >> 3);
  const int units = toff < U ? (U - toff + G - 1) / G : 0;
  if (units == 0) return;
  auto unit_slabs = [&](int u) __attribute__((always_inline)) {
    const int uu = toff + u * G;
    const int pi = __builtin_amdgcn_readfirstlane(QPROB(uu));
    const int split = (uu - QSEL(pi, u0)) / QSEL(pi, tiles);
    const int sps = QSEL(pi, sps);
    return min(sps, (QSEL(pi, M) + QM - 1) / QM - split * sps);
  };
  int total = 0;
  for (int u = 0; u < units; ++u) total += unit_slabs(u);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  if (wave >= 8) {
    // ================================ loader wave: rows 16 lw .. 16 lw + 15 of each of the six images = 12 one-KiB pieces per stage
    const int lw = wave - 8;
    const int prow = lane >> 3, pc = lane & 7;
    int l_unit = 0, l_kt = 0, l_nk = unit_slabs(0), l_slot = 0, l_step = 0;
    int n0 = 0, k0 = 0, slab0 = 0, M = 0;
    const uint16_t* Y = nullptr;
    const uint16_t* X = nullptr;
    long ldy = 0, ldx = 0;
    int yskip = 0;
    auto set_unit = [&](int u) __attribute__((always_inline)) {
      const int uu = toff + u * G;
      const int pi = __builtin_amdgcn_readfirstlane(QPROB(uu));
      const int tiles = QSEL(pi, tiles), nKt = QSEL(pi, nKt);
      const int local = uu - QSEL(pi, u0);
      const int split = local / tiles, tile = local - split * tiles;
      n0 = (tile / nKt) * QT;
      k0 = (tile % nKt) * QT;
      slab0 = split * QSEL(pi, sps);
      M = QSEL(pi, M);
      Y = QSEL(pi, Y);
      X = QSEL(pi, X);
      ldy = QSEL(pi, ldy);
      ldx = QSEL(pi, ldx);
      yskip = QSEL(pi, yskip);
    };
    const unsigned char* zline = reinterpret_cast<const unsigned char*>(g_zero_line);
    auto issue_group = [&]() __attribute__((always_inline)) {
      const bool real = l_step < total;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + l_slot * Q_STAGE + lw * 2048);
      const int tok0 = (slab0 + l_kt) * QM + lw * 16 + prow;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int tok = tok0 + 8 * jj;
        const int lch = pc ^ (2 * (((prow >> 1) & 1) | (jj << 1)));      // bit1(row) = bit1(prow), bit3(row) = jj  (16 lw is a multiple of 16)
        const bool in = real && tok < M;
        const size_t ty = (size_t)min(tok, M - 1);
        const size_t tyy = yskip > 0 ? ty + ty / (size_t)yskip + 1 : ty;
        const unsigned char* ys = reinterpret_cast<const unsigned char*>(Y + tyy * (size_t)ldy + n0 + lch * 8);
        const unsigned char* xs = reinterpret_cast<const unsigned char*>(X + ty * (size_t)ldx + k0 + lch * 8);
#pragma unroll
        for (int h = 0; h < 3; ++h) dma_piece(in ? ys + 128 * h : zline, dst + h * Q_IMG + jj * 1024);
#pragma unroll
        for (int h = 0; h < 3; ++h) dma_piece(real ? xs + 128 * h : zline, dst + (3 + h) * Q_IMG + jj * 1024);
      }
      if (real) {
        ++l_step;
        l_slot = (l_slot == Q_NSTAGE - 1) ? 0 : l_slot + 1;
        if (++l_kt == l_nk) {
          l_kt = 0;
          ++l_unit;
          if (l_step < total) {
            set_unit(l_unit);
            l_nk = unit_slabs(l_unit);
          }
        }
      }
    };
    set_unit(0);
    issue_group();
    issue_group();
    for (int g = 0; g < total; ++g) {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // group g landed; group g + 1 (real or dummy, always 12 pieces) may still fly
      __builtin_amdgcn_s_barrier();                          // B_g: the MFMA waves are done reading slot (g - 1) % 3
      if (g + 1 < total) issue_group();                      // group g + 2 -> that slot
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ================================ MFMA wave (wm, wn): dW rows wm * 48 .. + 47, columns wn * 96 .. + 95 of the tile
  const int wm = wave >> 1, wn = wave & 1;
  const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  // byte offset inside a stage of this lane's first transposed read of fragment (image, column c of the image), token block ks:
  // row r0 = 32 ks + 8 g4 + q (bit 2 clear: rows r0 and r0 + 4 share the swizzle), 16-byte chunk c / 8 + (p >> 1), half p & 1:
  //   image * 8 KiB + 128 r0 + 16 ((c / 8 + (p >> 1)) ^ sw) + 8 (p & 1)  =  [image * 8 KiB] + lanebase + ([2 c] ^ sw16)     (c / 8 and sw even)
  // -- the bracketed terms are wave-uniform (scalar registers); one xor + add per read instead of nine offset registers (the bias variant
  // spilled with them).
  const unsigned lanebase = (unsigned)(128 * (8 * g4 + q) + 16 * (p >> 1) + 8 * (p & 1));
  const unsigned sw16 = (unsigned)(16 * qswz(8 * g4 + q));
  unsigned imgA[3], cbA[3], imgB[6], cbB[6];        // wave-uniform
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = wm * 48 + i * 16;
    imgA[i] = (unsigned)((c >> 6) * Q_IMG);
    cbA[i] = (unsigned)(2 * (c & 63));
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int c = wn * 96 + j * 16;
    imgB[j] = (unsigned)((3 + (c >> 6)) * Q_IMG);
    cbB[j] = (unsigned)(2 * (c & 63));
  }
  f32x4 acc[3][6], accb[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  int c_unit = 0, c_kt = 0, c_nk = unit_slabs(0), c_slot = 0, gs = 0;
  bf16x8 a0[3], b0[6], a1[3], b1[6];
#define Q_READ(AF, BF, slot, ks)                                                                   \
  do {                                                                                             \
    unsigned sw_ = sw16;                                                                           \
    asm volatile("" : "+v"(sw_));      /* keeps the per-fragment offsets out of the registers */    \
    const unsigned char* st_ = smem + (slot) * Q_STAGE + (ks) * 4096 + lanebase;                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) AF[i_] = lds_tr_pair(st_ + imgA[i_] + (cbA[i_] ^ sw_)); \
    _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_) BF[j_] = lds_tr_pair(st_ + imgB[j_] + (cbB[j_] ^ sw_)); \
  } while (0)
#define Q_MFMA(AF, BF)                                                                             \
  _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_)                                                 \
      _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_)                                             \
          acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AF[i_], BF[j_], acc[i_][j_], 0, 0, 0)

  __builtin_amdgcn_s_barrier();            // B_0
  asm volatile("" ::: "memory");
  Q_READ(a0, b0, 0, 0);
  // the current unit, decoded once per unit (two integer divisions and a dozen selects: per K-step they cost a quarter of the kernel)
  int pi = 0, split = 0, tn = 0, tk = 0;
  bool do_bias = false;
  auto decode_unit = [&](int u) __attribute__((always_inline)) {
    const int uu = toff + u * G;
    pi = __builtin_amdgcn_readfirstlane(QPROB(uu));
    const int tiles = QSEL(pi, tiles), nKt = QSEL(pi, nKt);
    const int local = uu - QSEL(pi, u0);
    split = local / tiles;
    const int tile = local - split * tiles;
    tn = tile / nKt;
    tk = tile - tn * nKt;
    do_bias = BIAS && wn == 0 && tk == 0 && QSEL(pi, bpart) != nullptr;
  };
  decode_unit(0);
  while (gs < total) {
    Q_READ(a1, b1, c_slot, 1);
    Q_MFMA(a0, b0);
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 3; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], ones, accb[i], 0, 0, 0);
    }
    const int next_slot = (c_slot == Q_NSTAGE - 1) ? 0 : c_slot + 1;
    if (gs + 1 < total) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own reads of this slot are done: the loaders may refill it after the next barrier
      __builtin_amdgcn_s_barrier();                        // B_{g+1}: group g + 1 has landed
      asm volatile("" ::: "memory");
    }
    Q_READ(a0, b0, next_slot, 0);          // unconditional: after the last step it reads a stale slot, unused
    __builtin_amdgcn_sched_barrier(0);
    Q_MFMA(a1, b1);
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 3; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], ones, accb[i], 0, 0, 0);
    }
    c_slot = next_slot;
    ++gs;
    if (++c_kt < c_nk) continue;
    // ---- the unit's partial: part[split][n][k]; accumulator element r of tile (i, j) = dW row 4 g4 + r, column lane & 15
    {
      const int n0 = tn * QT + wm * 48, k0 = tk * QT + wn * 96;
      const int N = QSEL(pi, N), K = QSEL(pi, K);
      float* po = QSEL(pi, part) + (size_t)split * N * K;
      float* bpart = QSEL(pi, bpart);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int k = k0 + j * 16 + (lane & 15);
#pragma unroll
          for (int r = 0; r < 4; ++r) po[(size_t)(n0 + i * 16 + 4 * g4 + r) * K + k] = acc[i][j][r];
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      if (do_bias) {
        if ((lane & 15) == 0) {
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) bpart[(size_t)split * N + n0 + i * 16 + 4 * g4 + r] = accb[i][r];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    c_kt = 0;
    ++c_unit;
    if (gs < total) {
      c_nk = unit_slabs(c_unit);
      decode_unit(c_unit);
    }
  }
#undef Q_READ
#undef Q_MFMA
}

#undef QSEL
#undef QPROB

// Token ranges per problem for a group: every unit walks at most `sps` slabs; pick the sps whose modelled time is least --
// rounds of 256 units x (sps slabs at ~900 TFLOP/s per chip + fill and partial store) + partial traffic (written, then re-read by the reduce).
void plan_group(QGroup* g) {
  int max_slab = 0;
  for (int i = 0; i < g->n; ++i) max_slab = max(max_slab, (g->p[i].M + QM - 1) / QM);
  double best_t = 1e30;
  int best = max_slab;
  for (int sps = 1; sps <= max_slab; ++sps) {
    long U = 0;
    double traffic = 0.0;
    for (int i = 0; i < g->n; ++i) {
      const int nslab = (g->p[i].M + QM - 1) / QM;
      const int S = (nslab + sps - 1) / sps;
      U += (long)g->p[i].tiles * S;
      traffic += 8.0 * S * g->p[i].N * g->p[i].K;
    }
    if (U > 2048) continue;
    const double rounds = (double)((U + 255) / 256);
    const double t = rounds * (sps * 1.34e-6 + 3.0e-6) + traffic / 4e12;
    if (t < best_t) { best_t = t; best = sps; }
  }
  int u0 = 0;
  for (int i = 0; i < g->n; ++i) {
    QProblem& q = g->p[i];
    const int nslab = (q.M + QM - 1) / QM;
    const int sps = min(best, nslab);
    q.S = (nslab + sps - 1) / sps;
    q.sps = (nslab + q.S - 1) / q.S;            // even ranges; every split owns at least one slab
    q.S = (nslab + q.sps - 1) / q.sps;
    q.u0 = u0;
    u0 += q.tiles * q.S;
  }
  g->U = u0;
}

void fill_problem(QProblem* q, const uint16_t* dY, long ldy, const uint16_t* X, long ldx, int M, int N, int K) {
  q->Y = dY; q->X = X; q->part = nullptr; q->bpart = nullptr; q->ldy = ldy; q->ldx = ldx;
  q->M = M; q->N = N; q->K = K; q->nKt = K / QT; q->tiles = (N / QT) * (K / QT); q->S = q->sps = q->u0 = 0;
  q->yskip = 0;
}

}  // namespace

// Shapes the producer/consumer kernel takes (everything else stays on wgrad_kernel): both dimensions multiples of 192.
bool tr_wgrad_pc_fits(int M, int N, int K, long ldy, long ldx, int yskip) {
  return yskip >= 0 && N % QT == 0 && K % QT == 0 && ldy % 8 == 0 && ldx % 8 == 0 && M >= QM;
}

// Split count a single problem would get (workspace sizing: part[S][N][K] + bpart[S][N]).
int tr_wgrad_pc_splits(int M, int N, int K) {
  QGroup g;
  g.n = 1;
  fill_problem(&g.p[0], nullptr, N, nullptr, K, M, N, K);
  plan_group(&g);
  return g.p[0].S;
}

// Split counts of n <= 4 problems launched together; mnk[i] = {M, N, K}.
void tr_wgrad_pc_group_splits(const int (*mnk)[3], int n, int* S) {
  QGroup g;
  g.n = n;
  for (int i = 0; i < n; ++i) fill_problem(&g.p[i], nullptr, mnk[i][1], nullptr, mnk[i][2], mnk[i][0], mnk[i][1], mnk[i][2]);
  plan_group(&g);
  for (int i = 0; i < n; ++i) S[i] = g.p[i].S;
}

static void pad_group(QGroup* g) {
  for (int i = g->n; i < 4; ++i) {
    g->p[i] = g->p[0];
    g->p[i].u0 = 0x7fffffff;
  }
}

// Launch of one problem; returns the number of splits written (part[S][N][K], bpart[S][N] when bpart != nullptr).  S_max: what the
// workspace holds.
int tr_wgrad_pc_launch(const uint16_t* dY, long ldy, int yskip, const uint16_t* X, long ldx, float* part, float* bpart, int M, int N, int K,
                       int S_max, hipStream_t st) {
  QGroup g;
  g.n = 1;
  fill_problem(&g.p[0], dY, ldy, X, ldx, M, N, K);
  g.p[0].yskip = yskip;
  plan_group(&g);
  QProblem& q = g.p[0];
  if (q.S > S_max) {
    const int nslab = (M + QM - 1) / QM;
    q.sps = (nslab + S_max - 1) / S_max;
    q.S = (nslab + q.sps - 1) / q.sps;
    g.U = q.tiles * q.S;
  }
  q.part = part;
  q.bpart = bpart;
  pad_group(&g);
  if (bpart != nullptr) hipLaunchKernelGGL(wgrad_pc_kernel<true>, dim3(256), dim3(768), 0, st, g.p[0], g.p[1], g.p[2], g.p[3], g.U);
  else hipLaunchKernelGGL(wgrad_pc_kernel<false>, dim3(256), dim3(768), 0, st, g.p[0], g.p[1], g.p[2], g.p[3], g.U);
  return q.S;
}

// Launch of n <= 4 problems (weight + bias partials of all); ws must hold sum_i S_i (N_i K_i + N_i) floats for the split counts of
// tr_wgrad_pc_group_splits.  Layout: part_0 .. part_{n-1}, bpart_0 .. bpart_{n-1}.
// direct_w / direct_b (nullable): the final destinations.  When EVERY problem runs as one token range (S = 1: the late, short stages) the
// "partials" are the results: they are stored straight to the destinations and the function returns true -- no reduce launch, no 2 x 28 MB
// round trip per DeiT-B block.  Only for overwriting callers (a partial store does not add).
bool tr_wgrad_pc_group_launch(const uint16_t* const* dY, const long* ldy, const uint16_t* const* X, const long* ldx, const int (*mnk)[3], int n,
                              float* ws, int* S, float** part, float** bpart, float* const* direct_w, float* const* direct_b, hipStream_t st) {
  QGroup g;
  g.n = n;
  for (int i = 0; i < n; ++i) fill_problem(&g.p[i], dY[i], ldy[i], X[i], ldx[i], mnk[i][0], mnk[i][1], mnk[i][2]);
  plan_group(&g);
  bool direct = direct_w != nullptr && direct_b != nullptr;
  for (int i = 0; i < n; ++i) direct = direct && g.p[i].S == 1;
  float* at = ws;
  for (int i = 0; i < n; ++i) {
    g.p[i].part = direct ? direct_w[i] : at;
    at += (size_t)g.p[i].S * g.p[i].N * g.p[i].K;
  }
  for (int i = 0; i < n; ++i) {
    g.p[i].bpart = direct ? direct_b[i] : at;
    at += (size_t)g.p[i].S * g.p[i].N;
  }
  for (int i = 0; i < n; ++i) {
    S[i] = g.p[i].S;
    part[i] = g.p[i].part;
    bpart[i] = g.p[i].bpart;
  }
  pad_group(&g);
  hipLaunchKernelGGL(wgrad_pc_kernel<true>, dim3(256), dim3(768), 0, st, g.p[0], g.p[1], g.p[2], g.p[3], g.U);
  return direct;
}
