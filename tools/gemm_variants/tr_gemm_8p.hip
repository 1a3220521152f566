// Round-4 lab, third GEMM attempt: the cdna guide's "256 x 256, 8-phase" structure (cdna_hip_programming.md, "The 256^2 8-phase template")
// rebuilt from its description for this repo's operand layout (A [M,K], W [N,K], both K-contiguous; out bf16 [M,N] = A W^T + bias).
// NOT part of the library: included by tools/gemm_8p_lab.cpp after tr_gemm.hip, timed and checked against gemm_bf16_pc there.
//
// Structure.  One workgroup of EIGHT waves per 256 x 256 tile, no loader waves: wave (wr, wc) = (w >> 2, w & 3) owns 128 rows x 64 columns
// (acc[8][4] = 128 registers).  BK = 64; the LDS holds two K-tiles (2 x 64 KB), each as four 16-KB half-tiles.  A K-tile is four PHASES of
// 16 MFMAs (one quadrant of the wave's output x K = 64); each phase = {ds_reads of the registers the coming quadrants need; one half-tile
// of LDS-DMA prefetch (2 pieces per wave); barrier; MFMAs; barrier}.  The two wave groups (wr = 0 / 1) run ONE BARRIER apart, so while
// one issues its 16 MFMAs the other reads its fragments and issues its DMA: the matrix pipe and the LDS/address pipes alternate
// between the two waves of every SIMD.  vmcnt is counted (6 = three half-tiles in flight), never 0 inside the loop.
//
// What makes a restage safe one phase behind the reads is the CUT of a K-tile into half-tiles by the phase that reads them, not by rows:
//   B-early = the W rows every wave reads in phase 1 (its output columns' first two 16-column fragments), A-early = the activation rows
//   every wave reads in phase 1 (its first four 16-row fragments), B-late (phase 2), A-late (phase 3); phase 4 reads nothing.
// Reads of tile E (even buffer): phases 1-3; restaged (for tile E + 2) in phases 2, 3, 4, 5.  Tile O (odd buffer): read in 5-7, restaged in
// 6, 7, 8 and phase 1 of the next iteration.  Phase 4's vmcnt(6) retires everything issued up to phase 1 = all of O before phase 5 reads
// it; phase 8's retires E' before the next phase 1.
//
// Operand roles: the W fragment is the MFMA A operand, so a lane's accumulator registers run along N; the W rows are dealt to the
// MFMA rows so that lane (fq, j) of fragment F holds column 16 fq + 4 F + j: sixteen consecutive columns, one 32-byte store per row.
namespace e8 {

constexpr int BM = 256, BN = 256;
constexpr int HALF = 16384, BUF = 65536;          // bytes; halves of a buffer: 0 B-early, 1 A-early, 2 B-late, 3 A-late

__device__ __forceinline__ void piece(const uint16_t* sbase, unsigned voff, unsigned lds_dst) { issue_piece(sbase, voff, lds_dst); }

template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm_bf16_8p(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, const float* __restrict__ bias,
                                                       uint16_t* __restrict__ out, int M, int N, int K, int ntiles, int nNt) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int nk = K >> 6;                                   // K-tiles (even: K % 128 == 0)
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read addresses inside a half-tile: row (16 x fragment + fr), logical chunk 4 s + fq, swizzled by (row >> 1) & 7 = (fr >> 1) & 7
  const unsigned sw = (unsigned)((fr >> 1) & 7);
  const unsigned rdA = (unsigned)((wr * 64 + fr) * 128), rdB = (unsigned)((wc * 32 + fr) * 128);
  const unsigned cx0 = (((unsigned)fq) ^ sw) << 4, cx1 = (((unsigned)(4 + fq)) ^ sw) << 4;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = (tile / nNt) * BM, n0 = (tile % nNt) * BN;
    // ---- staging sources: this wave's two pieces (8 LDS rows each) of each of the four half-tile kinds
    unsigned oAe[2], oAl[2], oBe[2], oBl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 8 * (2 * wave + j) + (lane >> 3), c = lane & 7;
      const int lc = c ^ ((r >> 1) & 7);
      const int i = r & 15;
      const int ml = 128 * (r >> 6) + 16 * ((r >> 4) & 3) + i;                     // A-early row; A-late: + 64
      const int nl = 64 * (r >> 5) + 16 * (i >> 2) + 4 * ((r >> 4) & 1) + (i & 3);    // B-early row; B-late: + 8
      oAe[j] = ((unsigned)min(m0 + ml, M - 1) * (unsigned)K) * 2u + (unsigned)lc * 16u;
      oAl[j] = ((unsigned)min(m0 + ml + 64, M - 1) * (unsigned)K) * 2u + (unsigned)lc * 16u;
      oBe[j] = ((unsigned)min(n0 + nl, N - 1) * (unsigned)K) * 2u + (unsigned)lc * 16u;
      oBl[j] = ((unsigned)min(n0 + nl + 8, N - 1) * (unsigned)K) * 2u + (unsigned)lc * 16u;
    }
    const unsigned pdst = lds0 + (unsigned)wave * 2048u;
    // half h of K-tile t into buffer (t & 1)
#define E8_STAGE(h, t)                                                                                                   \
  do {                                                                                                                   \
    const unsigned kb__ = (unsigned)(t) * 128u;                                                                          \
    const unsigned d__ = pdst + (unsigned)((t) & 1) * BUF + (unsigned)(h) * HALF;                                        \
    if ((h) == 0) { piece(W, oBe[0] + kb__, d__); piece(W, oBe[1] + kb__, d__ + 1024u); }                                \
    if ((h) == 1) { piece(A, oAe[0] + kb__, d__); piece(A, oAe[1] + kb__, d__ + 1024u); }                                \
    if ((h) == 2) { piece(W, oBl[0] + kb__, d__); piece(W, oBl[1] + kb__, d__ + 1024u); }                                \
    if ((h) == 3) { piece(A, oAl[0] + kb__, d__); piece(A, oAl[1] + kb__, d__ + 1024u); }                                \
  } while (0)

    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = z4;

    // ---- prologue: K-tile 0 whole, K-tile 1 without its A-late half (that one is phase 1's)
    E8_STAGE(0, 0); E8_STAGE(1, 0); E8_STAGE(2, 0); E8_STAGE(3, 0);
    E8_STAGE(0, 1); E8_STAGE(1, 1); E8_STAGE(2, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (wr == 1) asm volatile("s_barrier" ::: "memory");   // the second wave group runs one barrier behind from here on

    bf16x8 fa[4][2], fbe[2][2], fbl[2][2];                 // A fragments of the current row half; W fragments early / late
#define E8_READ_A(bufb, half)                                                                                            \
  _Pragma("unroll") for (int mf = 0; mf < 4; ++mf) {                                                                     \
    const unsigned char* p__ = smem + (bufb) * BUF + (half) * HALF + rdA + mf * 2048;                                    \
    fa[mf][0] = *reinterpret_cast<const bf16x8*>(p__ + cx0);                                                             \
    fa[mf][1] = *reinterpret_cast<const bf16x8*>(p__ + cx1);                                                             \
  }
#define E8_READ_B(dst, bufb, half)                                                                                       \
  _Pragma("unroll") for (int f = 0; f < 2; ++f) {                                                                        \
    const unsigned char* p__ = smem + (bufb) * BUF + (half) * HALF + rdB + f * 2048;                                     \
    dst[f][0] = *reinterpret_cast<const bf16x8*>(p__ + cx0);                                                             \
    dst[f][1] = *reinterpret_cast<const bf16x8*>(p__ + cx1);                                                             \
  }
#define E8_MFMA(mh, nh, fb)                                                                                              \
  do {                                                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                                       \
    _Pragma("unroll") for (int mf = 0; mf < 4; ++mf)                                                                     \
      _Pragma("unroll") for (int f = 0; f < 2; ++f) {                                                                    \
        acc[(mh) * 4 + mf][(nh) * 2 + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[f][0], fa[mf][0], acc[(mh) * 4 + mf][(nh) * 2 + f], 0, 0, 0); \
        acc[(mh) * 4 + mf][(nh) * 2 + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[f][1], fa[mf][1], acc[(mh) * 4 + mf][(nh) * 2 + f], 0, 0, 0); \
      }                                                                                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                                       \
  } while (0)
#define E8_BAR() asm volatile("s_barrier" ::: "memory")       /* with the clobber: no LDS read may move across it */
#define E8_LGKM0()                                       \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)

    // four phases of the K-tile in buffer `bufb`; S1..S4: what each phase stages (a statement; empty in the last iteration)
#define E8_KTILE(bufb, S1, S2, S3, S4, WAIT4)                                                                            \
  do {                                                                                                                   \
    /* phase 1: W early (4 reads, first), A early (8 reads) */                                                           \
    E8_READ_B(fbe, bufb, 0);                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    E8_READ_A(bufb, 1);                                                                                                  \
    S1;                                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");        /* the W-early reads are back: their half may be restaged next phase */ \
    E8_BAR(); E8_LGKM0();                                                                                                \
    E8_MFMA(0, 0, fbe);                                                                                                  \
    E8_BAR();                                                                                                            \
    /* phase 2: W late */                                                                                                \
    E8_READ_B(fbl, bufb, 2);                                                                                             \
    S2;                                                                                                                  \
    E8_BAR(); E8_LGKM0();                                                                                                \
    E8_MFMA(0, 1, fbl);                                                                                                  \
    E8_BAR();                                                                                                            \
    /* phase 3: A late (the same registers: the early rows' last MFMA has been issued) */                                \
    E8_READ_A(bufb, 3);                                                                                                  \
    S3;                                                                                                                  \
    E8_BAR(); E8_LGKM0();                                                                                                \
    E8_MFMA(1, 1, fbl);                                                                                                  \
    E8_BAR();                                                                                                            \
    /* phase 4: no reads; the counted wait that makes the OTHER buffer readable from the next phase on */                \
    S4;                                                                                                                  \
    WAIT4;                                                                                                               \
    E8_BAR();                                                                                                            \
    E8_MFMA(1, 0, fbe);                                                                                                  \
    E8_BAR();                                                                                                            \
  } while (0)

    for (int it = 0; it < nk / 2 - 1; ++it) {
      const int te = 2 * it;
      E8_KTILE(0, E8_STAGE(3, te + 1), E8_STAGE(0, te + 2), E8_STAGE(1, te + 2), E8_STAGE(2, te + 2),
               asm volatile("s_waitcnt vmcnt(6)" ::: "memory"));
      E8_KTILE(1, E8_STAGE(3, te + 2), E8_STAGE(0, te + 3), E8_STAGE(1, te + 3), E8_STAGE(2, te + 3),
               asm volatile("s_waitcnt vmcnt(6)" ::: "memory"));
    }
    {   // last pair of K-tiles: only the odd tile's A-late half is still to come
      const int te = nk - 2;
      E8_KTILE(0, E8_STAGE(3, te + 1), (void)0, (void)0, (void)0, asm volatile("s_waitcnt vmcnt(0)" ::: "memory"));
      E8_KTILE(1, (void)0, (void)0, (void)0, (void)0, (void)0);
    }
    if (wr == 0) asm volatile("s_barrier" ::: "memory");   // realign the two groups: the next tile's prologue writes both buffers

    // ---- C: lane (fr, fq) holds rows 16 MF + fr, columns 16 fq + 4 F + j of its wave's 128 x 64 block
    const int nb = n0 + 64 * wc + 16 * fq;
    float bs[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) bs[q] = bias[min(nb + q, N - 1)];
#pragma unroll
    for (int MF = 0; MF < 8; ++MF) {
      const int m = m0 + 128 * wr + 16 * MF + fr;
      if (m < M && nb < N) {
        unsigned pk[8];
#pragma unroll
        for (int F = 0; F < 4; ++F) {
          float v0 = acc[MF][F][0] + bs[4 * F], v1 = acc[MF][F][1] + bs[4 * F + 1], v2 = acc[MF][F][2] + bs[4 * F + 2], v3 = acc[MF][F][3] + bs[4 * F + 3];
          if (EPI == TR_EPI_GELU_BF16) {
            const f32x2 g0 = gelu2(f32x2{v0, v1}), g1 = gelu2(f32x2{v2, v3});
            v0 = g0[0]; v1 = g0[1]; v2 = g1[0]; v3 = g1[1];
          }
          pk[2 * F] = pack_bf16x2(v0, v1);
          pk[2 * F + 1] = pack_bf16x2(v2, v3);
        }
        uint16_t* o = out + (size_t)m * N + nb;
        *reinterpret_cast<uint4*>(o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint4*>(o + 8) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
#undef E8_STAGE
#undef E8_READ_A
#undef E8_READ_B
#undef E8_MFMA
#undef E8_KTILE
#undef E8_BAR
#undef E8_LGKM0
  }
}

}  // namespace e8
