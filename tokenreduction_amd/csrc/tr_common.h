// Internal helpers shared by the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/tokenreduction_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define TR_WAVE 64

// thread-local error text behind tr_last_error()
void tr_set_error(const char* fmt, ...);

#define TR_REQUIRE(cond, code, ...)        \
  do {                                     \
    if (!(cond)) {                         \
      tr_set_error(__VA_ARGS__);           \
      return (code);                       \
    }                                      \
  } while (0)

// Launch profiler (tr_profile_begin / tr_profile_end, csrc/tr_vit.hip): while a recording is active (process-wide: autograd runs the
// backward on its own thread), every TR_CHECK_LAUNCH drops a HIP event on the recording's stream, so consecutive marks bracket the launches of one entry point (the
// event-bracketed duration: kernel + its dependent-launch boundary).  tr_prof_note() names the next mark and gives it the
// algorithmic FLOPs / bytes of the launch; without a note the mark carries the entry point's name and zeros.  Inactive: one
// thread-local load per launch.
void tr_prof_mark(const char* label);
void tr_prof_restart();
void tr_prof_note(const char* label, double flops, double bytes);

#define TR_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      tr_set_error("%s: launch failed: %s", (name), hipGetErrorString(e__));    \
      return TR_ERR_LAUNCH;                                                     \
    }                                                                           \
    tr_prof_mark(name);                                                         \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is sticky per kernel AND per device: raise it only when a launch needs more than any earlier one
// on the current device did (otherwise a host-side driver call per launch).  One cache per expansion site: use one site per kernel
// instantiation.  (A single process-wide cache left the second GPU of a one-process multi-GPU program at the default 64 KB limit.)
#define TR_RESERVE_LDS(fn_ptr, bytes, what)                                                                                     \
  do {                                                                                                                          \
    static std::atomic<size_t> have__[64];                                                                                      \
    int dev__ = 0;                                                                                                              \
    (void)hipGetDevice(&dev__);                                                                                                 \
    std::atomic<size_t>& slot__ = have__[(unsigned)dev__ & 63u];                                                                \
    const size_t want__ = (bytes);                                                                                              \
    if (want__ > slot__.load(std::memory_order_relaxed)) {                                                                      \
      hipError_t e__ = hipFuncSetAttribute((fn_ptr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want__);                  \
      TR_REQUIRE(e__ == hipSuccess, TR_ERR_LAUNCH, "%s: cannot reserve %zu B of LDS: %s", (what), want__, hipGetErrorString(e__)); \
      slot__.store(want__, std::memory_order_relaxed);                                                                          \
    }                                                                                                                           \
  } while (0)

static inline bool tr_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) {
  return __builtin_bit_cast(float, (unsigned int)h << 16);
}

// two floats -> packed bf16x2 (RNE; a plain cast lowers to v_cvt_pk_bf16_f32 and keeps NaN a NaN)
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned int, v);
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it waits for every global load the
// wave has in flight -- which turns a register-staged prefetch (loads of slab k+1 issued before the MFMAs of slab k) into a
// load-wait-compute sequence.  Use between LDS producers and consumers when global loads should stay in flight across it.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane: quad butterfly, half-row mirror, row mirror --
// four full-rate v_add_f32_dpp (a __shfl_xor is a ds_bpermute on gfx9: LDS pipe, and that pipe feeds the MFMA operands)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));    // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));    // row_mirror
  return v;
}

typedef __attribute__((ext_vector_type(2))) float f32x2;

// GELU(x) = x * Phi(x) (the exact-erf form nn.GELU() computes) with Phi(x) ~= sigmoid(a1 x + a3 x^3 + a5 x^5), a minimax fit
// over [-8, 8]: |gelu_fit - gelu_erf| <= 2.6e-5 absolute everywhere (tools/fit_gelu.py) -- 100x below the bf16 resolution of
// the hidden activations it is rounded to.  9 VALU ops per element, written on float2 so hipcc emits v_pk_{mul,fma,add}_f32.
// (The Abramowitz-Stegun erf used in the first version cost ~27 ops/element: in-kernel stamps showed the GELU epilogue at
// ~8k cycles per 256x128 tile, as much as the tile's whole K-loop.  A transcendental-free form -- Phi - 1/2 as a degree-9 polynomial in
// x^2, 14 packed ops per pair, 4e-5 abs -- was measured 9 % SLOWER per fc1 launch than this one: exp2 and rcp are cheap next to ten FMAs.)
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
  constexpr float L2E = 1.44269504088896340736f;
  constexpr float C1 = -1.59501577f * L2E, C3 = -7.40112920e-02f * L2E, C5 = 7.03033576e-04f * L2E;
  f32x2 xc;
  xc[0] = __builtin_amdgcn_fmed3f(x[0], -8.0f, 8.0f);
  xc[1] = __builtin_amdgcn_fmed3f(x[1], -8.0f, 8.0f);
  const f32x2 x2 = xc * xc;
  f32x2 p = x2 * C5 + C3;
  p = p * x2 + C1;
  const f32x2 z = p * xc;                               // -log2(e) * (a1 x + a3 x^3 + a5 x^5)
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(z[0]);
  e[1] = __builtin_amdgcn_exp2f(z[1]);
  e = e + 1.0f;
  f32x2 r;
  r[0] = __builtin_amdgcn_rcpf(e[0]);
  r[1] = __builtin_amdgcn_rcpf(e[1]);
  return x * r;
}

// d/dx [x Phi(x)] = Phi(x) + x phi(x) = 1/2 + o(x), o odd (the exact-erf derivative nn.GELU differentiates): o(x) ~= x P(t) with
// t = 2 x^2 / c^2 - 1 on |x| <= c = 4.5, x clamped beyond (o(4.5) is within 7e-5 of its limit 1/2); P of degree 9 in the monomial basis of
// t in [-1, 1], where fp32 Horner has no cancellation (tools/fit_gelu_grad.py).  |gelu2_grad - exact| <= 6e-5 absolute everywhere: 60x below
// the bf16 resolution of the gradient it multiplies.  No transcendental: 13 packed ops per PAIR of elements (the first version, the derivative
// of the forward's sigmoid fit, needed exp2 + rcp per element, quarter rate -- the fused epilogue of tr_gemm_dgelu_bf16 is VALU-bound).
__device__ __forceinline__ f32x2 gelu2_grad(f32x2 x) {
  constexpr float C = 4.5f;
  constexpr float P0 = 1.594457889e-01f, P1 = -8.990845048e-02f, P2 = 8.617554151e-02f, P3 = -9.554615128e-02f, P4 = 1.051488888e-01f,
                  P5 = -8.720398095e-02f, P6 = 4.744845022e-02f, P7 = -4.606752329e-02f, P8 = 5.617017796e-02f, P9 = -2.454702451e-02f;
  f32x2 xc;
  xc[0] = __builtin_amdgcn_fmed3f(x[0], -C, C);
  xc[1] = __builtin_amdgcn_fmed3f(x[1], -C, C);
  const f32x2 t = (xc * xc) * (2.0f / (C * C)) - 1.0f;
  f32x2 p = t * P9 + P8;
  p = p * t + P7;
  p = p * t + P6;
  p = p * t + P5;
  p = p * t + P4;
  p = p * t + P3;
  p = p * t + P2;
  p = p * t + P1;
  p = p * t + P0;
  return p * xc + 0.5f;
}

// eight bf16 gradients times gelu'(eight bf16 pre-activations), rounded back to bf16: one 16-byte line of tr_gelu_bwd_bf16 and of the
// fused data-gradient epilogue (tr_gemm_dgelu_bf16) -- the two are bitwise the same by construction
__device__ __forceinline__ void dgelu_line(const unsigned int (&dh)[4], const unsigned int (&pre)[4], unsigned int (&out)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 x = {__builtin_bit_cast(float, pre[i] << 16), __builtin_bit_cast(float, pre[i] & 0xffff0000u)};
    const f32x2 g = {__builtin_bit_cast(float, dh[i] << 16), __builtin_bit_cast(float, dh[i] & 0xffff0000u)};
    const f32x2 d = g * gelu2_grad(x);
    out[i] = pack_bf16x2(d[0], d[1]);
  }
}

// XCD-aware bijective remap of a linear block id (cdna guide T1): blocks b and b+8 share an XCD/L2, so give
// each XCD a CONTIGUOUS chunk of the logical tile order.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}
