// Row helpers shared by the kernels that finish with a LayerNorm over a token row held in registers
// (tr_norm.hip, tr_tome.hip, tr_cluster.hip): one wave per row, `LN_MAX_CHUNKS` float4 chunks per lane.
#ifndef TR_ROWOPS_H
#define TR_ROWOPS_H
#include "tr_common.h"

namespace {

constexpr int LN_MAX_CHUNKS = 4;  // float4 chunks per lane -> D <= 1024
// launch KERNEL<..., NCH> with NCH = ceil(D / 256) in 1..4
#define TR_DISPATCH_NCH(D_, ...)                        \
  do {                                                  \
    switch (((D_) + 255) / 256) {                       \
      case 1: { constexpr int NCH = 1; __VA_ARGS__; } break; \
      case 2: { constexpr int NCH = 2; __VA_ARGS__; } break; \
      case 3: { constexpr int NCH = 3; __VA_ARGS__; } break; \
      default: { constexpr int NCH = 4; __VA_ARGS__; } break; \
    }                                                   \
  } while (0)

// Normalise one row held as `nch` float4 chunks per lane; two-pass (mean, then centred variance) in registers.
// NCH = 64-lane chunks the row actually spans (ceil(D / 256)): the callers instantiate per NCH so that no lane issues loads
// for chunks the row does not have (the branch-free loads clamp the chunk index instead of predicating).
template <bool F32, int NCH>
__device__ __forceinline__ void ln_row_store(float4 (&v)[NCH], int nchunks, int lane, int D, float eps,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             void* __restrict__ yrow_) {
  // gamma / beta do not depend on the statistics: fetched first, branch-free (chunk index clamped), so they are in flight during
  // the two reductions instead of costing one dependent round trip per chunk in the output loop
  float4 gm[NCH], bt[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = min(lane + 64 * c, nchunks - 1);
    gm[c] = *reinterpret_cast<const float4*>(gamma + 4 * ch);
    bt[c] = *reinterpret_cast<const float4*>(beta + 4 * ch);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (lane + 64 * c < nchunks) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    if (lane + 64 * c < nchunks) {
      const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  const float rstd = F32 ? 1.0f / sqrtf(wave_sum(q) / (float)D + eps) : rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
      const float4 g = gm[c], b = bt[c];
      const float o0 = (v[c].x - mean) * rstd * g.x + b.x, o1 = (v[c].y - mean) * rstd * g.y + b.y;
      const float o2 = (v[c].z - mean) * rstd * g.z + b.z, o3 = (v[c].w - mean) * rstd * g.w + b.w;
      if (F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(yrow_) + 4 * ch) = make_float4(o0, o1, o2, o3);
      } else {
        uint2 pk;
        pk.x = pack_bf16x2(o0, o1);
        pk.y = pack_bf16x2(o2, o3);
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(yrow_) + 4 * ch) = pk;
      }
    }
  }
}

// The fp32 residual stream (and the bf16 delta) is read once and written once per LayerNorm and not touched again until the next
// one, ~400 MB of GEMM/attention traffic later: nontemporal loads/stores keep it from displacing the lines the next kernel wants
// (the normalised bf16 output, which the following GEMM reads at once, is stored with the default policy).  Measured with cold
// caches at M = 50432, D = 384: 50.4 -> 40.9 us (tools/ln_lab.py); back to back on cache-resident rows it costs 7 %.
#ifdef TR_LN_NO_NT
#define LN_LOAD(p) (*(p))
#define LN_STORE(v, p) (*(p) = (v))
#else
#define LN_LOAD(p) __builtin_nontemporal_load(p)
#define LN_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif

typedef __attribute__((ext_vector_type(4))) float ln_f4;
__device__ __forceinline__ float4 ln_nt_load4(const float* p) {
  const ln_f4 t = LN_LOAD(reinterpret_cast<const ln_f4*>(p));
  return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ void ln_nt_store4(const float4& v, float* p) {
  const ln_f4 t = {v.x, v.y, v.z, v.w};
  LN_STORE(t, reinterpret_cast<ln_f4*>(p));
}

// 4 bf16 -> 4 fp32
__device__ __forceinline__ float4 bf16x4_to_f32(uint2 u) {
  return make_float4(bf16_bits_to_f32((unsigned short)(u.x & 0xffffu)), bf16_bits_to_f32((unsigned short)(u.x >> 16)),
                     bf16_bits_to_f32((unsigned short)(u.y & 0xffffu)), bf16_bits_to_f32((unsigned short)(u.y >> 16)));
}

// pending-residual chunk: bf16 (fast path) or fp32 (validation path)
template <bool F32>
__device__ __forceinline__ float4 load_delta4(const void* base, size_t elem) {
  if (F32) return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem);
  return bf16x4_to_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem));
}


}  // namespace
#endif  // TR_ROWOPS_H
