// lab: what do v_permlane16_swap / v_permlane32_swap return?  hipcc --offload-arch=gfx950 tools/lab/permlane_probe.hip -o tools/bin/permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float quad_rows_max(float v) {
  // (the elements of the builtin's result are copied to scalars first: __builtin_bit_cast applied to `a[1]` directly read element 0
  // -- hipcc 7.2 -- and the reduction silently lost the partner's value)
  const auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned a0 = a[0], a1 = a[1];
  v = fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1));
  const auto c = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned c0 = c[0], c1 = c[1];
  return fmaxf(__builtin_bit_cast(float, c0), __builtin_bit_cast(float, c1));
}
__global__ void k(float* out, const float* in) {
  const unsigned l = threadIdx.x;
  float x = in[l];
  out[l] = quad_rows_max(x);
  float y = x * 2.0f + 1.0f;
  out[64 + l] = quad_rows_max(y);
}
int main() {
  float* d; float* di; hipMalloc(&d, 128 * 4); hipMalloc(&di, 64 * 4);
  float hi[64]; for (int i = 0; i < 64; ++i) hi[i] = (float)((i * 37) % 64);
  hipMemcpy(di, hi, sizeof(hi), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, di);
  float h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 2; ++r) {
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
      float want = 0; for (int g = 0; g < 4; ++g) { float v = hi[(i & 15) + 16 * g]; if (r) v = v * 2 + 1; want = fmaxf(want, v); }
      if (h[r * 64 + i] != want) ++bad;
    }
    printf("case %d: %d wrong lanes; first values %g %g %g (in %g %g %g %g)\n", r, bad, h[r*64], h[r*64+1], h[r*64+17], hi[0], hi[16], hi[32], hi[48]);
  }
  return 0;
}
