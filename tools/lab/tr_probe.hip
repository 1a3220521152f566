// Probe of ds_read_b64_tr_b16 semantics on gfx950 (cdna guide T10): per 16-lane group, lane 4q+p supplies the address of row q,
// columns 4p..4p+3 of a 4x16 block of 16-bit elements; lane i receives column i, row q in element q.
// Build: hipcc --offload-arch=gfx950 -O2 tools/lab/tr_probe.hip -o tools/bin/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const unsigned short* in, int* out) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) sm[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + (8 * g + q) * 64 + 4 * p));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + (8 * g + 4 + q) * 64 + 4 * p));
  for (int e = 0; e < 4; ++e) { out[lane * 8 + e] = a[e]; out[lane * 8 + 4 + e] = b[e]; }
}
int main() {
  unsigned short h[64 * 64];
  for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c) h[r * 64 + c] = (unsigned short)(r * 64 + c);
  unsigned short* d; int* o; int ho[64 * 8];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
    const int want = (8 * (l >> 4) + e) * 64 + (l & 15);   // element e of lane l = A[k = 8(l>>4)+e][col l&15]
    if (ho[l * 8 + e] != want) { if (bad < 8) printf("lane %d e %d got %d (r%d c%d) want r%d c%d\n", l, e, ho[l*8+e], ho[l*8+e]/64, ho[l*8+e]%64, want/64, want%64); ++bad; }
  }
  printf(bad ? "TR_PROBE FAIL %d\n" : "TR_PROBE PASS\n", bad);
  return bad != 0;
}
