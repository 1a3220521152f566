// Lab: which resource keeps a memory-bound launch of one stream from becoming resident beside a persistent one-workgroup-per-CU launch of
// another stream?  "hog": 256 workgroups that hold T threads, R vector registers per lane and L bytes of LDS and spin for a fixed wall time
// (s_memrealtime) -- a stand-in for gemm_bf16_pc / mlp_fused_kernel with no memory traffic and no matrix work.  "copy": a streaming
// float4 copy of 256 MiB, 256-thread workgroups, < 32 registers, no LDS -- a stand-in for the LayerNorm launches.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/coresidency_hw tools/lab/coresidency_hw.hip && /tmp/coresidency_hw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int R>
__global__ __launch_bounds__(1024) void hog(unsigned long long ticks, unsigned* sink) {
  extern __shared__ unsigned char lds[];
  if (R >= 64) asm volatile("v_mov_b32 v%0, 0" ::"n"(R - 1) : "memory");     // the allocation follows the highest register named
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks == 0x7fffffffffffffffull) sink[0] = lds[threadIdx.x];
}
template <> __global__ __launch_bounds__(1024) void hog<168>(unsigned long long ticks, unsigned* sink) {
  extern __shared__ unsigned char lds[];
  asm volatile("v_mov_b32 v167, 0" ::: "v167", "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks == 0x7fffffffffffffffull) sink[0] = lds[threadIdx.x];
}
template <> __global__ __launch_bounds__(1024) void hog<120>(unsigned long long ticks, unsigned* sink) {
  extern __shared__ unsigned char lds[];
  asm volatile("v_mov_b32 v119, 0" ::: "v119", "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks == 0x7fffffffffffffffull) sink[0] = lds[threadIdx.x];
}
template <> __global__ __launch_bounds__(1024) void hog<32>(unsigned long long ticks, unsigned* sink) {
  extern __shared__ unsigned char lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks == 0x7fffffffffffffffull) sink[0] = lds[threadIdx.x];
}
template <int R>
__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  if (R == 48) asm volatile("v_mov_b32 v47, 0" ::: "v47", "memory");
  if (R == 24) asm volatile("v_mov_b32 v23, 0" ::: "v23", "memory");
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i];
}
template <> __global__ __launch_bounds__(1024) void hog<152>(unsigned long long ticks, unsigned* sink) {
  extern __shared__ unsigned char lds[];
  asm volatile("v_mov_b32 v151, 0" ::: "v151", "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks == 0x7fffffffffffffffull) sink[0] = lds[threadIdx.x];
}

int main() {
  const size_t n = (size_t)128 << 20 >> 4;                 // 128 MiB read + 128 MiB written
  float4 *a, *b; unsigned* sink;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&sink, 4096));
  CK(hipMemset(a, 1, n * 16));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  CK(hipFuncSetAttribute((const void*)hog<168>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)hog<120>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)hog<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const unsigned long long ticks = 6000;                  // 60 us at 100 MHz
  const int reps = 40;
  CK(hipFuncSetAttribute((const void*)hog<152>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto run = [&](int R, int T, int L, int CR, int CT, bool with_hog, bool with_copy) {
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) {
      if (with_hog) {
        if (R == 168) hipLaunchKernelGGL(hog<168>, dim3(256), dim3(T), L, sa, ticks, sink);
        else if (R == 152) hipLaunchKernelGGL(hog<152>, dim3(256), dim3(T), L, sa, ticks, sink);
        else if (R == 120) hipLaunchKernelGGL(hog<120>, dim3(256), dim3(T), L, sa, ticks, sink);
        else hipLaunchKernelGGL(hog<32>, dim3(256), dim3(T), L, sa, ticks, sink);
      }
      if (with_copy) {
        const dim3 g((unsigned)((n + CT - 1) / CT)), bl(CT);
        if (CR == 48) hipLaunchKernelGGL(copy4<48>, g, bl, 0, sb, a, b, n);
        else if (CR == 24) hipLaunchKernelGGL(copy4<24>, g, bl, 0, sb, a, b, n);
        else hipLaunchKernelGGL(copy4<8>, g, bl, 0, sb, a, b, n);
      }
    }
    CK(hipDeviceSynchronize());
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
  };
  const int L160 = 160 * 1024;
  const int cfg[][5] = {{168, 768, L160, 8, 256},  {168, 768, L160, 24, 256}, {168, 768, L160, 48, 256}, {168, 640, L160, 48, 256}, {168, 640, L160, 48, 64},
                        {168, 640, L160, 48, 128}, {168, 640, L160, 24, 256}, {168, 512, L160, 48, 256}, {152, 768, L160, 48, 256}, {152, 768, L160, 48, 64},
                        {152, 768, L160, 24, 256}, {120, 768, L160, 48, 256}, {32, 768, L160, 48, 256},  {32, 256, 0, 48, 256}};
  for (const auto& c : cfg) {
    run(c[0], c[1], c[2], c[3], c[4], true, true);
    const double h = run(c[0], c[1], c[2], c[3], c[4], true, false), m = run(c[0], c[1], c[2], c[3], c[4], false, true),
                 both = run(c[0], c[1], c[2], c[3], c[4], true, true);
    printf("hog %3d regs x %4d threads, %3d KiB LDS | copy %2d regs x %3d threads: alone %6.1f us, copy alone %6.1f us, both streams %6.1f us per pair (sum %6.1f)\n",
           c[0], c[1], c[2] >> 10, c[3], c[4], h, m, both, h + m);
  }
  return 0;
}
