// Dev tool: times the product GEMM kernel (tokenreduction_amd/csrc/tr_gemm.hip, included verbatim) on the DeiT-S shapes,
// optionally with one cost ablated (-DTR_ABLATE_NO_STORE / _NO_MFMA / _NO_LDS / _NO_DMA) to find the dominant cost
// (cdna guide section 7 "the diagnostic loop", step 2).  Ablated builds compute garbage; only their time is read.
//   hipcc -O3 --offload-arch=gfx950 -I.. tools/gemm_lab.cpp -o gemm_lab [-DTR_ABLATE_...]
#include "../tokenreduction_amd/csrc/tr_gemm.hip"
#include <vector>
#include <cstdlib>
static thread_local char g_err[512];
void tr_set_error(const char* fmt, ...) { (void)fmt; }
int main(int argc, char** argv) {
  struct Shape { const char* name; int M, N, K, epi; };
  Shape shapes[] = {{"qkv  s1", 50432, 1152, 384, TR_EPI_BF16},   {"fc1  s1", 50432, 1536, 384, TR_EPI_GELU_BF16},
                    {"proj s1", 50432, 384, 384, TR_EPI_BF16}, {"fc2  s1", 50432, 384, 1536, TR_EPI_BF16},
                    {"qkv  s4", 17408, 1152, 384, TR_EPI_BF16},   {"fc2  s4", 17408, 384, 1536, TR_EPI_BF16},
                    {"big     ", 8192, 8192, 8192, TR_EPI_BF16}};
  for (auto& sh : shapes) {
    size_t na = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, no = (size_t)sh.M * sh.N;
    uint16_t *A, *W; float* bias; void* out;
    hipMalloc(&A, na * 2); hipMalloc(&W, nw * 2); hipMalloc(&bias, sh.N * 4); hipMalloc(&out, no * 4);
    std::vector<uint16_t> h(na > nw ? na : nw);
    srand(1);
    for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));  // random-sign bf16 around +-[0.008,0.03]
    hipMemcpy(A, h.data(), na * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, sh.N * 4); hipMemset(out, 0, no * 4);
#ifdef TR_DIAG_STAMPS
    if (sh.epi <= TR_EPI_GELU_BF16 && sh.M == 50432) {
      unsigned long long* stamps; hipMalloc(&stamps, 128 * 5 * 8); hipMemset(stamps, 0, 128 * 5 * 8);
      for (int i = 0; i < 3; ++i) tr_gemm_bf16(A, W, bias, out, (const float*)stamps, 0, sh.M, sh.N, sh.K, sh.epi, nullptr);
      hipDeviceSynchronize();
      std::vector<unsigned long long> hs(128 * 5);
      hipMemcpy(hs.data(), stamps, 128 * 5 * 8, hipMemcpyDeviceToHost);
      for (int w = 0; w < 2; ++w) {
        printf("  wave %d: step: H1   mid-wait+barrier   H2   END(epilogue)   loop-back | total   (shader cycles)\n", w ? 7 : 0);
        for (int g = 0; g < 26; ++g) {
          unsigned long long* t = &hs[(w * 64 + g) * 5];
          unsigned long long* tn = &hs[(w * 64 + g + 1) * 5];
          printf("   g=%2d  %6llu  %6llu  %6llu  %6llu  %6llu | %6llu\n", g, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], tn[0] - t[4], tn[0] - t[0]);
        }
      }
      hipFree(stamps);
    }
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) tr_gemm_bf16(A, W, bias, out, nullptr, 0, sh.M, sh.N, sh.K, sh.epi, nullptr);
    hipDeviceSynchronize();
    const int iters = 30;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) tr_gemm_bf16(A, W, bias, out, nullptr, 0, sh.M, sh.N, sh.K, sh.epi, nullptr);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double us = 1e3 * ms / iters, tf = 2.0 * sh.M * sh.N * sh.K / (us * 1e-6) / 1e12;
    printf("%-8s M=%6d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s", sh.name, sh.M, sh.N, sh.K, us, tf);
#ifdef TR_DIAG_CLOCK
    if (sh.epi <= TR_EPI_GELU_BF16) {
      unsigned long long hs[2];
      hipMemcpy(hs, (char*)out + (size_t)sh.M * sh.N * 2, 16, hipMemcpyDeviceToHost);
      printf("   [WG 8: %llu shader cycles in %.2f us -> %.0f MHz]", hs[0], hs[1] / 100.0, hs[0] / (hs[1] / 100.0));
    }
#endif
    printf("\n");
    hipFree(A); hipFree(W); hipFree(bias); hipFree(out);
  }
  return 0;
}
