// Dev tool (round 4): the guide's 8-phase GEMM structure (tools/gemm_variants/tr_gemm_8p.hip) against the product kernel, same process, interleaved
// rounds, outputs compared bit for bit.   hipcc -O3 --offload-arch=gfx950 tools/gemm_8p_lab.cpp -o gemm_8p_lab
#include "../tokenreduction_amd/csrc/tr_gemm.hip"
#include "gemm_variants/tr_gemm_8p.hip"
#include <vector>
#include <cstdlib>
#include <cstring>
void tr_set_error(const char* fmt, ...) { (void)fmt; }
void tr_prof_note(const char*, double, double) {}
void tr_prof_mark(const char*) {}
template <int EPI>
static void run8p(const uint16_t* A, const uint16_t* W, const float* bias, uint16_t* out, int M, int N, int K) {
  const int nMt = (M + 255) / 256, nNt = (N + 255) / 256, nt = nMt * nNt;
  static bool once = false;
  if (!once) { hipFuncSetAttribute(reinterpret_cast<const void*>(e8::gemm_bf16_8p<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072); once = true; }
  hipLaunchKernelGGL((e8::gemm_bf16_8p<EPI>), dim3(nt < 256 ? nt : 256), dim3(512), 131072, nullptr, A, W, bias, out, M, N, K, nt, nNt);
}
int main(int argc, char** argv) {
  struct Shape { const char* name; int M, N, K, epi; };
  Shape shapes[] = {{"fc1-S", 50432, 1536, 384, TR_EPI_BF16}, {"fc1-S gelu", 50432, 1536, 384, TR_EPI_GELU_BF16}, {"qkv-S", 50432, 1152, 384, TR_EPI_BF16},
                    {"fc2-S", 50432, 384, 1536, TR_EPI_BF16}, {"qkv-B", 25216, 2304, 768, TR_EPI_BF16}, {"fc1-B", 25216, 3072, 768, TR_EPI_BF16},
                    {"fc2-B", 25216, 768, 3072, TR_EPI_BF16}, {"proj-B", 25216, 768, 768, TR_EPI_BF16}, {"ragged", 1000, 512, 128, TR_EPI_BF16},
                    {"4096^3", 4096, 4096, 4096, TR_EPI_BF16}};
  for (auto& sh : shapes) {
    size_t na = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, no = (size_t)sh.M * sh.N;
    uint16_t *A, *W, *o0, *o1; float* bias;
    hipMalloc(&A, na * 2); hipMalloc(&W, nw * 2); hipMalloc(&bias, sh.N * 4); hipMalloc(&o0, no * 2); hipMalloc(&o1, no * 2);
    std::vector<uint16_t> h(na > nw ? na : nw);
    srand(1);
    for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    hipMemcpy(A, h.data(), na * 2, hipMemcpyHostToDevice);
    for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    hipMemcpy(W, h.data(), nw * 2, hipMemcpyHostToDevice);
    std::vector<float> hb(sh.N);
    for (auto& v : hb) v = (float)(rand() % 64 - 32) / 256.f;
    hipMemcpy(bias, hb.data(), sh.N * 4, hipMemcpyHostToDevice);
    hipMemset(o0, 0, no * 2); hipMemset(o1, 0xff, no * 2);
    auto pc = [&] { tr_gemm_bf16(A, W, bias, o0, nullptr, 0, sh.M, sh.N, sh.K, sh.epi, nullptr); };
    auto p8 = [&] { if (sh.epi == TR_EPI_GELU_BF16) run8p<TR_EPI_GELU_BF16>(A, W, bias, o1, sh.M, sh.N, sh.K); else run8p<TR_EPI_BF16>(A, W, bias, o1, sh.M, sh.N, sh.K); };
    pc(); p8();
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", sh.name, hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<uint16_t> r0(no), r1(no);
    hipMemcpy(r0.data(), o0, no * 2, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), o1, no * 2, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < no; ++i) if (r0[i] != r1[i]) { if (!bad) first = i; ++bad; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best[2] = {1e9, 1e9};
    for (int round = 0; round < 4; ++round)
      for (int v = 0; v < 2; ++v) {
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < 10; ++i) { if (v == 0) pc(); else p8(); }
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms * 100.0 < best[v]) best[v] = ms * 100.0;
      }
    const double fl = 2.0 * sh.M * sh.N * sh.K;
    printf("%-10s M=%6d N=%5d K=%5d  pc %7.1f us %6.0f TF | 8p %7.1f us %6.0f TF | mismatches %zu of %zu (first at %zu: %04x vs %04x)\n", sh.name, sh.M, sh.N,
           sh.K, best[0], fl / best[0] / 1e6, best[1], fl / best[1] / 1e6, bad, no, first, bad ? r0[first] : 0, bad ? r1[first] : 0);
    hipFree(A); hipFree(W); hipFree(bias); hipFree(o0); hipFree(o1);
  }
  return 0;
}
