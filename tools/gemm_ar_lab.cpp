// Dev tool (round 4): the resident-activation GEMM (tokenreduction_amd/csrc/tr_gemm_ar.hip) against the producer/consumer kernel
// (tr_gemm.hip), both included verbatim: bitwise comparison of TR_EPI_BF16, GELU within one bf16 step, and timings on the K = 384 shapes
// of DeiT-S at batch 256 (BASELINE configs[1]).  -DTR_ABLATE_NO_EPI / _NO_MFMA / _NO_DMA / _NO_STORE / _NO_GELU ablate one cost of
// BOTH kernels (garbage results, time only).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast tools/gemm_ar_lab.cpp -o tools/bin/gemm_ar_lab
#include "../tokenreduction_amd/csrc/tr_gemm.hip"
#include "gemm_variants/tr_gemm_ar.hip"
#include <cmath>
#include <functional>
#include <cstdlib>
#include <cstring>
#include <vector>
void tr_set_error(const char* fmt, ...) { (void)fmt; }
void tr_prof_mark(const char*) {}
void tr_prof_restart() {}
void tr_prof_note(const char*, double, double) {}

static uint16_t f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float bf2f(uint16_t h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static float frand() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }

struct Shape { const char* name; int M, N, K, epi; };

static double time_it(int iters, const std::function<void()>& fn) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) fn();
  hipDeviceSynchronize();
  hipEventRecord(e0, nullptr);
  for (int i = 0; i < iters; ++i) fn();
  hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return 1e3 * ms / iters;
}

#ifdef TR_AR_STAMPS
static void stamp_run() {
  const int M = 50432, N = 1152, K = 384;
  uint16_t *A, *W, *o; float* bias;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&bias, N * 4); hipMalloc(&o, (size_t)M * N * 2);
  hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(W, 0x3c, (size_t)N * K * 2); hipMemset(bias, 0, N * 4);
  for (int i = 0; i < 3; ++i) tr_gemm_bf16_ar(A, W, bias, o, M, N, K, 0, nullptr);
  hipDeviceSynchronize();
  static unsigned long long h[2][128][4];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(ar_stamps), sizeof(h));
  printf("step | MFMA wave 0: first half + hand-over   wait+barrier   second half (to next step) | service wave 0: barrier   round (serve, slabs)   wait for landing | step total\n");
  for (int g = 0; g < 40; ++g) {
    auto* t = h[0][g]; auto* tn = h[0][g + 1]; auto* v = h[1][g]; auto* vn = h[1][g + 1];
    printf("  g=%2d | %5llu %5llu %5llu | %5llu %5llu %5llu | %6llu\n", g, t[1] - t[0], t[2] - t[1], tn[0] - t[2], v[1] - v[0], v[2] - v[1], vn[0] - v[2], tn[0] - t[0]);
  }
}
#endif
int main(int argc, char** argv) {
#ifdef TR_AR_STAMPS
  stamp_run();
  return 0;
#endif
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  std::vector<Shape> shapes = {
      // correctness-first: ragged rows, fewer units than workgroups, K = 320
      {"ragged ", 1000, 384, 384, TR_EPI_BF16},   {"tiny   ", 37, 128, 384, TR_EPI_BF16},     {"k320 g ", 5000, 768, 320, TR_EPI_GELU_BF16},
      {"wide   ", 3000, 1536, 384, TR_EPI_BF16},
      // DeiT-S, batch 256, the four stages of keep_rate 0.7
      {"qkv  s1", 50432, 1152, 384, TR_EPI_BF16}, {"proj s1", 50432, 384, 384, TR_EPI_BF16}, {"fc1  s1", 50432, 1536, 384, TR_EPI_GELU_BF16},
      {"qkv  s2", 35328, 1152, 384, TR_EPI_BF16}, {"proj s2", 35328, 384, 384, TR_EPI_BF16}, {"fc1  s2", 35328, 1536, 384, TR_EPI_GELU_BF16},
      {"qkv  s3", 24832, 1152, 384, TR_EPI_BF16}, {"proj s3", 24832, 384, 384, TR_EPI_BF16}, {"fc1  s3", 24832, 1536, 384, TR_EPI_GELU_BF16},
      {"qkv  s4", 17408, 1152, 384, TR_EPI_BF16}, {"proj s4", 17408, 384, 384, TR_EPI_BF16}, {"fc1  s4", 17408, 1536, 384, TR_EPI_GELU_BF16},
      {"qkv b32", 6304, 1152, 384, TR_EPI_BF16}, {"fc1 b32", 6304, 1536, 384, TR_EPI_GELU_BF16}};
  if (quick) shapes.resize(7);
  double tot_old = 0, tot_new = 0, fl_tot = 0;
  for (auto& sh : shapes) {
    const size_t na = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, no = (size_t)sh.M * sh.N;
    uint16_t *A, *W, *o_old, *o_new; float* bias;
    hipMalloc(&A, na * 2); hipMalloc(&W, nw * 2); hipMalloc(&bias, sh.N * 4); hipMalloc(&o_old, no * 2 + 64); hipMalloc(&o_new, no * 2 + 64);
    srand(7);
    {
      std::vector<uint16_t> h(na);
      for (auto& v : h) v = f2bf(frand() * 1.7f);
      hipMemcpy(A, h.data(), na * 2, hipMemcpyHostToDevice);
      h.resize(nw);
      for (auto& v : h) v = f2bf(frand() * 0.06f);
      hipMemcpy(W, h.data(), nw * 2, hipMemcpyHostToDevice);
      std::vector<float> b(sh.N);
      for (auto& v : b) v = frand() * 0.5f;
      hipMemcpy(bias, b.data(), sh.N * 4, hipMemcpyHostToDevice);
    }
    hipMemset(o_old, 0, no * 2);
    tr_gemm_bf16(A, W, bias, o_old, nullptr, 0, sh.M, sh.N, sh.K, sh.epi, nullptr);
    hipDeviceSynchronize();
    std::vector<uint16_t> h_old(no), h_new(no);
    hipMemcpy(h_old.data(), o_old, no * 2, hipMemcpyDeviceToHost);
    const double fl = 2.0 * sh.M * sh.N * sh.K;
    const double us_old = time_it(20, [&] { tr_gemm_bf16(A, W, bias, o_old, nullptr, 0, sh.M, sh.N, sh.K, sh.epi, nullptr); });
    printf("%-8s M=%6d N=%5d K=%5d %s | pc %7.1f us %6.0f TF |", sh.name, sh.M, sh.N, sh.K, sh.epi == TR_EPI_BF16 ? "bf16" : "gelu", us_old,
           fl / us_old * 1e-6);
    double best = 1e30;
    {
      hipMemset(o_new, 0xff, no * 2);
      int rc = tr_gemm_bf16_ar(A, W, bias, o_new, sh.M, sh.N, sh.K, sh.epi, nullptr);
      hipError_t e = hipDeviceSynchronize();
      if (rc != 0 || e != hipSuccess) { printf(" ar rc=%d err=%s\n", rc, hipGetErrorString(e)); continue; }
      hipMemcpy(h_new.data(), o_new, no * 2, hipMemcpyDeviceToHost);
      size_t bad = 0; double maxd = 0; size_t first = (size_t)-1;
      for (size_t i = 0; i < no; ++i)
        if (h_old[i] != h_new[i]) {
          const double d = fabs((double)bf2f(h_old[i]) - (double)bf2f(h_new[i]));
          const double tol = sh.epi == TR_EPI_BF16 ? 0.0 : fmax(fabs((double)bf2f(h_old[i])) * 0.0079, 4e-5);
          if (!(d <= tol)) { ++bad; if (first == (size_t)-1) first = i; }
          if (d > maxd || d != d) maxd = d;
        }
      const double us = time_it(20, [&] { tr_gemm_bf16_ar(A, W, bias, o_new, sh.M, sh.N, sh.K, sh.epi, nullptr); });
      printf(" ar %7.1f us %6.0f TF %s", us, fl / us * 1e-6, bad ? "BAD" : "ok");
      if (bad) printf("(%zu, first row %zu col %zu, max %.3g)", bad, first / sh.N, first % sh.N, maxd);
      best = us;
    }
    printf(" | %.2fx\n", us_old / best);    if (sh.M > 10000) { tot_old += us_old; tot_new += best; fl_tot += fl; }
    hipFree(A); hipFree(W); hipFree(bias); hipFree(o_old); hipFree(o_new);
  }
  printf("DeiT-S K=384 shapes: pc %.0f us (%.0f TF), ar %.0f us (%.0f TF)\n", tot_old, fl_tot / tot_old * 1e-6, tot_new, fl_tot / tot_new * 1e-6);
  return 0;
}
