// Diagnostic (not part of the product): the weight-gradient GEMMs of csrc/glowk_train.h alone, on random planar operands of a
// level-0 shape, with a row stride (ld) that can differ from K -- what bounds them?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iaudiosourcesep_amd/csrc scripts/wgrad_bench.hip -o scripts/wgrad_bench.bin
//   ./scripts/wgrad_bench.bin [M N K S]
#include "glowk_train.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 512, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 32768;
  int S = argc > 4 ? atoi(argv[4]) : 24;
  const int batch = argc > 5 ? atoi(argv[5]) : 1;
  const float sb = argc > 6 ? (float)atof(argv[6]) : 4.0f;
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  unsigned st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hA) v = rnd() * 8.0f;
  for (auto& v : hB) v = rnd() * 3.0f;
  float *A, *B, *Cp, *C;
  CK(hipMalloc(&A, hA.size() * 4 * batch)); CK(hipMalloc(&B, hB.size() * 4 * batch));
  const int kslice = (((K + S - 1) / S) + 31) / 32 * 32;
  S = (K + kslice - 1) / kslice;
  CK(hipMalloc(&Cp, (size_t)batch * S * M * N * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
  for (int b = 0; b < batch; ++b) {
    CK(hipMemcpy(A + (size_t)b * hA.size(), hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B + (size_t)b * hB.size(), hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 2.0 * M * N * (double)K * batch;
  auto report = [&](const char* name, float ms, int reps) { printf("%-28s %8.1f us  %7.1f TFLOP/s (fp32-equivalent)\n", name, ms / reps * 1e3, flop / (ms / reps * 1e-3) / 1e12); };
  {
    WgradSplitArgs a; a.A = A; a.B = B; a.M = M; a.N = N; a.K = K; a.kslice = kslice; a.S = S; a.bsA = hA.size(); a.bsB = hB.size(); a.sa = 1.0f; a.sb = sb; a.Cpart = Cp; a.csz = (size_t)M * N; a.b_sums = 0; a.plain = getenv("PLAIN") ? 1 : 0;
    a.tm = (M + 127) / 128; a.tn = (N + (N >= 256 ? 127 : 63)) / (N >= 256 ? 128 : 64);
    const bool big = N >= 256;
    const bool w8 = getenv("W8") != nullptr;      // 8 waves, 256 x 128 tiles
    const int TM = (big && w8) ? 256 : 128, TN = big ? 128 : 64;
    a.tm = (M + TM - 1) / TM;
    dim3 grid(((M + TM - 1) / TM) * ((N + TN - 1) / TN) * S * batch);
    for (int it = 0; it < 3; ++it) { if (big && w8) hipLaunchKernelGGL((k_wgrad_h3<2, 2, 4, 2, true>), grid, dim3(512), 0, 0, a); else if (big) hipLaunchKernelGGL((k_wgrad_h3<2, 2, 2, 2, true>), grid, dim3(256), 0, 0, a); else hipLaunchKernelGGL((k_wgrad_h3<1, 2, 4, 1, true>), grid, dim3(256), 0, 0, a); }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < 20; ++it) { if (big && w8) hipLaunchKernelGGL((k_wgrad_h3<2, 2, 4, 2, true>), grid, dim3(512), 0, 0, a); else if (big) hipLaunchKernelGGL((k_wgrad_h3<2, 2, 2, 2, true>), grid, dim3(256), 0, 0, a); else hipLaunchKernelGGL((k_wgrad_h3<1, 2, 4, 1, true>), grid, dim3(256), 0, 0, a); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("M %d N %d K %d S %d batch %d grid %d x %d x %d\n", M, N, K, S, batch, grid.x, grid.y, grid.z);
    report("k_wgrad_h3 (fp16 split)", ms, 20);
    // spot check against fp64 on the host (batch entry 0)
    hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, 0, (const float*)Cp, S, (size_t)M * N, C, (size_t)0);
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int t = 0; t < 64; ++t) {
      const int m = (t * 37) % M, n = (t * 101) % N;
      double r = 0;
      for (int k = 0; k < K; ++k) r += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
      worst = std::fmax(worst, std::fabs(r - hC[(size_t)m * N + n])); scale = std::fmax(scale, std::fabs(r));
    }
    printf("   max |C - fp64| / max |C| over 64 entries: %.2e\n", worst / scale);
  }
  if (batch == 1) {
    WgradArgs a; a.A = A; a.B = B; a.M = M; a.N = N; a.a_ones = 0; a.K = K; a.kslice = kslice; a.Cpart = Cp; a.S = S; a.bsA = 0; a.bsB = 0; a.csz = (size_t)M * N;
    for (int wt = 1; wt <= 2; ++wt) {
      a.tm = (M + 64 * wt - 1) / (64 * wt); a.tn = (N + 64 * wt - 1) / (64 * wt);
      dim3 grid(a.tm * a.tn * S);
      auto go = [&]() { if (wt == 1) hipLaunchKernelGGL((k_wgrad_nt<1, true>), grid, dim3(256), 0, 0, a); else hipLaunchKernelGGL((k_wgrad_nt<2, true>), grid, dim3(256), 0, 0, a); };
      for (int it = 0; it < 3; ++it) go();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int it = 0; it < 20; ++it) go();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      report(wt == 1 ? "k_wgrad_nt<1> (fp32 MFMA)" : "k_wgrad_nt<2> (fp32 MFMA)", ms, 20);
    }
  }
  return 0;
}
