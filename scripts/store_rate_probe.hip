// What does a CU's vector-store path take?  Every wave streams stores into its own small, L2-resident region (so HBM is not in the
// picture): dword / dwordx2 / dwordx4 per lane, rows of 128 B (16 lanes x 8 B: the training sweep's pair stores) or fully contiguous,
// 4 or 8 waves per CU.  Prints bytes per clock and CU.      hipcc --offload-arch=gfx950 -O3 store_rate_probe.hip -o store_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int W, bool ROWS, bool NT>
__global__ __launch_bounds__(256) void k_store(float* buf, int iters, size_t per_wave_floats, int windows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* base = buf + ((size_t)blockIdx.x * 4 + wave) * per_wave_floats;
  // ROWS: lane (kq, n) writes W floats at row kq (rows 1 KB apart within the wave's region), columns W n .. ; else contiguous W floats per lane
  const unsigned off0 = ROWS ? ((lane >> 4) * 256 + (lane & 15) * W) * 4u : lane * W * 4u;
  const unsigned long long b0 = reinterpret_cast<unsigned long long>(base);
  const unsigned long long b = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(b0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)b0);
  f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
  for (int it = 0; it < iters; ++it) {
    unsigned off = off0 + (unsigned)((it & (windows - 1)) * 4096);      // 4-KB windows per wave (1: the same lines over and over: L2 only)
    if constexpr (W == 1 && NT) asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(off), "v"(v.x), "s"(b) : "memory");
    if constexpr (W == 2 && NT) { f32x2 p = {v.x, v.y}; asm volatile("global_store_dwordx2 %0, %1, %2 nt" ::"v"(off), "v"(p), "s"(b) : "memory"); }
    if constexpr (W == 4 && NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(off), "v"(v), "s"(b) : "memory");
    if constexpr (W == 1 && !NT) asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(v.x), "s"(b) : "memory");
    if constexpr (W == 2 && !NT) { f32x2 p = {v.x, v.y}; asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(p), "s"(b) : "memory"); }
    if constexpr (W == 4 && !NT) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(off), "v"(v), "s"(b) : "memory");
    v.x += 1.0f;
  }
}

template <int W, bool ROWS, bool NT>
int run(const char* name, float* buf, int wgs_per_cu, int cus, size_t per_wave, int windows) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_store<W, ROWS, NT>), dim3(cus * wgs_per_cu), dim3(256), 0, 0, buf, 200, per_wave, windows);
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_store<W, ROWS, NT>), dim3(cus * wgs_per_cu), dim3(256), 0, 0, buf, iters, per_wave, windows);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)cus * wgs_per_cu * 4 * iters * 64 * W * 4;
  int clk = 0; CHK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
  printf("%-26s %s, %d window(s) of 4 KB per wave, %d waves/CU: %7.2f TB/s  = %5.1f B/clk/CU at %.2f GHz, %5.1f clk per wave-instruction and CU\n", name, NT ? "nt   " : "plain", windows, 4 * wgs_per_cu, bytes / ms / 1e9,
         bytes / ms / 1e-3 / cus / (clk * 1e3), clk * 1e-6, (double)ms * 1e-3 * clk * 1e3 / ((double)wgs_per_cu * 4 * iters));
  return 0;
}

int main() {
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const size_t per_wave = 8192;      // floats: 32 KB
  float* buf; CHK(hipMalloc(&buf, (size_t)cus * 2 * 4 * per_wave * 4));
  for (int windows = 1; windows <= 8; windows *= 8)
    for (int w = 1; w <= 2; ++w) {
      if (run<1, true, true>("dword, 64-B row segments", buf, w, cus, per_wave, windows)) return 1;
      if (run<2, true, true>("dwordx2, 128-B rows", buf, w, cus, per_wave, windows)) return 1;
      if (run<4, true, true>("dwordx4, 256-B rows", buf, w, cus, per_wave, windows)) return 1;
      if (run<4, false, true>("dwordx4, contiguous", buf, w, cus, per_wave, windows)) return 1;
      if (run<1, true, false>("dword, 64-B row segments", buf, w, cus, per_wave, windows)) return 1;
      if (run<2, true, false>("dwordx2, 128-B rows", buf, w, cus, per_wave, windows)) return 1;
      if (run<4, true, false>("dwordx4, 256-B rows", buf, w, cus, per_wave, windows)) return 1;
      if (run<4, false, false>("dwordx4, contiguous", buf, w, cus, per_wave, windows)) return 1;
    }
  return 0;
}
