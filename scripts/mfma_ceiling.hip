// Diagnostic microbenchmark (not part of the product): what does a 1-wave-per-SIMD stream of
// v_mfma_f32_32x32x2_f32 on 16 accumulator tiles sustain on this chip, (A) operands in registers,
// (B) A operand re-read from LDS with ds_read_b128 exactly like k_net's conv2 block?
// build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_ceiling.hip -o /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, int iters, unsigned long long* clk) {
  extern __shared__ float4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) lds4[i] = make_float4(in[i & 1023], in[(i + 1) & 1023], in[(i + 2) & 1023], in[(i + 3) & 1023]);
  __syncthreads();
  f32x16 acc[16];
  for (int f = 0; f < 16; ++f) for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
  f32x16 h;
  for (int r = 0; r < 16; ++r) h[r] = in[(lane + r) & 1023];
  float a0 = in[lane];
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 wv;
        if (MODE == 0) { wv = make_float4(a0, a0, a0, a0); }
        else { wv = lds4[((r * 4 + g) * 64 + lane)]; }
        acc[4 * g + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, h[r], acc[4 * g + 0], 0, 0, 0);
        acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, h[r], acc[4 * g + 1], 0, 0, 0);
        acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, h[r], acc[4 * g + 2], 0, 0, 0);
        acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, h[r], acc[4 * g + 3], 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int f = 0; f < 16; ++f) for (int r = 0; r < 16; ++r) s += acc[f][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  const int grid = 256, iters = 2000;
  float *in, *out; unsigned long long* clk;
  hipMalloc(&in, 4096); hipMalloc(&out, grid * 256 * 4); hipMalloc(&clk, grid * 16);
  std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 65536, 0, in, out, iters, clk);
      else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 65536, 0, in, out, iters, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> c(2 * grid); hipMemcpy(c.data(), clk, grid * 16, hipMemcpyDeviceToHost);
      double flops = (double)grid * 4 * iters * 256.0 * (32 * 32 * 2 * 2);
      double mhz = (double)c[0] / ((double)c[1] / 100.0);   // s_memrealtime ticks at 100 MHz
      printf("mode %d rep %d: %.3f ms  %.1f TFLOP/s  cycles/MFMA %.2f  in-kernel clock %.0f MHz\n", mode, rep, ms, flops / ms / 1e9,
             (double)c[0] / (iters * 256.0), mhz);
    }
  }
  return 0;
}
