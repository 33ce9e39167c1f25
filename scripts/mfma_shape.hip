// Diagnostic (not part of the product): sustained f16 MFMA rate of the two shapes under this chip's power management,
// with the operand pattern of k_net_h3 (A fragments re-read from LDS, B in registers, 2 waves per SIMD, random data).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_shape.hip -o scripts/mfma_shape.bin && ./scripts/mfma_shape.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k(const float4* src, float* out, int iters) {
  __shared__ float4 lds[2048 * 4];   // 128 KiB of A fragments
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048 * 4; i += 512) lds[i] = src[i];
  __syncthreads();
  const h8* buf = reinterpret_cast<const h8*>(lds) + lane;
  h8 b0, b1, b2, b3;
  for (int j = 0; j < 8; ++j) {
    b0[j] = (_Float16)(0.01f * ((lane * 7 + j * 13) % 97 - 48));
    b1[j] = (_Float16)(0.01f * ((lane * 11 + j * 5) % 89 - 44));
    b2[j] = (_Float16)(0.01f * ((lane * 3 + j * 17) % 83 - 41));
    b3[j] = (_Float16)(0.01f * ((lane * 5 + j * 19) % 79 - 39));
  }
  if (SHAPE == 32) {
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {          // per tile: hi, lo fragments (2 KiB), 2 k-steps -> 6 MFMAs like h3_Y
        const h8 ah = buf[((it & 3) * 32 + t * 4 + 0) * 64], al = buf[((it & 3) * 32 + t * 4 + 1) * 64];
        const h8 ch = buf[((it & 3) * 32 + t * 4 + 2) * 64], cl = buf[((it & 3) * 32 + t * 4 + 3) * 64];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, b2, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, b3, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, b2, acc[t], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + tid] = s;
  } else {
    f32x4 acc[32];                           // 16 row blocks x 2 pixel halves
    for (int t = 0; t < 32; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {         // per 16-row block: hi, lo fragments of ONE 32-wide k-step (2 KiB per 2 blocks)
        const h8 ah = buf[((it & 3) * 32 + t * 2 + 0) * 64], al = buf[((it & 3) * 32 + t * 2 + 1) * 64];
        acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, b0, acc[2 * t], 0, 0, 0);
        acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, b2, acc[2 * t + 1], 0, 0, 0);
        acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b1, acc[2 * t], 0, 0, 0);
        acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b3, acc[2 * t + 1], 0, 0, 0);
        acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b0, acc[2 * t], 0, 0, 0);
        acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b2, acc[2 * t + 1], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int t = 0; t < 32; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + tid] = s;
  }
}

// fp32-input MFMA (v_mfma_f32_32x32x2_f32), operand pattern of k_net_f32: A from LDS as float4 (4 k-steps per read), B in registers
__global__ __launch_bounds__(256, 1) void kf32(const float4* src, float* out, int iters) {
  __shared__ float4 lds[2048 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048 * 4; i += 256) lds[i] = src[i];
  __syncthreads();
  float b[16];
  for (int j = 0; j < 16; ++j) b[j] = 0.01f * ((lane * 7 + j * 13) % 97 - 48);
  f32x16 acc[16];
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 w = lds[(((it & 7) * 16 + r * 4 + g) * 64 + lane)];
        acc[4 * g + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, b[r], acc[4 * g + 0], 0, 0, 0);
        acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, b[r + 4], acc[4 * g + 1], 0, 0, 0);
        acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, b[r + 8], acc[4 * g + 2], 0, 0, 0);
        acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, b[r + 12], acc[4 * g + 3], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + tid] = s;
}

int main() {
  const int n4 = 2048 * 4;
  std::vector<_Float16> h((size_t)n4 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) * 0.37f);
  float4* src; float* out;
  hipMalloc(&src, n4 * 16); hipMalloc(&out, 1024 * 512 * 4);
  hipMemcpy(src, h.data(), n4 * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 1024;
  for (int rep = 0; rep < 3; ++rep)
    for (int shape : {32, 16}) {
      for (int w = 0; w < 2; ++w) { if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(grid), dim3(512), 0, 0, src, out, iters); else hipLaunchKernelGGL(k<16>, dim3(grid), dim3(512), 0, 0, src, out, iters); }
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int L = 6;
      for (int w = 0; w < L; ++w) { if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(grid), dim3(512), 0, 0, src, out, iters); else hipLaunchKernelGGL(k<16>, dim3(grid), dim3(512), 0, 0, src, out, iters); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)L * grid * 8 /*waves*/ * iters * 48.0 * 32768.0;   // 48 x (32x32x16) or 96 x (16x16x32) per iteration
      printf("shape %dx%d: %.2f ms  %.1f TFLOP/s (f16 MFMA, LDS-fed)\n", shape, shape, ms, flops / ms * 1e-9);
    }
  for (int rep = 0; rep < 3; ++rep) {
    const int it32 = 1500;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kf32, dim3(grid), dim3(256), 0, 0, src, out, it32);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int L = 6;
    for (int w = 0; w < L; ++w) hipLaunchKernelGGL(kf32, dim3(grid), dim3(256), 0, 0, src, out, it32);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)L * grid * 4 /*waves*/ * it32 * 64.0 * 4096.0;   // 64 x (32x32x2) per iteration
    printf("fp32 32x32x2: %.2f ms  %.1f TFLOP/s (fp32 MFMA, LDS-fed, 1 wave per SIMD)\n", ms, flops / ms * 1e-9);
  }
  return 0;
}
