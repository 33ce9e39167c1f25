// glowk handle-free entry points: the BASIS update kernel and mixture, the Philox device RNG, CRC-32C (run_basis_sep.py:131-181, tile_io / tf_checkpoint)
#include "glowk_engine.h"
#include "glowk_basis.h"

using namespace glowk_eng;

extern "C" {

int glowk_basis_update(float* x1_dev, float* x2_dev, const float* g1_dev, const float* g2_dev, const float* mixed_dev, size_t n,
                       float eta, float lambda_recon, const float* eps1_dev, const float* eps2_dev, uint64_t seed, uint64_t step,
                       uint64_t offset, int* nonfinite_dev, void* stream) {
  if (!x1_dev || !x2_dev || !g1_dev || !g2_dev || !mixed_dev) return fail("null tensor");
  if (offset % 4) return fail("basis_update: the stream offset must be a multiple of 4 elements");
  if (n == 0) return 0;
  if (n > ((size_t)1 << 40)) return fail("basis_update: too many elements");
  if (!(eta >= 0.0f)) return fail("basis_update: eta must be non-negative");
  DeviceGuard dg(ptr_device(x1_dev));
  BasisArgs a;
  a.x1 = x1_dev; a.x2 = x2_dev; a.g1 = g1_dev; a.g2 = g2_dev; a.mixed = mixed_dev; a.eps1 = eps1_dev; a.eps2 = eps2_dev; a.n = n;
  a.eta = eta; a.lambda_recon = lambda_recon; a.noise_scale = std::sqrt(2.0f * eta); a.seed = seed; a.step = step; a.q0 = offset / 4; a.nonfinite = nonfinite_dev;
  const size_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(k_basis_update, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  LAUNCHCHK("k_basis_update");
  return 0;
}

int glowk_basis_mix(const float* x1_dev, const float* x2_dev, float* out_dev, size_t n, void* stream) {
  if (!x1_dev || !x2_dev || !out_dev) return fail("null tensor");
  if (n == 0) return 0;
  DeviceGuard dg(ptr_device(x1_dev));
  hipLaunchKernelGGL(k_basis_mix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x1_dev, x2_dev, out_dev, n);
  LAUNCHCHK("k_basis_mix");
  return 0;
}

int glowk_random(float* out_dev, size_t n, uint64_t seed, uint64_t step, int which, int uniform, uint64_t offset, void* stream) {
  if (!out_dev) return fail("null tensor");
  if (which < 0 || which > 15) return fail("random: stream id must be 0..15");
  if (offset % 4) return fail("random: the stream offset must be a multiple of 4 elements");
  if (n == 0) return 0;
  DeviceGuard dg(ptr_device(out_dev));
  const size_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(k_basis_noise, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out_dev, n, seed, step,
                     (uint32_t)which, uniform, offset / 4);
  LAUNCHCHK("k_basis_noise");
  return 0;
}

int glowk_add_noise(const float* x_dev, float* out_dev, size_t n, float sigma, uint64_t seed, uint64_t step, int which, uint64_t offset,
                    void* stream) {
  if (!x_dev || !out_dev) return fail("null tensor");
  if (which < 0 || which > 15) return fail("add_noise: stream id must be 0..15");
  if (offset % 4) return fail("add_noise: the stream offset must be a multiple of 4 elements");
  if (n == 0) return 0;
  DeviceGuard dg(ptr_device(out_dev));
  const size_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(k_add_noise, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_dev, out_dev, n, sigma, seed,
                     step, (uint32_t)which, offset / 4);
  LAUNCHCHK("k_add_noise");
  return 0;
}

namespace glowk_eng {
struct Crc32cTable {
  uint32_t t[8][256];
  Crc32cTable() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
  }
};
}  // namespace glowk_eng

uint32_t glowk_crc32c(const void* host_data, size_t n) {
  static const Crc32cTable tbl;            // function-local static: initialised once, thread-safe by the language (C++11 [stmt.dcl])
  const uint32_t (&table)[8][256] = tbl.t;
  const unsigned char* p = static_cast<const unsigned char*>(host_data);
  uint32_t c = 0xFFFFFFFFu;
  while (n >= 8) {                         // slicing-by-8
    uint32_t lo, hi;
    std::memcpy(&lo, p, 4);
    std::memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
        table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
    p += 8; n -= 8;
  }
  while (n--) c = table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}


}  // extern "C"
