// glowk device code, part 3: the BASIS Langevin update (run_basis_sep.py:131-181, dB branch, two sources) as ONE elementwise
// kernel with its own counter-based RNG.  HBM-bound and tiny next to the two log_prob_grad calls of a step; what matters at
// the reference's 30 tiles is that it is one launch instead of the dozen elementwise/reduction launches of a tensor library.
//
//   eps_k  = sqrt(2 eta) N(0, I)                                                 :163-164
//   mix    = 10/ln10 (logsumexp_k(x_k ln10/10) - ln 2)                            :133-141  (g, sum in power)
//   m_k    = softmax_k(x_k ln10/10)                                               :143-147  (grad_g)
//   x_k   <- x_k + eta (grad_logprob_k + lambda m_k (mixed - mix)) + eps_k        :180-181
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1) -> 4 x 32 random bits.  Stateless, so element e of
// step t of stream w always gets the same draw whatever the grid: counter = (e / 4, t, w, 0), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// four standard normals for elements 4 q .. 4 q + 3 of (seed, step, which): two Box-Muller pairs
__device__ __forceinline__ void normal4(uint64_t seed, uint64_t step, uint32_t which, uint64_t q, float (&z)[4]) {
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32) ^ (which << 28), (uint32_t)step, (uint32_t)(step >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float u1 = ((float)(c[2 * p] >> 8) + 0.5f) * (1.0f / 16777216.0f);        // (0, 1): 24 bits, never 0
    const float u2 = ((float)(c[2 * p + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.28318530717958647692f * u2, &sn, &cs);
    z[2 * p] = r * cs;
    z[2 * p + 1] = r * sn;
  }
}

struct BasisArgs {
  float* x1;            // [n] in place
  float* x2;
  const float* g1;      // [n] grad log p_1(x1), grad log p_2(x2) (compute_grad_logprob, :174-175)
  const float* g2;
  const float* mixed;   // [n] the observed mixture
  const float* eps1;    // optional [n] standard-normal draws supplied by the caller (tests replay the oracle's); null: device RNG
  const float* eps2;
  size_t n;
  float eta, lambda_recon, noise_scale;   // noise_scale = sqrt(2 eta)
  uint64_t seed, step;
  uint64_t q0;          // index of x[0] in the logical noise stream, in groups of four elements (shards: the global element offset / 4)
  int* nonfinite;       // optional: set to 1 if a gradient, a mixing term or an updated value is not finite (the reference's asserts, :183-191)
};

__device__ __forceinline__ bool basis_bad(float v) { return !(fabsf(v) <= 3.0e38f); }

// one thread = four consecutive elements (one Philox call per source)
__global__ __launch_bounds__(256) void k_basis_update(BasisArgs a) {
  const uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const size_t e0 = (size_t)q * 4;
  if (e0 >= a.n) return;
  float z1[4], z2[4];
  if (!a.eps1) normal4(a.seed, a.step, 0u, a.q0 + q, z1);
  if (!a.eps2) normal4(a.seed, a.step, 1u, a.q0 + q, z2);
  const float L10 = 0.23025850929940457f;   // ln 10 / 10
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t e = e0 + j;
    if (e >= a.n) break;
    const float x1 = a.x1[e], x2 = a.x2[e], g1 = a.g1[e], g2 = a.g2[e];
    const float n1 = a.eps1 ? a.eps1[e] : z1[j], n2 = a.eps2 ? a.eps2[e] : z2[j];
    const float s1 = x1 * L10, s2 = x2 * L10;
    const float mx = fmaxf(s1, s2);
    const float e1 = expf(s1 - mx), e2 = expf(s2 - mx);
    const float den = e1 + e2;
    const float mix = (1.0f / L10) * (mx + logf(den) - 0.69314718055994531f);
    const float m1 = e1 / den, m2 = e2 / den;
    const float r = a.lambda_recon * (a.mixed[e] - mix);
    const float y1 = x1 + a.eta * (g1 + m1 * r) + a.noise_scale * n1;
    const float y2 = x2 + a.eta * (g2 + m2 * r) + a.noise_scale * n2;
    bad |= basis_bad(g1) | basis_bad(g2) | basis_bad(mix) | basis_bad(y1) | basis_bad(y2);
    a.x1[e] = y1;
    a.x2[e] = y2;
  }
  if (bad && a.nonfinite) *a.nonfinite = 1;
}

// g(x1, x2) alone (the mixture of two sources in dB, sum in power)
__global__ __launch_bounds__(256) void k_basis_mix(const float* __restrict__ x1, const float* __restrict__ x2, float* __restrict__ out, size_t n) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const float L10 = 0.23025850929940457f;
  const float s1 = x1[e] * L10, s2 = x2[e] * L10;
  const float mx = fmaxf(s1, s2);
  out[e] = (1.0f / L10) * (mx + logf(expf(s1 - mx) + expf(s2 - mx)) - 0.69314718055994531f);
}

// the standard-normal draws k_basis_update makes for (seed, step, which); also the engine's general device RNG
// (uniform = 1: U(0, 1) instead -- the reference starts the chain from uniform noise, run_basis_sep.py:360-361)
__global__ __launch_bounds__(256) void k_basis_noise(float* __restrict__ out, size_t n, uint64_t seed, uint64_t step, uint32_t which, int uniform,
                                                    uint64_t q0) {
  const uint64_t ql = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const size_t e0 = (size_t)ql * 4;
  if (e0 >= n) return;
  const uint64_t q = q0 + ql;     // position in the logical stream: a shard draws what the whole batch would have drawn for its elements
  float z[4];
  if (uniform) {
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32) ^ (which << 28), (uint32_t)step, (uint32_t)(step >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int j = 0; j < 4; ++j) z[j] = ((float)(c[j] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  } else {
    normal4(seed, step, which, q, z);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (e0 + j < n) out[e0 + j] = z[j];
}

// out = x + sigma N(0, I) with the draws of (seed, step, which) from element 4 q0 on: the noise of train_noisy_glow.py:31 (the
// noise-conditioned priors of BASIS are trained on X + tf.random.normal(X.shape) * noise) without a tensor-library kernel
__global__ __launch_bounds__(256) void k_add_noise(const float* __restrict__ x, float* __restrict__ out, size_t n, float sigma, uint64_t seed,
                                                  uint64_t step, uint32_t which, uint64_t q0) {
  const uint64_t ql = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const size_t e0 = (size_t)ql * 4;
  if (e0 >= n) return;
  float z[4];
  normal4(seed, step, which, q0 + ql, z);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (e0 + j < n) out[e0 + j] = fmaf(sigma, z[j], x[e0 + j]);
}
