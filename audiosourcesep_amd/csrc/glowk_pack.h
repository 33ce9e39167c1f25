// glowk host-side weight packing: the reference's tensors (HWIO conv kernels, [in, out] 1x1 factors, BatchNorm vectors) -> the
// images the kernels stream (MFMA A-operand lane order, fp16 hi/lo splits, folded per-pixel affines).  Host-only C++ (no HIP):
// glowk.hip includes it for the engine, tests/pack_sanitize_main.cpp compiles it alone under -fsanitize=address,undefined and
// -fsanitize=thread (tests/test_pack_sanitizers.py).
#pragma once
#include "../../include/glowk.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#ifndef GLOWK_ACT_SCALE
#define GLOWK_ACT_SCALE 4.0f   // activation scale before the fp16 split (glowk_kernels.h has the rationale)
#endif

#ifndef __HIP__
struct float4;   // (device vector type; only pointers to it appear here -- a host-only build has no HIP headers)
#endif

namespace {

// IEEE binary16 <-> binary32 in integer arithmetic, round to nearest even: bit-for-bit what v_cvt_f16_f32 / the compiler's
// _Float16 conversions give (host compilers older than GCC 12 have no _Float16 in C++, and the sanitizer builds use them)
inline uint16_t f32_to_f16(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
  const uint32_t em = x & 0x7FFFFFFFu;
  if (em >= 0x7F800000u) return (uint16_t)(sign | (em > 0x7F800000u ? 0x7E00u : 0x7C00u));   // NaN, inf
  if (em >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                   // >= 65520 rounds to inf
  if (em >= 0x38800000u) {                                                                    // normal range (>= 2^-14)
    const uint32_t m = em - 0x38000000u;                                                      // exponent bias 127 -> 15
    return (uint16_t)(sign | ((m + 0xFFFu + ((m >> 13) & 1u)) >> 13));
  }
  if (em < 0x33000000u) return sign;                                                          // < 2^-25 (and the tie at 2^-25) -> 0
  const uint32_t e = em >> 23, mant = (em & 0x7FFFFFu) | 0x800000u, shift = 126u - e;         // subnormal: n * 2^-24, shift in [14, 24]
  return (uint16_t)(sign | ((mant + (1u << (shift - 1)) - 1u + ((mant >> shift) & 1u)) >> shift));
}
inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
  uint32_t x;
  if (e == 0x1Fu) x = sign | 0x7F800000u | (m << 13);
  else if (e) x = sign | ((e + 112u) << 23) | (m << 13);
  else if (!m) x = sign;
  else {
    int s = 0;
    uint32_t mm = m;
    while (!(mm & 0x400u)) { mm <<= 1; ++s; }
    x = sign | ((uint32_t)(113 - s) << 23) | ((mm & 0x3FFu) << 13);
  }
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

}  // namespace

// (the two types that appear in signatures shared between the engine's translation units have external linkage)
namespace glowk_pk {

struct StepDev {            // device pointers into the arena
  const float* K1p = nullptr;
  const float* ep = nullptr;
  const float4* R0p = nullptr;
  const float4* RHp = nullptr;   // f16x3 ring image (null: shape not supported by k_net_h3)
  const float* epH = nullptr;    // its epilogue constants
  const float4* RSp = nullptr;   // forward image for the 16x16x32 kernel (k_net_h3s), or null
  const float4* RSBp = nullptr;  // backward image for it (only packed together with RSp: a level uses one kernel family)
  const float4* RHBp = nullptr;  // f16x3 image of the backward network (null: not supported: exact fp32 backward)
  float scb1 = 1.f, scb2 = 1.f, scb3 = 1.f;
  float sc1 = 1.f, sc2 = 1.f, sc3 = 1.f;
  float xlim_f = 0.f, xlim_b = 0.f;   // range guard of the split kernels: largest |network input| (forward: v_b; backward: g_o) for which no
                                      // hidden value can leave the fp16 range (L1 bounds of the weights; pack_step)
  const float* K3bp = nullptr;   // backward: conv3^T operands of the small-conv chain [NF][9c/2][64]
  const float4* RBp = nullptr;   // backward ring image: K2^T chunk fo | K3b operands of block fo+1 ; conv1^T chunks
  const float *Afwd = nullptr, *bfwd = nullptr, *Ainv = nullptr, *binv = nullptr, *b3 = nullptr;
  size_t arena_off = 0;     // offset (floats) of this step's block in the arena
};

struct Level {
  int h, w, c;              // tensor the steps of this block act on
  int z_off, z_width;       // channel slice of the latent this block contributes
  int Cz;                   // channels per latent pixel of its factored-out half (0 for the last block)
  std::vector<std::vector<float>> host[GLOWK_NUM_STEP_TENSORS];  // [tensor][step] -> values
  std::vector<StepDev> dev;
};

}  // namespace glowk_pk
using glowk_pk::Level;
using glowk_pk::StepDev;

namespace {


size_t step_tensor_size(const glowk_config& cfg, const Level& lv, int id) {
  const size_t c = lv.c, F = cfg.F;
  switch (id) {
    case GLOWK_ACTNORM_LOG_SCALE: case GLOWK_ACTNORM_SHIFT: case GLOWK_INV1X1_SIGN_S: case GLOWK_INV1X1_LOG_S:
    case GLOWK_CONV3_BIAS: return c;
    case GLOWK_INV1X1_P: case GLOWK_INV1X1_L: case GLOWK_INV1X1_U: case GLOWK_INV1X1_P_INV: return c * c;
    case GLOWK_CONV1_KERNEL: return 9 * (c / 2) * F;
    case GLOWK_CONV2_KERNEL: return F * F;
    case GLOWK_CONV3_KERNEL: return 9 * F * c;
    case GLOWK_CONV1_BIAS: case GLOWK_CONV2_BIAS:
    case GLOWK_BN1_GAMMA: case GLOWK_BN1_BETA: case GLOWK_BN1_MEAN: case GLOWK_BN1_VAR:
    case GLOWK_BN2_GAMMA: case GLOWK_BN2_BETA: case GLOWK_BN2_MEAN: case GLOWK_BN2_VAR: return F;
    default: return 0;
  }
}

// ---- small dense helpers (double) --------------------------------------------------------------
typedef std::vector<double> Mat;  // row-major c x c

Mat matmul(const Mat& a, const Mat& b, int c) {
  Mat r(c * c, 0.0);
  for (int i = 0; i < c; ++i)
    for (int k = 0; k < c; ++k) {
      const double aik = a[i * c + k];
      for (int j = 0; j < c; ++j) r[i * c + j] += aik * b[k * c + j];
    }
  return r;
}

bool invert(const Mat& m, int c, Mat& out) {  // Gauss-Jordan, partial pivoting
  Mat a = m;
  out.assign(c * c, 0.0);
  for (int i = 0; i < c; ++i) out[i * c + i] = 1.0;
  for (int col = 0; col < c; ++col) {
    int piv = col;
    for (int r = col + 1; r < c; ++r)
      if (std::fabs(a[r * c + col]) > std::fabs(a[piv * c + col])) piv = r;
    if (a[piv * c + col] == 0.0) return false;
    if (piv != col)
      for (int j = 0; j < c; ++j) {
        std::swap(a[piv * c + j], a[col * c + j]);
        std::swap(out[piv * c + j], out[col * c + j]);
      }
    const double d = 1.0 / a[col * c + col];
    for (int j = 0; j < c; ++j) { a[col * c + j] *= d; out[col * c + j] *= d; }
    for (int r = 0; r < c; ++r) {
      if (r == col) continue;
      const double f = a[r * c + col];
      if (f == 0.0) continue;
      for (int j = 0; j < c; ++j) { a[r * c + j] -= f * a[col * c + j]; out[r * c + j] -= f * out[col * c + j]; }
    }
  }
  return true;
}

inline int rho(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }
inline size_t pad4(size_t n) { return (n + 3) & ~size_t(3); }

struct StepLayout {
  size_t K1p, ep, R0p, K3bp, RBp, RHp, epH, RSp, RSBp, RHBp, Afwd, bfwd, Ainv, binv, b3, total;
  size_t slotH;        // floats per main chunk of the f16x3 image (0: shape not supported by k_net_h3)
  size_t slotHB;       // the same for the backward network's image
  size_t slotS;        // and for the forward image of the 16x16x32 kernel
  size_t slotSB;       // ... its backward image
  size_t slotB, k3fB;  // backward ring image: floats per slot; floats of its conv3^T-operand part (0: not in the ring)
  size_t slot0, k1f0;  // floats per slot of k_net_f32's ring image; floats of its conv1 part (0: not in the ring)
};

StepLayout step_layout(int c, int F) {
  const int CI = c / 2, NF = F / 32, KS1 = (9 * CI) / 2, NMT = (9 * c + 31) / 32;
  StepLayout L;
  size_t o = 0;
  L.K1p = o; o += pad4((size_t)NF * KS1 * 64);
  L.ep = o; o += pad4((size_t)6 * F);
  {
    size_t k1f = (size_t)(((size_t)KS1 * 256 + 1023) / 1024) * 256;                 // conv1 MFMA operands [KS1][64], 1-KiB pieces
    if (2 * ((size_t)NF * 1024 + k1f) * 4 + (size_t)6 * F * 4 > 160 * 1024) k1f = 0;  // Ring1::K1_IN_RING == false
    L.k1f0 = k1f;
    L.slot0 = (size_t)NF * 1024 + k1f;
    L.R0p = o; o += (size_t)(NF + NMT) * L.slot0;
  }
  {
    const int KS3 = (9 * c) / 2, NM1 = (9 * CI + 31) / 32;
    L.K3bp = o; o += pad4((size_t)NF * KS3 * 64);
    size_t k3f = (size_t)(((size_t)KS3 * 256 + 1023) / 1024) * 256;
    if (2 * ((size_t)NF * 1024 + k3f) * 4 + (size_t)6 * F * 4 > 160 * 1024) k3f = 0;   // Ring1<c, NF>::K1_IN_RING == false
    L.k3fB = k3f;
    L.slotB = (size_t)NF * 1024 + k3f;
    L.RBp = o; o += (size_t)(NF + NM1) * L.slotB;
  }
  {
    const int KS = (9 * CI + 1 + 15) / 16, NFH = NF / 2;                                              // RingH<CI, 18 CI, NF, fwd>
    const size_t ephn = pad4((size_t)F + 32 * NMT);
    const size_t lds = (size_t)3 * NFH * 4096 + (size_t)2 * KS * 2048 + ephn * 4;
    const bool fitsH = lds <= 160 * 1024 && NF % 4 == 0 && KS <= 5 && NMT <= 6;
    const int KSS = (9 * CI + 1 + 31) / 32, NMS = (18 * CI + 15) / 16, NCH = (NFH * NMS + 2 * NFH - 1) / (2 * NFH);     // RingS<CI, NF>
    const size_t ldss = (size_t)3 * NFH * 4096 + (size_t)2 * KSS * 4096 + pad4((size_t)F + 16 * NMS) * 4;
    // the 16x16x32 kernel alone (RingS<CI, 18 CI, NF, fwd, 4>::FITS without a RingH instance): c = 32 at n_filters = 512 --
    // 18 row blocks of P in three fused groups, K = 145 in five k-steps, four passes; forward direction only
    const bool fitsSonly = !fitsH && NF % 4 == 0 && NF / 4 >= 2 && KSS <= 5 && NMS <= 18 && NMS % 6 == 0 && ldss <= 160 * 1024;
    const bool fitsS = (fitsH && ldss <= 160 * 1024 && KSS <= 3 && NMS <= 12 && NCH >= 2) || fitsSonly;
    L.slotH = fitsH ? (size_t)NFH * 1024 : 0;
    L.RHp = o; o += fitsH ? (size_t)NF * KS * 512 + (size_t)2 * (NF + NMT) * NFH * 1024 : 0;
    L.epH = o; o += (fitsH || fitsS) ? ephn : 0;
    {
      L.slotS = fitsS ? (size_t)NFH * 1024 : 0;
      L.RSp = o; o += fitsS ? (size_t)NF * KSS * 1024 + (size_t)2 * (NF + NCH) * NFH * 1024 : 0;
      // RingS<c, 9 CI, NF, bwd>: K = 9c in k-steps of 32, 9 CI output rows in blocks of 16
      const int KSSB = (9 * c + 31) / 32, NMSB = (9 * CI + 15) / 16, NCHB = (NFH * NMSB + 2 * NFH - 1) / (2 * NFH);
      const bool fitsSB = fitsS && KSSB <= 9 && NMSB <= 12;     // (KSSB in 6..9: the half-wave form only, RingS<.., MODE | 32, 4>)
      L.slotSB = fitsSB ? (size_t)NFH * 1024 : 0;
      L.RSBp = o; o += fitsSB ? (size_t)NF * KSSB * 1024 + (size_t)2 * (NF + NCHB) * NFH * 1024 : 0;
    }
    const int KSB = (9 * c + 15) / 16, NMB = (9 * CI + 31) / 32;                                      // RingH<c, 9 CI, NF, bwd>
    const size_t ldsb = (size_t)3 * (NF / 4) * 4096 + (size_t)2 * KSB * 2048 + pad4((size_t)F + 32 * NMB) * 4 + (size_t)2 * NF * 1024;   // 4-pass form
    const bool fitsHB = ldsb <= 160 * 1024 && NF % 4 == 0 && KSB <= 9 && NMB <= 6;
    L.slotHB = fitsHB ? (size_t)NFH * 1024 : 0;
    L.RHBp = o; o += fitsHB ? (size_t)NF * KSB * 512 + (size_t)2 * (NF + NMB) * NFH * 1024 : 0;
  }
  L.Afwd = o; o += pad4((size_t)c * c);
  L.bfwd = o; o += pad4(c);
  L.Ainv = o; o += pad4((size_t)c * c);
  L.binv = o; o += pad4(c);
  L.b3 = o; o += pad4(c);
  L.total = o;
  return L;
}

// ActNorm + 1x1 of one step folded into per-pixel affines (forward and inverse), conv3 bias, log-det constant.
// dst is the step's block of the arena staging; returns false + message on a singular 1x1
bool pack_affine(const glowk_config& cfg, const Level& lv, int k, float* dst, double* ld_const_out, std::string* err) {
  const int c = lv.c, F = cfg.F;
  const StepLayout L = step_layout(c, F);
  auto T = [&](int id) -> const float* { return lv.host[id][k].data(); };

  // --- 1x1: W = P (L*mask + I) (U*mask^T + diag(sign*exp(log_S)))  (flow_tfp_bijectors.py:300-303) ---
  Mat Pm(c * c), Lm(c * c), Um(c * c);
  for (int i = 0; i < c; ++i)
    for (int j = 0; j < c; ++j) {
      Pm[i * c + j] = T(GLOWK_INV1X1_P)[i * c + j];
      Lm[i * c + j] = (i > j) ? T(GLOWK_INV1X1_L)[i * c + j] : (i == j ? 1.0 : 0.0);
      Um[i * c + j] = (i < j) ? T(GLOWK_INV1X1_U)[i * c + j]
                              : (i == j ? (double)T(GLOWK_INV1X1_SIGN_S)[i] * std::exp((double)T(GLOWK_INV1X1_LOG_S)[i]) : 0.0);
    }
  const Mat Wm = matmul(Pm, matmul(Lm, Um, c), c);
  Mat Pi, Li, Ui;
  if (!invert(Pm, c, Pi) || !invert(Lm, c, Li) || !invert(Um, c, Ui)) {
    *err = "singular 1x1 convolution factor";
    return false;
  }
  {
    // the reference's inverse multiplies by its stored variable P_inv (:313), not by inv(P): honour one that was loaded
    const float* pin = T(GLOWK_INV1X1_P_INV);
    bool set = false;
    for (int i = 0; i < c * c; ++i) set |= pin[i] != 0.0f;
    if (set)
      for (int i = 0; i < c * c; ++i) Pi[i] = pin[i];
  }
  const Mat Winv = matmul(Ui, matmul(Li, Pi, c), c);  // :309-315
  const float* ls = T(GLOWK_ACTNORM_LOG_SCALE);
  const float* sh = T(GLOWK_ACTNORM_SHIFT);
  double sum_ls = 0, sum_lS = 0;
  for (int i = 0; i < c; ++i) { sum_ls += ls[i]; sum_lS += T(GLOWK_INV1X1_LOG_S)[i]; }
  *ld_const_out = (double)lv.h * lv.w * (sum_ls + sum_lS);  // :250-253, :319-322
  for (int ci = 0; ci < c; ++ci)
    for (int co = 0; co < c; ++co) {
      dst[L.Afwd + ci * c + co] = (float)(std::exp((double)ls[ci]) * Wm[ci * c + co]);      // actnorm then 1x1
      dst[L.Ainv + ci * c + co] = (float)(Winv[ci * c + co] * std::exp(-(double)ls[co]));    // 1x1^-1 then actnorm^-1
    }
  for (int co = 0; co < c; ++co) {
    double s = 0;
    for (int ci = 0; ci < c; ++ci) s += (double)sh[ci] * Wm[ci * c + co];
    dst[L.bfwd + co] = (float)s;
    dst[L.binv + co] = (float)(-(double)sh[co] * std::exp(-(double)ls[co]));
    dst[L.b3 + co] = T(GLOWK_CONV3_BIAS)[co];
  }
  return true;
}

// pack one step into dst (host staging of the arena)
// The f16 images as a MAP for the device-side re-pack of the training step (glowk_train.h: k_repack_f16): every half of every
// image is one element of six scaled fp32 source arrays (or zero), scaled by its layer's power of two and split hi / lo.
// pack_step(..., f16_map) runs the very loops that write the images, with the source arrays holding 1-based CODES instead of
// values: code ranges [K1f | K2f | K3f | W3b | W2b | K1 raw], bit 30 = lo half; positions no image loop writes stay -1.
struct F16Codes { size_t A, B, C, D, E, G, total; };
inline F16Codes f16_code_bases(int c, int F) {
  const size_t CI = c / 2;
  F16Codes q;
  q.A = 0; q.B = q.A + (9 * CI + 1) * F; q.C = q.B + (size_t)F * F; q.D = q.C + (size_t)9 * F * c; q.E = q.D + (size_t)9 * c * F;
  q.G = q.E + (size_t)F * F; q.total = q.G + 9 * CI * F;
  return q;
}

bool pack_step(const glowk_config& cfg, const Level& lv, int k, float* dst, double* ld_const_out, float* scales3 /* [8]: fwd, bwd scales; fwd, bwd input limits */,
               std::string* err, int* f16_map = nullptr) {
  const int c = lv.c, F = cfg.F, CI = c / 2, CO = c, NF = F / 32, KS1 = (9 * CI) / 2, NMT = (9 * c + 31) / 32;
  const StepLayout L = step_layout(c, F);
  auto T = [&](int id) -> const float* { return lv.host[id][k].data(); };
  if (!pack_affine(cfg, lv, k, dst, ld_const_out, err)) return false;

  // --- BN (inference) folded to y = g*x + d, applied after bias+ReLU (flow_tfk_layers.py:75-78) ---
  const int bn[2][4] = {{GLOWK_BN1_GAMMA, GLOWK_BN1_BETA, GLOWK_BN1_MEAN, GLOWK_BN1_VAR},
                        {GLOWK_BN2_GAMMA, GLOWK_BN2_BETA, GLOWK_BN2_MEAN, GLOWK_BN2_VAR}};
  const int bias_id[2] = {GLOWK_CONV1_BIAS, GLOWK_CONV2_BIAS};
  for (int s = 0; s < 2; ++s)
    for (int f = 0; f < F; ++f) {
      const double g = (double)T(bn[s][0])[f] / std::sqrt((double)T(bn[s][3])[f] + (double)cfg.bn_eps);
      const double d = (double)T(bn[s][1])[f] - (double)T(bn[s][2])[f] * g;
      dst[L.ep + (3 * s + 0) * F + f] = T(bias_id[s])[f];
      dst[L.ep + (3 * s + 1) * F + f] = (float)g;
      dst[L.ep + (3 * s + 2) * F + f] = (float)d;
    }

  // --- conv kernels in MFMA A-operand lane order (see glowk_kernels.h) ---
  const float* K1 = T(GLOWK_CONV1_KERNEL);  // [9*CI][F]
  for (int fi = 0; fi < NF; ++fi)
    for (int ks = 0; ks < KS1; ++ks)
      for (int l = 0; l < 64; ++l) {
        const int i = l & 31, hh = l >> 5, kk = 2 * ks + hh;
        dst[L.K1p + ((size_t)fi * KS1 + ks) * 64 + l] = (kk < 9 * CI) ? K1[(size_t)kk * F + fi * 32 + i] : 0.0f;
      }
  const float* K2 = T(GLOWK_CONV2_KERNEL);  // [f_in][f_out]
  for (int fi = 0; fi < NF; ++fi)
    for (int r = 0; r < 16; ++r)
      for (int g = 0; g < NF / 4; ++g)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 4; ++e) {
            const int i = l & 31, hh = l >> 5, fo = 4 * g + e;
            const size_t idx = (((size_t)r * (NF / 4) + g) * 64 + l) * 4 + e;             // within ring slot fi (main part)
            dst[L.R0p + (size_t)fi * L.slot0 + idx] = K2[(size_t)(fi * 32 + rho(r, hh)) * F + fo * 32 + i];
          }
  const float* K3 = T(GLOWK_CONV3_KERNEL);  // [tap][f][co]
  for (int mt = 0; mt < NMT; ++mt)
    for (int fo = 0; fo < NF; ++fo)
      for (int r4 = 0; r4 < 4; ++r4)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 4; ++e) {
            const int i = l & 31, hh = l >> 5, r = 4 * r4 + e;
            const int m = mt * 32 + i, f = fo * 32 + rho(r, hh);
            const size_t idx = ((((size_t)fo) * 4 + r4) * 64 + l) * 4 + e;                  // within ring slot NF + mt
            float v = 0.0f;
            if (m < 9 * CO) {
              const int tap = m / CO, co = m % CO;
              v = K3[((size_t)tap * F + f) * CO + co];
            }
            dst[L.R0p + (size_t)(NF + mt) * L.slot0 + idx] = v;
          }
  // (the K2 / K3 loops above wrote the main parts of k_net_f32's ring image (Ring1): slot c < NF = K2 chunk c, slot NF+mt = K3
  // chunk mt);  the conv1 MFMA operands of hidden block c+1 ride behind the main part of slot c
  if (L.k1f0) {
    const size_t mainf = (size_t)NF * 1024, k1n = (size_t)KS1 * 64;
    for (int c2 = 0; c2 < NF + NMT; ++c2) {
      const int k1blk = (c2 < NF) ? (c2 + 1) % NF : 0;
      std::memcpy(dst + L.R0p + (size_t)c2 * L.slot0 + mainf, dst + L.K1p + (size_t)k1blk * k1n, k1n * 4);
    }
  }
  // ---- f16x3 image (k_net_h3): weights scaled by a power of two, split hi/lo in fp16, A operands in fragment order ----
  for (int i = 0; i < 6; ++i) scales3[i] = 1.0f;
  scales3[6] = scales3[7] = 0.0f;
  // values about to be split must stay below the fp16 maximum after the activation scale (a little margin for rounding)
  const double RANGE_LIM = 60000.0 / (double)GLOWK_ACT_SCALE;
  auto pow2_scale = [](const float* w, size_t n) {
    float m = 0.0f;
    for (size_t i = 0; i < n; ++i) m = std::fmax(m, std::fabs(w[i]));
    int e = 0;
    if (m > 0.0f) { std::frexp(m, &e); }          // m = f * 2^e, f in [0.5, 1)
    return 14 - e;                                // |w| * 2^S < 2^14: hi well inside fp16, lo ~2^3 (normal)
  };
  // one A-operand element: scaled, split, stored as half j of this lane's 16 bytes in the hi row and in the lo row after it
  const F16Codes QC = f16_code_bases(c, F);
  auto put = [dst, f16_map](float* row_lane, int j, int hl, float w, int S) {
    if (f16_map) {      // map mode: w is a code (0 = a zero element)
      f16_map[(size_t)(row_lane - dst) * 2 + (size_t)hl * 512 + j] = (int)w | (hl << 30);
      return;
    }
    const float ws = std::ldexp(w, S);
    const uint16_t hi = f32_to_f16(ws);
    const uint16_t lo = f32_to_f16(ws - f16_to_f32(hi));
    uint16_t* dsth = reinterpret_cast<uint16_t*>(row_lane) + (size_t)hl * 64 * 8;
    dsth[j] = hl ? lo : hi;
  };
  if (L.slotH || L.slotS) {
    const int KS = (9 * CI + 1 + 15) / 16;
    // Every per-channel constant of the epilogues is folded into the weights (host, fp64):
    //  * BatchNorm y = g*r + d with g = m * 2^e (|m| in [0.5,1)): the power of two scales the channel's own producer
    //    (a row scale: exact), the mantissa m multiplies the consumer's weight column, and W.d joins the consumer's bias
    //    (conv2: its bias; conv3: one constant per (tap, channel) row of P -- each row of P is a 1x1 of the pixel's own h2,
    //    so there are no border terms);
    //  * conv1's bias rides in a padding row of its 16-wide k-steps (the B fragment holds a constant 1 there);
    //  * conv2's bias is the initial value of its accumulators.
    // What is left in the kernel is  B = split(max(acc * 2^-S, 0))  with one uniform power of two per layer.
    std::vector<float> K1f((size_t)(9 * CI + 1) * F), K2f((size_t)F * F), K3f((size_t)9 * F * CO);
    std::vector<double> b2f(F), pbf((size_t)9 * CO, 0.0);
    std::vector<int> e1(F), e2(F);
    {
      const float* ep = dst + L.ep;     // [b1 | g1 | d1 | b2 | g2 | d2] as packed above
      std::vector<double> m1(F), m2(F);
      for (int f = 0; f < F; ++f) {
        int e;
        m1[f] = std::frexp((double)ep[F + f], &e);      e1[f] = ep[F + f] != 0.0f ? e : 0;
        m2[f] = std::frexp((double)ep[4 * F + f], &e);  e2[f] = ep[4 * F + f] != 0.0f ? e : 0;
      }
      for (int kk = 0; kk <= 9 * CI; ++kk)
        for (int f = 0; f < F; ++f)
          K1f[(size_t)kk * F + f] = std::ldexp(kk < 9 * CI ? K1[(size_t)kk * F + f] : ep[f], e1[f]);
      for (int fo = 0; fo < F; ++fo) b2f[fo] = (double)ep[3 * F + fo];
      for (int fi = 0; fi < F; ++fi)
        for (int fo = 0; fo < F; ++fo) {
          const double w = (double)K2[(size_t)fi * F + fo];
          K2f[(size_t)fi * F + fo] = (float)std::ldexp(w * m1[fi], e2[fo]);
          b2f[fo] += w * (double)ep[2 * F + fi];
        }
      for (int tap = 0; tap < 9; ++tap)
        for (int f = 0; f < F; ++f)
          for (int co = 0; co < CO; ++co) {
            const double w = (double)K3[((size_t)tap * F + f) * CO + co];
            K3f[((size_t)tap * F + f) * CO + co] = (float)(w * m2[f]);
            pbf[(size_t)tap * CO + co] += w * (double)ep[5 * F + f];
          }
    }
    {
      // ---- range guard, forward network (glowk_kernels.h: range8): with |input| <= X the value split after conv1 is at most
      //      2^e1[f] (sum_k |K1[k][f]| X + |b1[f]|), the one split after conv2 at most
      //      2^e2[f'] (sum_f |K2[f][f']| (|g1[f]| r1max[f] + |d1[f]|) + |b2[f']|): the largest X that keeps all of them in range
      const float* ep = dst + L.ep;     // [b1 | g1 | d1 | b2 | g2 | d2]
      std::vector<double> n1(F, 0.0);   // sum_k |K1[k][f]|
      for (int kk = 0; kk < 9 * CI; ++kk)
        for (int f = 0; f < F; ++f) n1[f] += std::fabs((double)K1[(size_t)kk * F + f]);
      double xl = RANGE_LIM;
      auto tighten = [&](double A, double B) {
        const double room = RANGE_LIM - B;
        if (room <= 0.0) xl = 0.0;
        else if (A > 0.0) xl = std::min(xl, room / A);
      };
      for (int f = 0; f < F; ++f) tighten(std::ldexp(n1[f], e1[f]), std::ldexp(std::fabs((double)ep[f]), e1[f]));
      std::vector<double> A2(F, 0.0), B2(F, 0.0);
      for (int fi = 0; fi < F; ++fi) {
        const double ga = std::fabs((double)ep[F + fi]), a = ga * n1[fi], b = ga * std::fabs((double)ep[fi]) + std::fabs((double)ep[2 * F + fi]);
        for (int fo = 0; fo < F; ++fo) {
          const double w = std::fabs((double)K2[(size_t)fi * F + fo]);
          A2[fo] += w * a;
          B2[fo] += w * b;
        }
      }
      for (int fo = 0; fo < F; ++fo) tighten(std::ldexp(A2[fo], e2[fo]), std::ldexp(B2[fo] + std::fabs((double)ep[3 * F + fo]), e2[fo]));
      scales3[6] = (float)(xl * (double)GLOWK_ACT_SCALE * (1.0 - 1e-6));   // the kernel compares its scaled gathers
    }
    const int S1 = pow2_scale(K1f.data(), K1f.size()), S2 = pow2_scale(K2f.data(), K2f.size()), S3 = pow2_scale(K3f.data(), K3f.size());
    if (f16_map) {      // from here on the arrays only steer the image loops: codes instead of values
      for (size_t i = 0; i < K1f.size(); ++i) K1f[i] = (float)(QC.A + i + 1);
      for (size_t i = 0; i < K2f.size(); ++i) K2f[i] = (float)(QC.B + i + 1);
      for (size_t i = 0; i < K3f.size(); ++i) K3f[i] = (float)(QC.C + i + 1);
    }
    const float act = GLOWK_ACT_SCALE;
    scales3[0] = std::ldexp(1.0f, -S1); scales3[1] = std::ldexp(1.0f, -S2); scales3[2] = std::ldexp(1.0f, -S3) / act;
    {
      float* eh = dst + L.epH;          // [conv2 accumulator init (F) | pb (32 * NMT)]
      for (int f = 0; f < F; ++f) eh[f] = (float)std::ldexp((double)act * b2f[f], e2[f] + S2);
      for (int m = 0; m < 32 * NMT; ++m) eh[F + m] = m < 9 * CO ? (float)pbf[m] : 0.0f;
    }
    // image (RingH): conv1 operands of every hidden block [NF][KS][hi|lo][64 lanes] half8 (natural k order, row 9 CI = bias),
    // then per pass: K2 chunks 0..NF-1 and conv3 chunks 0..NMT-1, a chunk = NFH tiles x [2 k-steps][hi|lo][64 lanes] half8
    const int NFH = NF / 2, G0N = NMT < 3 ? NMT : 3, G1N = NMT - G0N > 0 ? NMT - G0N : 1;
    const size_t k1blk = (size_t)KS * 2 * 256, chunkf = (size_t)NFH * 1024;
    float* img = dst + L.RHp;
    if (L.slotH)
    for (int blk = 0; blk < NF; ++blk)
      for (int s2 = 0; s2 < KS; ++s2)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) {
            const int i = l & 31, hh = l >> 5, kk = 16 * s2 + 8 * hh + j;
            const float w = kk <= 9 * CI ? K1f[(size_t)kk * F + blk * 32 + i] : 0.0f;
            float* row_lane = img + (size_t)blk * k1blk + ((size_t)(s2 * 2) * 64 + l) * 4;
            put(row_lane, j, 0, w, S1);
            put(row_lane, j, 1, w, S1);
          }
    if (L.slotH)
    for (int ps = 0; ps < 2; ++ps)
      for (int ch = 0; ch < NF + NMT; ++ch) {
        float* chunk = img + (size_t)NF * k1blk + ((size_t)ps * (NF + NMT) + ch) * chunkf;
        for (int tp = 0; tp < NFH; ++tp)
          for (int s2 = 0; s2 < 2; ++s2)
            for (int l = 0; l < 64; ++l)
              for (int j = 0; j < 8; ++j) {
                const int i = l & 31, hh = l >> 5;
                const int kloc = 16 * s2 + 8 * (j >> 2) + 4 * hh + (j & 3);      // accumulator-derived k order
                float w = 0.0f;
                int S;
                if (ch < NF) { w = K2f[(size_t)(ch * 32 + kloc) * F + (ps * NFH + tp) * 32 + i]; S = S2; }
                else {
                  const int t = (ch - NF) * NFH + tp;                            // RingH::tile_fo / tile_mt
                  const int fo = t < NFH * G0N ? t / G0N : (t - NFH * G0N) / G1N;
                  const int mt = t < NFH * G0N ? t % G0N : G0N + (t - NFH * G0N) % G1N;
                  const int m = mt * 32 + i, f = (ps * NFH + fo) * 32 + kloc;
                  S = S3;
                  if (m < 9 * CO) { const int tap = m / CO, co = m % CO; w = K3f[((size_t)tap * F + f) * CO + co]; }
                }
                float* row_lane = chunk + ((size_t)((tp * 2 + s2) * 2) * 64 + l) * 4;   // hi row of (tile, k-step); 16 B per lane
                put(row_lane, j, 0, w, S);
                put(row_lane, j, 1, w, S);
              }
      }
    // ---- the same network for k_net_h3s (RingS): 16-row A tiles, one k-step of 32 per hidden block; k slot (kq, j) of an
    //      accumulator-derived B fragment is channel 16 (j >> 2) + 4 kq + (j & 3) of the block ----
    if (L.slotS) {
      const int KSS = (9 * CI + 1 + 31) / 32, NMS = (18 * CI + 15) / 16, NRB = 2 * NFH, TPC = 2 * NFH, NT = NFH * NMS;
      const int NCH = (NT + TPC - 1) / TPC;
      const size_t k1blkS = (size_t)KSS * 4 * 256;
      float* imgS = dst + L.RSp;
      for (int blk = 0; blk < NF; ++blk)
        for (int s2 = 0; s2 < KSS; ++s2)
          for (int rb = 0; rb < 2; ++rb)
            for (int l = 0; l < 64; ++l)
              for (int j = 0; j < 8; ++j) {
                const int i = l & 15, kq = l >> 4, kk = 32 * s2 + 8 * kq + j;
                const float w = kk <= 9 * CI ? K1f[(size_t)kk * F + blk * 32 + rb * 16 + i] : 0.0f;
                float* row_lane = imgS + (size_t)blk * k1blkS + ((size_t)((s2 * 2 + rb) * 2) * 64 + l) * 4;
                put(row_lane, j, 0, w, S1);
                put(row_lane, j, 1, w, S1);
              }
      for (int ps = 0; ps < 2; ++ps)
        for (int ch = 0; ch < NF + NCH; ++ch) {
          float* chunk = imgS + (size_t)NF * k1blkS + ((size_t)ps * (NF + NCH) + ch) * chunkf;
          for (int tp = 0; tp < NRB; ++tp)      // NRB row blocks (K2 chunk) or TPC = NRB conv3 tiles
            for (int l = 0; l < 64; ++l)
              for (int j = 0; j < 8; ++j) {
                const int i = l & 15, kq = l >> 4;
                const int kloc = 16 * (j >> 2) + 4 * kq + (j & 3);
                float w = 0.0f;
                int S = S2;
                if (ch < NF) w = K2f[(size_t)(ch * 32 + kloc) * F + ps * NFH * 32 + tp * 16 + i];
                else {
                  const int t = (ch - NF) * TPC + tp;
                  S = S3;
                  if (t < NT) {
                    const int gi = t / (NFH * 6), gn = NMS - 6 * gi < 6 ? NMS - 6 * gi : 6, tl = t - gi * NFH * 6;   // RingS::tile_fo / tile_mt
                    const int fo = tl / gn, mt = 6 * gi + tl % gn;
                    const int m = mt * 16 + i, f = (ps * NFH + fo) * 32 + kloc;
                    if (m < 9 * CO) { const int tap = m / CO, co = m % CO; w = K3f[((size_t)tap * F + f) * CO + co]; }
                  }
                }
                float* row_lane = chunk + ((size_t)(tp * 2) * 64 + l) * 4;
                put(row_lane, j, 0, w, S);
                put(row_lane, j, 1, w, S);
              }
        }
    }
  }

  // ---- f16x3 image of the backward network (k_net_h3, NET_BWD): g_a2 = g2 * mask2 * conv3^T(g_o), g_a1 = g1 * mask1 * K2 g_a2,
  //      per-tap conv1^T.  The BatchNorm factors multiply the producing layer's output rows (a weight change, any sign). ----
  if (L.slotHB || L.slotSB) {
    const int KSB = (9 * c + 15) / 16, NMB = (9 * CI + 31) / 32, NFH = NF / 2;
    const int G0N = NMB < 3 ? NMB : 3, G1N = NMB - G0N > 0 ? NMB - G0N : 1;
    const float* ep = dst + L.ep;       // [b1 | g1 | d1 | b2 | g2 | d2]
    std::vector<float> W3b((size_t)9 * c * F), W2b((size_t)F * F);
    for (int kk = 0; kk < 9 * c; ++kk)
      for (int f = 0; f < F; ++f) { const int tap = kk / c, co = kk % c; W3b[(size_t)kk * F + f] = K3[((size_t)tap * F + f) * c + co] * ep[4 * F + f]; }
    for (int f2 = 0; f2 < F; ++f2)
      for (int f1 = 0; f1 < F; ++f1) W2b[(size_t)f2 * F + f1] = K2[(size_t)f1 * F + f2] * ep[F + f1];      // [k = f2][out = f1]
    {
      // ---- range guard, backward network: with |g_o| <= X,  |g_a2[f]| <= sum_k |W3b[k][f]| X  and
      //      |g_a1[f1]| <= sum_f2 |W2b[f2][f1]| |g_a2[f2]|  (no constants: the backward network is linear) ----
      std::vector<double> a1(F, 0.0), a2(F, 0.0);
      for (int kk = 0; kk < 9 * c; ++kk)
        for (int f = 0; f < F; ++f) a1[f] += std::fabs((double)W3b[(size_t)kk * F + f]);
      for (int f2 = 0; f2 < F; ++f2)
        for (int f1 = 0; f1 < F; ++f1) a2[f1] += std::fabs((double)W2b[(size_t)f2 * F + f1]) * a1[f2];
      double amax = 1.0;
      for (int f = 0; f < F; ++f) amax = std::max(amax, std::max(a1[f], a2[f]));
      scales3[7] = (float)(RANGE_LIM / amax * (double)GLOWK_ACT_SCALE * (1.0 - 1e-6));
    }
    const int S1 = pow2_scale(W3b.data(), W3b.size()), S2 = pow2_scale(W2b.data(), W2b.size()), S3 = pow2_scale(K1, (size_t)9 * CI * F);
    std::vector<float> K1codes;
    const float* K1b = K1;             // conv1^T of the backward network reads the raw kernel
    if (f16_map) {
      for (size_t i = 0; i < W3b.size(); ++i) W3b[i] = (float)(QC.D + i + 1);
      for (size_t i = 0; i < W2b.size(); ++i) W2b[i] = (float)(QC.E + i + 1);
      K1codes.resize((size_t)9 * CI * F);
      for (size_t i = 0; i < K1codes.size(); ++i) K1codes[i] = (float)(QC.G + i + 1);
      K1b = K1codes.data();
    }
    scales3[3] = std::ldexp(1.0f, -S1); scales3[4] = std::ldexp(1.0f, -S2); scales3[5] = std::ldexp(1.0f, -S3) / GLOWK_ACT_SCALE;
    const size_t k1blk = (size_t)KSB * 2 * 256, chunkf = (size_t)NFH * 1024;
    float* img = dst + L.RHBp;
    if (L.slotHB)
    for (int blk = 0; blk < NF; ++blk)
      for (int s2 = 0; s2 < KSB; ++s2)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) {
            const int i = l & 31, hh = l >> 5, kk = 16 * s2 + 8 * hh + j;
            const float w = kk < 9 * c ? W3b[(size_t)kk * F + blk * 32 + i] : 0.0f;
            float* row_lane = img + (size_t)blk * k1blk + ((size_t)(s2 * 2) * 64 + l) * 4;
            put(row_lane, j, 0, w, S1);
            put(row_lane, j, 1, w, S1);
          }
    if (L.slotHB)
    for (int ps = 0; ps < 2; ++ps)
      for (int ch = 0; ch < NF + NMB; ++ch) {
        float* chunk = img + (size_t)NF * k1blk + ((size_t)ps * (NF + NMB) + ch) * chunkf;
        for (int tp = 0; tp < NFH; ++tp)
          for (int s2 = 0; s2 < 2; ++s2)
            for (int l = 0; l < 64; ++l)
              for (int j = 0; j < 8; ++j) {
                const int i = l & 31, hh = l >> 5;
                const int kloc = 16 * s2 + 8 * (j >> 2) + 4 * hh + (j & 3);
                float w = 0.0f;
                int S;
                if (ch < NF) { w = W2b[(size_t)(ch * 32 + kloc) * F + (ps * NFH + tp) * 32 + i]; S = S2; }
                else {
                  const int t = (ch - NF) * NFH + tp;
                  const int fo = t < NFH * G0N ? t / G0N : (t - NFH * G0N) / G1N;
                  const int mt = t < NFH * G0N ? t % G0N : G0N + (t - NFH * G0N) % G1N;
                  const int m = mt * 32 + i, f = (ps * NFH + fo) * 32 + kloc;
                  S = S3;
                  if (m < 9 * CI) w = K1b[(size_t)m * F + f];         // conv1^T: row (tap, cin), contraction over the hidden channel
                }
                float* row_lane = chunk + ((size_t)((tp * 2 + s2) * 2) * 64 + l) * 4;
                put(row_lane, j, 0, w, S);
                put(row_lane, j, 1, w, S);
              }
      }
    // ---- the same backward network for k_net_h3s (RingS<c, 9 CI, NF, bwd>) ----
    if (L.slotSB) {
      const int KSSB = (9 * c + 31) / 32, NMSB = (9 * CI + 15) / 16, NRB = 2 * NFH, TPC = 2 * NFH, NT = NFH * NMSB;
      const int NCHB = (NT + TPC - 1) / TPC, GS0 = NMSB < 6 ? NMSB : 6, GS1 = NMSB - GS0 > 0 ? NMSB - GS0 : 1;
      const size_t k1blkS = (size_t)KSSB * 4 * 256;
      float* imgS = dst + L.RSBp;
      for (int blk = 0; blk < NF; ++blk)
        for (int s2 = 0; s2 < KSSB; ++s2)
          for (int rb = 0; rb < 2; ++rb)
            for (int l = 0; l < 64; ++l)
              for (int j = 0; j < 8; ++j) {
                const int i = l & 15, kq = l >> 4, kk = 32 * s2 + 8 * kq + j;
                const float w = kk < 9 * c ? W3b[(size_t)kk * F + blk * 32 + rb * 16 + i] : 0.0f;
                float* row_lane = imgS + (size_t)blk * k1blkS + ((size_t)((s2 * 2 + rb) * 2) * 64 + l) * 4;
                put(row_lane, j, 0, w, S1);
                put(row_lane, j, 1, w, S1);
              }
      for (int ps = 0; ps < 2; ++ps)
        for (int ch = 0; ch < NF + NCHB; ++ch) {
          float* chunk = imgS + (size_t)NF * k1blkS + ((size_t)ps * (NF + NCHB) + ch) * chunkf;
          for (int tp = 0; tp < NRB; ++tp)
            for (int l = 0; l < 64; ++l)
              for (int j = 0; j < 8; ++j) {
                const int i = l & 15, kq = l >> 4;
                const int kloc = 16 * (j >> 2) + 4 * kq + (j & 3);
                float w = 0.0f;
                int S = S2;
                if (ch < NF) w = W2b[(size_t)(ch * 32 + kloc) * F + ps * NFH * 32 + tp * 16 + i];
                else {
                  const int t = (ch - NF) * TPC + tp;
                  S = S3;
                  if (t < NT) {
                    const int fo = t < NFH * GS0 ? t / GS0 : (t - NFH * GS0) / GS1;
                    const int mt = t < NFH * GS0 ? t % GS0 : GS0 + (t - NFH * GS0) % GS1;
                    const int m = mt * 16 + i, f = (ps * NFH + fo) * 32 + kloc;
                    if (m < 9 * CI) w = K1b[(size_t)m * F + f];
                  }
                }
                float* row_lane = chunk + ((size_t)(tp * 2) * 64 + l) * 4;
                put(row_lane, j, 0, w, S);
                put(row_lane, j, 1, w, S);
              }
        }
    }
  }

  // ---- backward images (input-gradient path): same kernel structure, transposed weights ----
  {
    const int KS3 = (9 * c) / 2, NM1 = (9 * CI + 31) / 32;
    // conv3^T as the small-conv chain: A[i = f local][k = (tap, co)] = K3[tap][f][co]
    for (int fo = 0; fo < NF; ++fo)
      for (int ks = 0; ks < KS3; ++ks)
        for (int l = 0; l < 64; ++l) {
          const int i = l & 31, hh = l >> 5, kk = 2 * ks + hh;
          float v = 0.0f;
          if (kk < 9 * c) { const int tap = kk / c, co = kk % c; v = K3[((size_t)tap * F + fo * 32 + i) * c + co]; }
          dst[L.K3bp + ((size_t)fo * KS3 + ks) * 64 + l] = v;
        }
    const size_t mainf = (size_t)NF * 1024, k3n = (size_t)KS3 * 64;
    for (int c2 = 0; c2 < NF + NM1; ++c2) {
      float* slot = dst + L.RBp + (size_t)c2 * L.slotB;
      if (c2 < NF) {
        // conv2^T chunk of hidden block fo = c2: A[i = f_in local of tile fi][k = f_out = fo*32 + rho(r,hh)] = K2[f_in][f_out]
        const int fo = c2;
        for (int r = 0; r < 16; ++r)
          for (int g = 0; g < NF / 4; ++g)
            for (int l = 0; l < 64; ++l)
              for (int e = 0; e < 4; ++e) {
                const int i = l & 31, hh = l >> 5, fi = 4 * g + e;
                slot[((((size_t)r * (NF / 4)) + g) * 64 + l) * 4 + e] = K2[(size_t)(fi * 32 + i) * F + fo * 32 + rho(r, hh)];
              }
        if (L.k3fB) std::memcpy(slot + mainf, dst + L.K3bp + (size_t)((fo + 1) % NF) * k3n, k3n * 4);
      } else {
        // conv1^T per-tap chunk mt: A[i = m local][k = f = fi*32 + rho(r,hh)] = K1[m][f], m = (tap, cin)
        const int mt = c2 - NF;
        for (int fi = 0; fi < NF; ++fi)
          for (int r4 = 0; r4 < 4; ++r4)
            for (int l = 0; l < 64; ++l)
              for (int e = 0; e < 4; ++e) {
                const int i = l & 31, hh = l >> 5, r = 4 * r4 + e;
                const int m = mt * 32 + i, f = fi * 32 + rho(r, hh);
                slot[((((size_t)fi * 4) + r4) * 64 + l) * 4 + e] = (m < 9 * CI) ? K1[(size_t)m * F + f] : 0.0f;
              }
      }
    }
  }
  return true;
}


// every step packs into its own block of the staging arena: steps are packed by a few host threads
struct PackJob { size_t l; int k; size_t off; double ldc; float sc[8]; std::string err; bool ok; };
inline void pack_all_steps(const glowk_config& cfg, const std::vector<Level>& levels, float* stage, std::vector<PackJob>& jobs, unsigned nthr) {
  std::atomic<size_t> next(0);
  auto work = [&]() {
    for (size_t j; (j = next.fetch_add(1)) < jobs.size();) {
      PackJob& jb = jobs[j];
      jb.ok = pack_step(cfg, levels[jb.l], jb.k, stage + jb.off, &jb.ldc, jb.sc, &jb.err);
    }
  };
  nthr = std::max(1u, std::min<unsigned>(nthr, (unsigned)jobs.size()));
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nthr; ++t) pool.emplace_back(work);
  work();
  for (std::thread& t : pool) t.join();
}

}  // namespace
