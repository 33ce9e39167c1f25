// glowk training: parameter gradients (the weight-gradient batches behind the backward sweep, the ActNorm / 1x1 chain rule), optimizer and
// the device-side refresh of the kernel images -- glowk_param_vector_size / glowk_param_offset / glowk_param_grad / glowk_apply_gradients
// (train_glow.py:29-44, train_utils.py:23-41).  The sweep itself (run_forward / run_backward) lives in glowk.hip.
#include "glowk_engine.h"
#include <chrono>
#include "glowk_train.h"

using namespace glowk_eng;

namespace glowk_eng {
using glowk_detail::num_cus;

// ---- training: layout of the flat parameter / gradient vector, scratch, per-step weight gradients ------------------------
// One block per step (creation order k within a level, levels in order), then the prior:
//   [actnorm log_scale c | actnorm shift c | L c^2 | log_S c | U c^2 | K1 9 ci F | K2 F^2 | K3 9 F c | b1 F | b2 F | b3 c |
//    gamma1 beta1 mean1 var1 gamma2 beta2 mean2 var2 (8 F)]          ... [prior loc E | prior log_scale E]
// (the frozen P, P_inv, sign_S stay on the host; the BatchNorm moving statistics ride along with zero gradient)
struct TrainOff { size_t als, ash, L, logS, U, K1, K2, K3, b1, b2, b3, bn, total; };
TrainOff train_off(int c, int F) {
  TrainOff t;
  size_t o = 0;
  const size_t ci = c / 2;
  t.als = o; o += c; t.ash = o; o += c; t.L = o; o += (size_t)c * c; t.logS = o; o += c; t.U = o; o += (size_t)c * c;
  o = pad4(o);
  t.K1 = o; o += 9 * ci * F; t.K2 = o; o += (size_t)F * F; t.K3 = o; o += (size_t)9 * F * c;
  t.b1 = o; o += F; t.b2 = o; o += F; t.b3 = o; o += pad4(c); t.bn = o; o += (size_t)8 * F;
  t.total = pad4(o);
  return t;
}
int train_tensor_off(const TrainOff& t, int id, size_t* off) {
  switch (id) {
    case GLOWK_ACTNORM_LOG_SCALE: *off = t.als; return 0;
    case GLOWK_ACTNORM_SHIFT: *off = t.ash; return 0;
    case GLOWK_INV1X1_L: *off = t.L; return 0;
    case GLOWK_INV1X1_LOG_S: *off = t.logS; return 0;
    case GLOWK_INV1X1_U: *off = t.U; return 0;
    case GLOWK_CONV1_KERNEL: *off = t.K1; return 0;
    case GLOWK_CONV2_KERNEL: *off = t.K2; return 0;
    case GLOWK_CONV3_KERNEL: *off = t.K3; return 0;
    case GLOWK_CONV1_BIAS: *off = t.b1; return 0;
    case GLOWK_CONV2_BIAS: *off = t.b2; return 0;
    case GLOWK_CONV3_BIAS: *off = t.b3; return 0;
    default: return 1;
  }
}
// (BatchNorm tensors: bn + {0..3} F for layer 1, bn + {4..7} F for layer 2; needs F, so resolved by the caller)
const int TRAIN_IDS[] = {GLOWK_ACTNORM_LOG_SCALE, GLOWK_ACTNORM_SHIFT, GLOWK_INV1X1_L, GLOWK_INV1X1_LOG_S, GLOWK_INV1X1_U, GLOWK_CONV1_KERNEL,
                         GLOWK_CONV2_KERNEL, GLOWK_CONV3_KERNEL, GLOWK_CONV1_BIAS, GLOWK_CONV2_BIAS, GLOWK_CONV3_BIAS, GLOWK_BN1_GAMMA, GLOWK_BN1_BETA,
                         GLOWK_BN1_MEAN, GLOWK_BN1_VAR, GLOWK_BN2_GAMMA, GLOWK_BN2_BETA, GLOWK_BN2_MEAN, GLOWK_BN2_VAR};
bool train_id_off(const TrainOff& t, int F, int id, size_t* off) {
  if (id >= GLOWK_BN1_GAMMA && id <= GLOWK_BN1_VAR) { *off = t.bn + (size_t)(id - GLOWK_BN1_GAMMA) * F; return true; }
  if (id >= GLOWK_BN2_GAMMA && id <= GLOWK_BN2_VAR) { *off = t.bn + (size_t)(4 + id - GLOWK_BN2_GAMMA) * F; return true; }
  return train_tensor_off(t, id, off) == 0;
}

void train_layout(glowk_handle* h) {
  h->tr_level_off.clear();
  size_t o = 0;
  for (const Level& lv : h->levels) {
    h->tr_level_off.push_back(o);
    o += train_off(lv.c, h->cfg.F).total * h->cfg.K;
  }
  h->tr_prior_off = o;
  o += 2 * pad4(h->prior_loc.size());
  h->tr_n = o;
}
size_t train_step_pos(const glowk_handle* h, int lvl, int k) { return h->tr_level_off[lvl] + train_off(h->levels[lvl].c, h->cfg.F).total * (size_t)k; }

constexpr int AFF_BLOCKS = 32;
constexpr size_t AFF_NOUT_MAX = 32 * 32 + 32;
constexpr size_t CPART_FLOATS = (size_t)32 << 20;

// per-step device blocks of a level evenly spaced?  (they are: one image per step inside one allocation per level; checked, not assumed)
bool level_uniform(const Level& lv) {
  const size_t K = lv.dev.size();
  if (K < 2) return true;
  const ptrdiff_t se = lv.dev[1].ep - lv.dev[0].ep, sa = lv.dev[1].Ainv - lv.dev[0].Ainv, sb = lv.dev[1].binv - lv.dev[0].binv;
  for (size_t k = 2; k < K; ++k)
    if (lv.dev[k].ep - lv.dev[0].ep != se * (ptrdiff_t)k || lv.dev[k].Ainv - lv.dev[0].Ainv != sa * (ptrdiff_t)k ||
        lv.dev[k].binv - lv.dev[0].binv != sb * (ptrdiff_t)k)
      return false;
  return true;
}

int ensure_train(glowk_handle* h, int N) {
  if (N <= h->trN) return 0;
  HIPCHK(hipDeviceSynchronize());
  float** bufs[] = {&h->trR1, &h->trR2, &h->trM1, &h->trM2, &h->trXcol, &h->trGcol, &h->trGv, &h->trGo, &h->trC1, &h->trC2, &h->trC3};
  for (float** b : bufs) { if (*b) hipFree(*b); *b = nullptr; }
  if (h->trAffPart) { hipFree(h->trAffPart); h->trAffPart = nullptr; }
  if (h->trKeep) { hipFree(h->trKeep); h->trKeep = nullptr; h->trKeepN = 0; }
  h->trN = 0;
  const size_t F = h->cfg.F, K = h->cfg.K;
  size_t qmax = 0, xcol = 0, gcol = 0, gv = 0;
  for (const Level& lv : h->levels) {
    const size_t Q = (size_t)N * lv.h * lv.w;
    qmax = std::max(qmax, Q);
    xcol = std::max(xcol, (size_t)(9 * (lv.c / 2) + 1) * Q);
    gcol = std::max(gcol, (size_t)9 * lv.c * Q);
    gv = std::max(gv, Q * lv.c);
  }
  // R1 / R2 of all steps at once (4 KB per pixel and step at n_filters 512: 5.6 GB for 32 tiles of 64x64, K = 32, L = 3): kept
  // by the saving forward pass when they fit a quarter of the free memory, otherwise every step re-runs its forward network
  {
    size_t per_tile = 0;
    h->trKeepOff.assign((size_t)h->cfg.L * h->cfg.K, 0);
    for (int lvl = 0; lvl < h->cfg.L; ++lvl)
      for (int j = 0; j < h->cfg.K; ++j) {
        h->trKeepOff[(size_t)lvl * h->cfg.K + j] = per_tile;
        per_tile += 2 * F * (size_t)h->levels[lvl].h * h->levels[lvl].w;
      }
    size_t free_b = 0, tot_b = 0;
    if (hipMemGetInfo(&free_b, &tot_b) == hipSuccess && per_tile * N * 4 <= free_b / 4 && !getenv("GLOWK_TRAIN_RECOMPUTE")) {
      HIPCHK(hipMalloc(&h->trKeep, per_tile * N * 4));
      h->trKeepN = N;
    }
  }
  // a level at a time (trNB = K): M1 / M2, g_o, g_v and the im2col arrays of all K steps of the largest level -- with R1 / R2 kept and
  // a third of what is then free; GLOWK_TRAIN_PERSTEP=1 forces the step-by-step path (both are tested)
  h->trNB = 1;
  if (h->trKeep && K > 1 && !getenv("GLOWK_TRAIN_PERSTEP")) {
    bool uniform = true;
    for (const Level& lv : h->levels) uniform = uniform && level_uniform(lv);
    const size_t extra = (K * (2 * F * qmax + xcol + gcol + 2 * gv)) * 4;
    size_t free_b = 0, tot_b = 0;
    if (uniform && hipMemGetInfo(&free_b, &tot_b) == hipSuccess && extra <= free_b / 3) h->trNB = (int)K;
  }
  const size_t nb = (size_t)h->trNB;
  if (!h->trKeep) HIPCHK(hipMalloc(&h->trR1, 2 * F * qmax * 4));   // R1 | R2 of the step at hand (recomputed)
  HIPCHK(hipMalloc(&h->trM1, nb * F * qmax * 4)); HIPCHK(hipMalloc(&h->trM2, nb * F * qmax * 4));
  HIPCHK(hipMalloc(&h->trXcol, nb * xcol * 4)); HIPCHK(hipMalloc(&h->trGcol, nb * gcol * 4));
  HIPCHK(hipMalloc(&h->trGv, nb * gv * 4)); HIPCHK(hipMalloc(&h->trGo, nb * gv * 4));
  HIPCHK(hipMalloc(&h->trC1, nb * F * (9 * 16 + 1) * 4));
  HIPCHK(hipMalloc(&h->trC2, nb * (F + 1) * F * 4));
  HIPCHK(hipMalloc(&h->trC3, nb * (F + 1) * 9 * 32 * 4));
  HIPCHK(hipMalloc(&h->trAffPart, nb * AFF_BLOCKS * AFF_NOUT_MAX * 8));
  if (!h->trCpart) {
    HIPCHK(hipMalloc(&h->trCpart, CPART_FLOATS * 4));
    HIPCHK(hipMalloc(&h->trAffSum, (size_t)h->cfg.L * h->cfg.K * AFF_NOUT_MAX * 8));
  }
  h->trN = N;
  return 0;
}

// nb GEMMs C_b[M][N] = A_b . B_b^T over K pixels (A_b = A + b bsA, B_b = B + b bsB, C_b = C + b csC): MFMA tiles, split over the pixel
// dimension into S slices where nb * tiles alone would not fill the chip, partials summed in a fixed order (bitwise repeatable, no
// atomics); S = 1 writes C directly.  split: the three-product fp16 form (k_wgrad_h3; sa / sb = the scales the operands are split at),
// else exact fp32 (k_wgrad_nt).
// b_sums: row M of every C_b = the row sums of B_b (the bias / BatchNorm-offset sums a row of ones appended to A would deliver): inside
// the split GEMM, by a k_rowsum pass after the exact one.
int launch_wgrad(glowk_handle* h, bool split, const float* A, ptrdiff_t bsA, int M, const float* B, ptrdiff_t bsB, int N, int K, int nb, float sa, float sb,
                 float* C, size_t csC, bool b_sums, hipStream_t s) {
  const bool big = split ? N >= 256 : (M >= 256 && N >= 256 && glowk_detail::env().wgrad_128);   // (fp32: 128 x 128 tiles measured 5 % slower than 64 x 64)
  const bool big8 = split && big && M % 256 == 0;   // 8 waves, 256 x 128: a quarter less staging per MFMA (254 -> 290 TFLOP/s on the level-0 conv2 batch)
  // 16 waves, 256 x 256 (square shapes: the conv2 gradient): 2/3 of the 8-wave form's staged bytes per MFMA -- staging is what bounds these
  // GEMMs --, 128 registers per wave (one k-step's fragments at a time): 332 -> 373 TFLOP/s on the level-0 batch of a 256-tile step
  const bool big16 = big8 && N % 256 == 0 && (K & 3) == 0 && !glowk_detail::env().wgrad_16_off;
  const int TM = big8 ? 256 : split ? 128 : big ? 128 : 64, TN = big16 ? 256 : split ? (big ? 128 : 64) : TM;
  const int tm = (M + TM - 1) / TM, tn = (N + TN - 1) / TN, tiles = tm * tn;
  static const int wg_env = getenv("GLOWK_WGRAD_WGS") ? atoi(getenv("GLOWK_WGRAD_WGS")) : 0;   // workgroups per CU the split aims at
  const int wg_per_cu = wg_env > 0 ? wg_env : big8 ? 1 : 2;                                       // (what fits a CU: 1 of the 8-wave form)
  int S = std::max(1, std::min((wg_per_cu * num_cus() + tiles * nb - 1) / (tiles * nb), (K + 255) / 256));
  const bool in_gemm = b_sums && split;
  const size_t n = (size_t)(M + (in_gemm ? 1 : 0)) * N;
  S = (int)std::max<size_t>(1, std::min<size_t>((size_t)S, CPART_FLOATS / (n * nb)));   // (the partial buffer bounds the split)
  const int kslice = (((K + S - 1) / S) + 31) / 32 * 32;
  S = (K + kslice - 1) / kslice;
  if (S > 1 && (size_t)S * nb * n > CPART_FLOATS) return fail("wgrad: partial buffer too small");
  float* out = S == 1 ? C : h->trCpart;
  const size_t csz = S == 1 ? csC : n;
  const bool vec = (K & 3) == 0;
  if (split) {
    WgradSplitArgs a;
    a.A = A; a.B = B; a.M = M; a.N = N; a.K = K; a.kslice = kslice; a.S = S; a.tm = tm; a.tn = tn; a.bsA = bsA; a.bsB = bsB; a.sa = sa; a.sb = sb;
    a.Cpart = out; a.csz = csz; a.b_sums = b_sums ? 1 : 0; a.plain = glowk_detail::env().wgrad_plain ? 1 : 0;
    const dim3 grid((unsigned)(tiles * S * nb));
    if (big16) hipLaunchKernelGGL((k_wgrad_h3<2, 2, 4, 4, true>), grid, dim3(1024), 0, s, a);
    else if (big8 && vec) hipLaunchKernelGGL((k_wgrad_h3<2, 2, 4, 2, true>), grid, dim3(512), 0, s, a);
    else if (big8) hipLaunchKernelGGL((k_wgrad_h3<2, 2, 4, 2, false>), grid, dim3(512), 0, s, a);
    else if (big && vec) hipLaunchKernelGGL((k_wgrad_h3<2, 2, 2, 2, true>), grid, dim3(256), 0, s, a);
    else if (big) hipLaunchKernelGGL((k_wgrad_h3<2, 2, 2, 2, false>), grid, dim3(256), 0, s, a);
    else if (vec) hipLaunchKernelGGL((k_wgrad_h3<1, 2, 4, 1, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_wgrad_h3<1, 2, 4, 1, false>), grid, dim3(256), 0, s, a);
    LAUNCHCHK("k_wgrad_h3");
  } else {
    WgradArgs a;
    a.A = A; a.B = B; a.M = M; a.N = N; a.a_ones = 0; a.K = K; a.kslice = kslice; a.Cpart = out; a.S = S; a.tm = tm; a.tn = tn; a.bsA = bsA; a.bsB = bsB;
    a.csz = csz;
    const dim3 grid((unsigned)(tiles * S * nb));
    if (big && vec) hipLaunchKernelGGL((k_wgrad_nt<2, true>), grid, dim3(256), 0, s, a);
    else if (big) hipLaunchKernelGGL((k_wgrad_nt<2, false>), grid, dim3(256), 0, s, a);
    else if (vec) hipLaunchKernelGGL((k_wgrad_nt<1, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_wgrad_nt<1, false>), grid, dim3(256), 0, s, a);
    LAUNCHCHK("k_wgrad_nt");
  }
  if (S > 1) {
    hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)((n + 255) / 256), nb), dim3(256), 0, s, (const float*)h->trCpart, S, n, C, csC);
    LAUNCHCHK("k_sum_parts");
  }
  if (b_sums && !in_gemm) {
    hipLaunchKernelGGL(k_rowsum, dim3(N, nb), dim3(256), 0, s, B, K, C + (size_t)M * N, bsB, csC);
    LAUNCHCHK("k_rowsum");
  }
  return 0;
}


// Weight gradients of the coupling networks of steps k0 .. k0 + nb - 1 of a level (k = backward order), as ONE batch of launches.
// Entry b reads the saved coupling input v + b v_bs [Q][c], the gradient wrt the network output g_o + b go_bs [Q][c], R1 + b r_bs
// (R2 follows at + F Q) and M1 / M2 + b m_bs, all planar [F][Q].
int train_network_grads(glowk_handle* h, TrainCtx* tc, int lvl, int k0, int nb, const float* v, ptrdiff_t v_bs, const float* g_o, ptrdiff_t go_bs,
                        const float* R1, ptrdiff_t r_bs, const float* M1, const float* M2, ptrdiff_t m_bs, int N, hipStream_t s, float bfac) {
  const Level& lv = h->levels[lvl];
  const int F = h->cfg.F, c = lv.c, ci = c / 2, Q = N * lv.h * lv.w;
  const int N1 = 9 * ci + 1, N3 = 9 * c;
  const float* R2 = R1 + (size_t)F * Q;
  const size_t xs = (size_t)N1 * Q, gs = (size_t)N3 * Q, c1s = (size_t)F * N1, c2s = (size_t)(F + 1) * F, c3s = (size_t)(F + 1) * N3;
  // (1) planar im2col operands
  hipLaunchKernelGGL(k_im2col_planar, dim3((Q + 255) / 256, nb), dim3(256), 0, s, v, c, ci, ci, Q, lv.h, lv.w, 1, 1, h->trXcol, v_bs, (ptrdiff_t)xs);
  hipLaunchKernelGGL(k_im2col_planar, dim3((Q + 255) / 256, nb), dim3(256), 0, s, g_o, c, 0, c, Q, lv.h, lv.w, -1, 0, h->trGcol, go_bs, (ptrdiff_t)gs);
  LAUNCHCHK("k_im2col_planar");
  // (2) the three GEMMs over the pixel dimension.  A split sweep left R / M in the units its kernels split them in; the im2col
  //     arrays take the scale of those kernels' own gathers.  Row F of C3 = the row sums of Gcol, row F of C2 = sum_q M2.
  static const bool gemm_f32 = getenv("GLOWK_WGRAD_F32") != nullptr;   // (A/B: the exact GEMMs under a split sweep)
  const bool sg = tc->split && !gemm_f32;
  const float act = sg ? GLOWK_ACT_SCALE : 1.0f;
  if (int rc = launch_wgrad(h, sg, R2, r_bs, F, h->trGcol, (ptrdiff_t)gs, N3, Q, nb, 1.0f, act, h->trC3, c3s, true, s)) return rc;
  if (int rc = launch_wgrad(h, sg, R1, r_bs, F, M2, m_bs, F, Q, nb, 1.0f, 1.0f, h->trC2, c2s, true, s)) return rc;
  if (int rc = launch_wgrad(h, sg, M1, m_bs, F, h->trXcol, (ptrdiff_t)xs, N1, Q, nb, 1.0f, act, h->trC1, c1s, false, s)) return rc;
  // (3) assemble into the flat gradient vector
  const TrainOff t = train_off(c, F);
  const float* p = h->tr_params + train_step_pos(h, lvl, k0);
  float* g = tc->grad + train_step_pos(h, lvl, k0);
  StepGradArgs a;
  a.F = F; a.c = c; a.K2 = p + t.K2; a.K3 = p + t.K3; a.bn = p + t.bn; a.ep = lv.dev[k0].ep; a.eps = h->cfg.bn_eps; a.scaled = tc->split ? 1 : 0;
  a.C1 = h->trC1; a.C2 = h->trC2; a.C3 = h->trC3; a.scale = tc->scale / bfac;   // (g_o, M1, M2 and their sums all carry bfac: a power of two, exact)
  a.dK1 = g + t.K1; a.dK2 = g + t.K2; a.dK3 = g + t.K3; a.db1 = g + t.b1; a.db2 = g + t.b2; a.db3 = g + t.b3;
  a.dgamma1 = g + t.bn; a.dbeta1 = g + t.bn + F; a.dgamma2 = g + t.bn + 4 * (size_t)F; a.dbeta2 = g + t.bn + 5 * (size_t)F;
  a.ps = t.total; a.es = nb > 1 ? (size_t)(lv.dev[k0 + 1].ep - lv.dev[k0].ep) : 0; a.c1s = c1s; a.c2s = c2s; a.c3s = c3s;
  const size_t work = std::max({(size_t)F * F, (size_t)9 * F * c, (size_t)9 * ci * F});
  hipLaunchKernelGGL(k_assemble_step_grads, dim3((unsigned)((work + 255) / 256), nb), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_assemble_channel_grads, dim3(F, nb), dim3(256), 0, s, a);
  LAUNCHCHK("k_assemble_step_grads");
  return 0;
}

// sums for the ActNorm / 1x1 gradients of steps k0 .. k0 + nb - 1: dA = sum_q u^T g_v, db = sum_q g_v  ->  trAffSum[(lvl K + k)]
int train_affine_sums(glowk_handle* h, int lvl, int k0, int nb, const float* v, ptrdiff_t v_bs, const float* gv, ptrdiff_t gv_bs, int N, hipStream_t s) {
  const Level& lv = h->levels[lvl];
  const StepDev& sd = lv.dev[k0];
  const int Q = N * lv.h * lv.w, c = lv.c;
  const size_t nout = (size_t)c * c + c;
  const ptrdiff_t a_bs = nb > 1 ? lv.dev[k0 + 1].Ainv - sd.Ainv : 0, b_bs = nb > 1 ? lv.dev[k0 + 1].binv - sd.binv : 0;
  CDISPATCH(c, hipLaunchKernelGGL((k_affine_wgrad<CC>), dim3(AFF_BLOCKS, nb), dim3(256), 0, s, v, gv, Q, sd.Ainv, sd.binv, h->trAffPart, v_bs, gv_bs, a_bs, b_bs));
  LAUNCHCHK("k_affine_wgrad");
  hipLaunchKernelGGL(k_sum_parts_f64, dim3((unsigned)((nout + 255) / 256), nb), dim3(256), 0, s, (const double*)h->trAffPart, AFF_BLOCKS, nout,
                     h->trAffSum + ((size_t)lvl * h->cfg.K + k0) * AFF_NOUT_MAX, AFF_NOUT_MAX);
  LAUNCHCHK("k_sum_parts_f64");
  return 0;
}


// ---- training: host side ------------------------------------------------------------------------------------------------
// host tensors -> flat vector (staging); the inverse is sync_host
void params_to_flat(const glowk_handle* h, std::vector<float>& flat) {
  flat.assign(h->tr_n, 0.0f);
  const int F = h->cfg.F;
  for (size_t l = 0; l < h->levels.size(); ++l) {
    const Level& lv = h->levels[l];
    const TrainOff t = train_off(lv.c, F);
    for (int k = 0; k < h->cfg.K; ++k) {
      float* dst = flat.data() + train_step_pos(h, (int)l, k);
      for (int id : TRAIN_IDS) {
        size_t off;
        train_id_off(t, F, id, &off);
        const std::vector<float>& v = lv.host[id][k];
        std::memcpy(dst + off, v.data(), v.size() * 4);
      }
    }
  }
  const size_t E = h->prior_loc.size();
  std::memcpy(flat.data() + h->tr_prior_off, h->prior_loc.data(), E * 4);
  std::memcpy(flat.data() + h->tr_prior_off + pad4(E), h->prior_log_scale.data(), E * 4);
}

int sync_host(glowk_handle* h) {
  if (!h->host_stale) return 0;
  std::vector<float> flat(h->tr_n);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(flat.data(), h->tr_params, h->tr_n * 4, hipMemcpyDeviceToHost));
  const int F = h->cfg.F;
  for (size_t l = 0; l < h->levels.size(); ++l) {
    Level& lv = h->levels[l];
    const TrainOff t = train_off(lv.c, F);
    for (int k = 0; k < h->cfg.K; ++k) {
      const float* src = flat.data() + train_step_pos(h, (int)l, k);
      for (int id : TRAIN_IDS) {
        size_t off;
        train_id_off(t, F, id, &off);
        std::vector<float>& v = lv.host[id][k];
        std::memcpy(v.data(), src + off, v.size() * 4);
      }
    }
  }
  const size_t E = h->prior_loc.size();
  std::memcpy(h->prior_loc.data(), flat.data() + h->tr_prior_off, E * 4);
  std::memcpy(h->prior_log_scale.data(), flat.data() + h->tr_prior_off + pad4(E), E * 4);
  h->host_stale = false;
  return 0;
}

// packed-image position -> position in the step's parameter block, for the exact-fp32 images of one level: the host packer
// itself, run on index-coded kernels (one tensor at a time: codes 1 .. n are exact in fp32)
int build_repack_map(glowk_handle* h, int lvl, std::vector<int>& map, size_t* region_off) {
  const glowk_config& cfg = h->cfg;
  const Level& src = h->levels[lvl];
  const int c = src.c, F = cfg.F;
  const StepLayout SL = step_layout(c, F);
  const TrainOff t = train_off(c, F);
  const size_t lo = SL.K1p, hi = SL.RHp;      // [K1p | ep | R0p | K3bp | RBp) -- the images k_net_f32 reads
  map.assign(hi - lo, -1);
  for (size_t i = SL.ep; i < SL.R0p; ++i) map[i - lo] = -2;
  Level tmp;
  tmp.h = src.h; tmp.w = src.w; tmp.c = c; tmp.z_off = 0; tmp.z_width = 0; tmp.Cz = 0;
  for (int id = 0; id < GLOWK_NUM_STEP_TENSORS; ++id) tmp.host[id].assign(1, std::vector<float>(step_tensor_size(cfg, src, id), 0.0f));
  for (int i = 0; i < c; ++i) { tmp.host[GLOWK_INV1X1_P][0][(size_t)i * c + i] = 1.0f; tmp.host[GLOWK_INV1X1_SIGN_S][0][i] = 1.0f; }
  std::fill(tmp.host[GLOWK_BN1_GAMMA][0].begin(), tmp.host[GLOWK_BN1_GAMMA][0].end(), 1.0f);
  std::fill(tmp.host[GLOWK_BN2_GAMMA][0].begin(), tmp.host[GLOWK_BN2_GAMMA][0].end(), 1.0f);
  std::fill(tmp.host[GLOWK_BN1_VAR][0].begin(), tmp.host[GLOWK_BN1_VAR][0].end(), 1.0f);
  std::fill(tmp.host[GLOWK_BN2_VAR][0].begin(), tmp.host[GLOWK_BN2_VAR][0].end(), 1.0f);
  std::vector<float> stage(SL.total);
  const int ids[3] = {GLOWK_CONV1_KERNEL, GLOWK_CONV2_KERNEL, GLOWK_CONV3_KERNEL};
  const size_t offs[3] = {t.K1, t.K2, t.K3};
  for (int w = 0; w < 3; ++w) {
    std::vector<float>& ten = tmp.host[ids[w]][0];
    if (ten.size() >= ((size_t)1 << 24)) return fail("repack map: tensor too large for exact index codes");
    for (size_t i = 0; i < ten.size(); ++i) ten[i] = (float)(i + 1);
    std::fill(stage.begin(), stage.end(), 0.0f);
    double ldc; float sc[8]; std::string err;
    if (!pack_step(cfg, tmp, 0, stage.data(), &ldc, sc, &err)) return fail("repack map: " + err);
    for (size_t i = lo; i < hi; ++i) {
      if (i >= SL.ep && i < SL.R0p) continue;
      const float v = stage[i];
      if (v != 0.0f) {
        if (map[i - lo] != -1) return fail("repack map: a packed position is written by two tensors");
        map[i - lo] = (int)(offs[w] + (size_t)v - 1);
      }
    }
    std::fill(ten.begin(), ten.end(), 0.0f);
  }
  *region_off = lo;
  return 0;
}

int train_begin(glowk_handle* h) {
  if (h->tr_active) return 0;
  if (!h->finalized) return fail("glowk_finalize_weights has not been called");
  if (h->tr_n == 0) train_layout(h);
  if (!h->tr_params) {
    HIPCHK(hipMalloc(&h->tr_params, h->tr_n * 4));
    HIPCHK(hipMalloc(&h->tr_m, h->tr_n * 4));
    HIPCHK(hipMalloc(&h->tr_v, h->tr_n * 4));
    HIPCHK(hipMemset(h->tr_m, 0, h->tr_n * 4));
    HIPCHK(hipMemset(h->tr_v, 0, h->tr_n * 4));
    h->tr_map.assign(h->levels.size(), nullptr);
    h->tr_map_n.assign(h->levels.size(), 0);
    for (size_t l = 0; l < h->levels.size(); ++l) {
      std::vector<int> map;
      size_t lo;
      if (int rc = build_repack_map(h, (int)l, map, &lo)) return rc;
      HIPCHK(hipMalloc(&h->tr_map[l], map.size() * 4));
      HIPCHK(hipMemcpy(h->tr_map[l], map.data(), map.size() * 4, hipMemcpyHostToDevice));
      h->tr_map_n[l] = map.size();
    }
    // the same for the fp16-split images: the packer in map mode
    h->tr_map16.assign(h->levels.size(), nullptr);
    h->tr_map16_n.assign(h->levels.size(), 0);
    size_t src_max = 0;
    for (size_t l = 0; l < h->levels.size(); ++l) {
      const Level& lv = h->levels[l];
      const StepLayout SL = step_layout(lv.c, h->cfg.F);
      if (!(SL.slotH || SL.slotS || SL.slotHB || SL.slotSB)) continue;
      Level tmp;
      tmp.h = lv.h; tmp.w = lv.w; tmp.c = lv.c; tmp.z_off = 0; tmp.z_width = 0; tmp.Cz = 0;
      for (int id = 0; id < GLOWK_NUM_STEP_TENSORS; ++id) tmp.host[id].assign(1, lv.host[id][0]);   // any valid step: only the layout matters
      std::vector<float> stage(SL.total, 0.0f);
      std::vector<int> map16(SL.total * 2, -1);
      double ldc; float sc[8]; std::string err;
      if (!pack_step(h->cfg, tmp, 0, stage.data(), &ldc, sc, &err, map16.data())) return fail("f16 repack map: " + err);
      const size_t lo = SL.RHp * 2, n = (SL.Afwd - SL.RHp) * 2;
      HIPCHK(hipMalloc(&h->tr_map16[l], n * 4));
      HIPCHK(hipMemcpy(h->tr_map16[l], map16.data() + lo, n * 4, hipMemcpyHostToDevice));
      h->tr_map16_n[l] = n;
      src_max = std::max(src_max, f16_code_bases(lv.c, h->cfg.F).total);
    }
    if (src_max) {
      HIPCHK(hipMalloc(&h->tr16_src, (size_t)h->cfg.L * h->cfg.K * src_max * 4));
      HIPCHK(hipMalloc(&h->tr16_S, (size_t)h->cfg.L * h->cfg.K * 6 * 4));
      HIPCHK(hipMalloc(&h->tr16_scales, (size_t)h->cfg.L * h->cfg.K * 8 * 4));
      h->tr16_src_max = src_max;
    }
  }
  std::vector<float> flat;
  params_to_flat(h, flat);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(h->tr_params, flat.data(), h->tr_n * 4, hipMemcpyHostToDevice));
  h->tr_active = true;
  h->host_stale = false;
  return 0;
}

// chain rule from the fused per-pixel affine v = u A + b (A = diag(e^ls) W, b = sh W, W = P Lm Um) to the reference's variables
// (flow_tfp_bijectors.py:236-239, 289-303), fp64.  sums = [dA (c x c) | db (c)] = sum_q u^T g_v, sum_q g_v; the log-det terms
// h w (sum ls + sum log_S) per sample add N h w to d/d ls_i and d/d log_S_i.  out: the step's [als | ash | L | logS | U] gradients.
void affine_chain_rule(const Level& lv, int k, const double* sums, int N, double scale, float* out, const TrainOff& t) {
  const int c = lv.c;
  auto T = [&](int id) -> const float* { return lv.host[id][k].data(); };
  const float *ls = T(GLOWK_ACTNORM_LOG_SCALE), *sh = T(GLOWK_ACTNORM_SHIFT), *P = T(GLOWK_INV1X1_P), *Lh = T(GLOWK_INV1X1_L), *Uh = T(GLOWK_INV1X1_U),
              *sg = T(GLOWK_INV1X1_SIGN_S), *lS = T(GLOWK_INV1X1_LOG_S);
  Mat Pm(c * c), Lm(c * c), Um(c * c);
  for (int i = 0; i < c; ++i)
    for (int j = 0; j < c; ++j) {
      Pm[i * c + j] = P[i * c + j];
      Lm[i * c + j] = (i > j) ? Lh[i * c + j] : (i == j ? 1.0 : 0.0);
      Um[i * c + j] = (i < j) ? Uh[i * c + j] : (i == j ? (double)sg[i] * std::exp((double)lS[i]) : 0.0);
    }
  const Mat PL = matmul(Pm, Lm, c), W = matmul(PL, Um, c);
  const double* dA = sums;
  const double* db = sums + (size_t)c * c;
  const double ld = (double)N * lv.h * lv.w;
  Mat dW(c * c);
  for (int i = 0; i < c; ++i) {
    double dls = ld, dsh = 0.0;
    const double e = std::exp((double)ls[i]);
    for (int j = 0; j < c; ++j) {
      dls += dA[i * c + j] * e * W[i * c + j];
      dsh += db[j] * W[i * c + j];
      dW[i * c + j] = e * dA[i * c + j] + (double)sh[i] * db[j];
    }
    out[t.als + i] = (float)(scale * dls);
    out[t.ash + i] = (float)(scale * dsh);
  }
  // dLm = P^T dW Um^T ; dUm = (P Lm)^T dW
  Mat Pt(c * c), Ut(c * c), PLt(c * c);
  for (int i = 0; i < c; ++i)
    for (int j = 0; j < c; ++j) { Pt[i * c + j] = Pm[j * c + i]; Ut[i * c + j] = Um[j * c + i]; PLt[i * c + j] = PL[j * c + i]; }
  const Mat dLm = matmul(matmul(Pt, dW, c), Ut, c), dUm = matmul(PLt, dW, c);
  for (int i = 0; i < c; ++i) {
    for (int j = 0; j < c; ++j) {
      out[t.L + i * c + j] = (i > j) ? (float)(scale * dLm[i * c + j]) : 0.0f;
      out[t.U + i * c + j] = (i < j) ? (float)(scale * dUm[i * c + j]) : 0.0f;
    }
    out[t.logS + i] = (float)(scale * (dUm[i * c + i] * (double)sg[i] * std::exp((double)lS[i]) + ld));
  }
}


}  // namespace glowk_eng

// =================================================================================================
extern "C" {

int glowk_finalize_weights(glowk_handle* h);

size_t glowk_param_vector_size(glowk_handle* h) {
  if (!h) return 0;
  if (h->tr_n == 0) train_layout(h);
  return h->tr_n;
}

int glowk_param_offset(glowk_handle* h, int level, int step, int tensor_id, size_t* offset, size_t* count) {
  if (!h || !offset || !count) return fail("null argument");
  if (h->tr_n == 0) train_layout(h);
  if (tensor_id == GLOWK_PRIOR_LOC || tensor_id == GLOWK_PRIOR_LOG_SCALE) {
    *offset = h->tr_prior_off + (tensor_id == GLOWK_PRIOR_LOG_SCALE ? pad4(h->prior_loc.size()) : 0);
    *count = h->prior_loc.size();
    return 0;
  }
  if (level < 0 || level >= h->cfg.L || step < 0 || step >= h->cfg.K) return fail("no such step");
  size_t off;
  if (!train_id_off(train_off(h->levels[level].c, h->cfg.F), h->cfg.F, tensor_id, &off)) return fail("tensor is not part of the parameter vector (frozen P, P_inv, sign_S)");
  *offset = train_step_pos(h, level, step) + off;
  *count = step_tensor_size(h->cfg, h->levels[level], tensor_id);
  return 0;
}

static int param_grad_impl(glowk_handle* h, const float* x_dev, int N, float scale, float* logp_dev, float* grad_dev, void* stream, bool allow_split,
                           bool* tripped) {
  const int prec = h->precision;
  struct Restore { glowk_handle* h; int p; ~Restore() { h->precision = p; } } restore{h, prec};
  if (int rc = check_ready(h, N)) return rc;       // (in a split precision this re-packs stale f16 images first)
  if (int rc = train_begin(h)) return rc;
  if (int rc = ensure_train(h, N)) return rc;
  // The sweep runs in the handle's arithmetic where the split kernels have training instances for every level (32x32x16 family,
  // forward and backward images) and the hiddens are kept by the forward pass; otherwise on the exact fp32 kernels.
  bool split = allow_split && prec != GLOWK_PREC_F32 && h->trKeep && N <= h->trKeepN && !getenv("GLOWK_TRAIN_F32");
  for (const Level& lv : h->levels) {
    if (!split) break;
    const StepDev& sd = lv.dev[0];
    split = (sd.RHp && sd.RHBp) || (sd.RSp && sd.RSBp);
    if (split) {      // ... and the launch policy has an instance of both storing kernels for this level at this batch size
      NetArgs pf = net_args(h, lv, sd, nullptr, lv.c, lv.c / 2, N), pb = net_args(h, lv, sd, nullptr, lv.c, 0, N);
      pb.RHp = sd.RHBp; pb.RSp = sd.RSBp;
      split = launch_net_raw(lv.c, h->cfg.F, pf, 10, nullptr, true) > 0 && launch_net_raw(lv.c, h->cfg.F, pb, 11, nullptr, true) > 0;
    }
  }
  h->precision = split ? GLOWK_PREC_F16X3 : GLOWK_PREC_F32;
  if (int rc = ensure_save(h, N)) return rc;
  hipStream_t s = (hipStream_t)stream;
  const glowk_config& cfg = h->cfg;
  float* z = h->bufGz;
  HIPCHK(hipMemsetAsync(grad_dev, 0, h->tr_n * 4, s));
  if (int rc = run_forward(h, x_dev, N, z, s, true, h->trKeep && N <= h->trKeepN)) return rc;
  const int E = h->Hl * h->Wl * h->Cl;
  if (logp_dev) {
    if (int rc = launch_prior(h, z, N, logp_dev, s)) return rc;
  }
  if (cfg.learntop) {
    hipLaunchKernelGGL(k_prior_wgrad, dim3((E + 255) / 256), dim3(256), 0, s, (const float*)z, N, E, h->d_loc, h->d_log_scale, scale,
                       grad_dev + h->tr_prior_off, grad_dev + h->tr_prior_off + pad4((size_t)E));
    LAUNCHCHK("k_prior_wgrad");
  }
  if (!h->tr_gmax) {
    HIPCHK(hipMalloc(&h->tr_gmax, sizeof(unsigned) * 64));
    HIPCHK(hipHostMalloc(&h->h_gmax, sizeof(float) * 64));
    h->tr_bfac.assign(4, 1.0f);
  }
  HIPCHK(hipMemsetAsync(h->tr_gmax, 0, sizeof(unsigned) * 64, s));
  const size_t steps = (size_t)cfg.L * cfg.K;
  if (!h->tr_side) {
    HIPCHK(hipStreamCreateWithFlags(&h->tr_side, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->tr_ev_sums, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->tr_ev_up, hipEventDisableTiming));
    HIPCHK(hipHostMalloc((void**)&h->h_sums, steps * AFF_NOUT_MAX * 8, hipHostMallocDefault));
    size_t up = 0;
    for (int lvl = 0; lvl < cfg.L; ++lvl) up += (size_t)cfg.K * train_off(h->levels[lvl].c, cfg.F).K1;
    HIPCHK(hipHostMalloc((void**)&h->h_up, up * 4, hipHostMallocDefault));
  }
  TrainCtx tc{grad_dev, scale, split, h->tr_ev_sums};
  // (the input gradient falls out of the sweep as well; the trainer has no use for it: it lands in the block-level scratch,
  //  which is free again by the time the last kernel of the sweep writes it)
  if (int rc = run_backward(h, x_dev, z, N, h->bufZ, s, &tc)) return rc;
  // ActNorm / 1x1: the per-step sums come down once -- on a side stream, as soon as the sweep has produced the last of them, while the
  // main stream still runs the last level's weight-gradient GEMMs --, the c x c chain rule runs on the host in fp64, the results go up
  // on the side stream (they land in the heads of the step blocks, which no kernel of the sweep writes after the initial memset), and
  // the caller's stream waits for that upload: no host join of the caller's stream at all
  hipStream_t side = h->tr_side;
  const bool host_times = getenv("GLOWK_HOST_TIMES") != nullptr;      // (diagnostic: what does the host's share of a training step take?)
  if (getenv("GLOWK_PG_JOIN")) HIPCHK(hipStreamSynchronize(s));       // (A/B timing: the host joins the caller's stream first, as it did before the side stream)
  HIPCHK(hipStreamWaitEvent(side, h->tr_ev_sums, 0));
  HIPCHK(hipMemcpyAsync(h->h_sums, h->trAffSum, steps * AFF_NOUT_MAX * 8, hipMemcpyDeviceToHost, side));
  if (split) HIPCHK(hipMemcpyAsync(h->h_flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, side));
  HIPCHK(hipMemcpyAsync(h->h_gmax, h->tr_gmax, sizeof(float) * 64, hipMemcpyDeviceToHost, side));
  HIPCHK(hipStreamSynchronize(side));
  const auto ht0 = std::chrono::steady_clock::now();
  if (split && h->h_flag[0]) {       // the range guard of the split arithmetic fired somewhere in the sweep: its gradients are not usable
    h->h_flag[0] = 0;
    HIPCHK(hipMemsetAsync(h->d_flag, 0, sizeof(int), side));
    HIPCHK(hipEventRecord(h->tr_ev_up, side));
    HIPCHK(hipStreamWaitEvent(s, h->tr_ev_up, 0));      // (the exact sweep that follows on the caller's stream starts after the reset)
    *tripped = true;
    return 0;
  }
  {
    // dynamic gradient scaling of the NEXT split sweep: per level, the power of two that puts this sweep's largest |g_o| a factor
    // 256 below what the static bound of the level's backward networks admits (xlim_b, in units of GLOWK_ACT_SCALE * g_o) -- room
    // for the largest gradient entry (a heavy-tailed quantity: it moved 44x between two early Adamax steps of the benchmark model)
    // to grow 256x from one step to the next before a sweep has to be repeated on the exact kernels
    float gm[4] = {0.f, 0.f, 0.f, 0.f};
    for (int lvl = 0; lvl < cfg.L; ++lvl)
      for (int i = 0; i < 16; ++i) gm[lvl] = h->h_gmax[16 * lvl + i] > gm[lvl] || !(h->h_gmax[16 * lvl + i] == h->h_gmax[16 * lvl + i]) ? h->h_gmax[16 * lvl + i] : gm[lvl];
    for (int lvl = 0; lvl < cfg.L; ++lvl) {
      float xl = 0.f;
      for (const StepDev& sd : h->levels[lvl].dev) xl = xl == 0.f ? sd.xlim_b : std::min(xl, sd.xlim_b);
      if (!(gm[lvl] > 0.f) || !(gm[lvl] < 3.0e38f) || !(xl > 0.f)) continue;
      int e = 0;
      std::frexp(xl / 256.0f / ((float)GLOWK_ACT_SCALE * gm[lvl]), &e);
      h->tr_bfac[lvl] = std::ldexp(1.0f, std::max(-100, std::min(100, e - 1)));
    }
  }
  float* out = h->h_up;      // (pinned; rewritten by the next call only after it has waited for the side stream, which these uploads are on)
  for (int lvl = 0; lvl < cfg.L; ++lvl) {
    const Level& lv = h->levels[lvl];
    const TrainOff t = train_off(lv.c, cfg.F);
    const size_t small = t.K1;            // [als | ash | L | logS | U] (padded) sit at the head of the step block
    std::memset(out, 0, (size_t)cfg.K * small * 4);
    for (int k = 0; k < cfg.K; ++k)
      affine_chain_rule(lv, k, h->h_sums + ((size_t)lvl * cfg.K + k) * AFF_NOUT_MAX, N, (double)scale, out + (size_t)k * small, t);
    HIPCHK(hipMemcpy2DAsync(grad_dev + h->tr_level_off[lvl], t.total * 4, out, small * 4, small * 4, cfg.K, hipMemcpyHostToDevice, side));
    out += (size_t)cfg.K * small;
  }
  HIPCHK(hipEventRecord(h->tr_ev_up, side));
  HIPCHK(hipStreamWaitEvent(s, h->tr_ev_up, 0));
  if (host_times) fprintf(stderr, "glowk_param_grad: host share (chain rule of %zu steps + uploads) %.0f us\n", steps,
                          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ht0).count());
  return 0;
}

int glowk_param_grad(glowk_handle* h, const float* x_dev, int N, float scale, float* logp_dev, float* grad_dev, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (!x_dev || !grad_dev) return fail("null tensor");
  bool tripped = false;
  if (int rc = param_grad_impl(h, x_dev, N, scale, logp_dev, grad_dev, stream, true, &tripped)) return rc;
  if (!tripped) return 0;
  if (h->range_policy == GLOWK_RANGE_ERROR) {
    g_err = "training sweep: a hidden activation or gradient left the fp16 range of the split arithmetic -- use GLOWK_PREC_F32 or GLOWK_RANGE_FALLBACK";
    return GLOWK_ERR_RANGE;
  }
  ++h->range_fallbacks;      // FALLBACK (and IGNORE: a gradient vector of NaNs would poison the parameters): the exact kernels
  return param_grad_impl(h, x_dev, N, scale, logp_dev, grad_dev, stream, false, &tripped);
}

int glowk_apply_gradients(glowk_handle* h, const float* grad_dev, int optimizer, float lr, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (!grad_dev) return fail("null tensor");
  if (optimizer != 0 && optimizer != 1) return fail("optimizer argument should be adam (0) or adamax (1)");   // train_utils.py:40
  if (!h->tr_active) return fail("glowk_apply_gradients: no gradient has been computed for the current parameters (call glowk_param_grad first)");
  hipStream_t s = (hipStream_t)stream;
  const glowk_config& cfg = h->cfg;
  const int F = cfg.F;
  const double b1 = 0.9, b2 = 0.999;
  bool stale16 = false;      // set when a level's fp16-split images are NOT refreshed below (handle in exact fp32)
  h->tr_t += 1;
  const double t = (double)h->tr_t;
  const float lr_t = optimizer == 1 ? (float)(lr / (1.0 - std::pow(b1, t))) : (float)(lr * std::sqrt(1.0 - std::pow(b2, t)) / (1.0 - std::pow(b1, t)));
  hipLaunchKernelGGL(k_optimizer, dim3((unsigned)((h->tr_n + 255) / 256)), dim3(256), 0, s, h->tr_params, grad_dev, h->tr_m, h->tr_v, h->tr_n, optimizer,
                     lr_t, (float)b1, (float)b2, 1e-7f);
  LAUNCHCHK("k_optimizer");
  // ---- refresh what the kernels read: conv images (device permutation), BatchNorm/bias block, fused affines, prior.  The levels are
  //      independent and their kernels small and latency-bound (serial fp64 sums in the host packer's order, so that the images stay
  //      bit for bit the host-packed ones): each level runs on a stream of its own (the exact-fp32 image permutation, which nothing in
  //      the chain of the split images needs, on a second one).  The host's share -- the c x c fp64 fold of ActNorm + 1x1 -- needs only
  //      the updated small tensors: they come down FIRST, so the fold runs while the device refreshes the images; the split kernels'
  //      scale arguments (host-side step descriptors) come down before the last, longest kernel of the chain ----
  const int L = cfg.L;
  if (h->tr_streams.empty()) {
    h->tr_streams.resize(2 * L); h->tr_events.resize(1 + 4 * L);     // streams: [lvl] chain, [L + lvl] side; events: [0] fork, [1 + lvl] chain done,
    for (hipStream_t& t : h->tr_streams) HIPCHK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking));      // [1 + L + lvl] small tensors down,
    for (hipEvent_t& e : h->tr_events) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));        // [1 + 2L + lvl] scales down, [1 + 3L + lvl] side done
    size_t tot = 0;
    h->tr_pin_off.assign(L + 1, 0);
    for (int lvl = 0; lvl < L; ++lvl) {
      const StepLayout SL = step_layout(h->levels[lvl].c, F);
      const TrainOff t = train_off(h->levels[lvl].c, F);
      h->tr_pin_off[lvl] = tot;
      tot += (size_t)cfg.K * (8 + t.K1 + h->levels[lvl].c + (SL.total - SL.Afwd));
    }
    h->tr_pin_off[L] = tot;
    HIPCHK(hipHostMalloc((void**)&h->tr_pinned, tot * 4, hipHostMallocDefault));
  }
  HIPCHK(hipEventRecord(h->tr_events[0], s));
  // pinned staging of level lvl: scales | small tensors | conv3 biases | folded affine blocks
  auto pin_sc = [&](int lvl) { return h->tr_pinned + h->tr_pin_off[lvl]; };
  auto pin_sm = [&](int lvl) { return pin_sc(lvl) + (size_t)cfg.K * 8; };
  auto pin_b3 = [&](int lvl) { return pin_sm(lvl) + (size_t)cfg.K * train_off(h->levels[lvl].c, F).K1; };
  auto pin_blk = [&](int lvl) { return pin_b3(lvl) + (size_t)cfg.K * h->levels[lvl].c; };
  std::vector<char> refresh16(L, 0);
  for (int lvl = 0; lvl < L; ++lvl) {
    Level& lv = h->levels[lvl];
    hipStream_t ls = h->tr_streams[lvl], side = h->tr_streams[L + lvl];
    HIPCHK(hipStreamWaitEvent(ls, h->tr_events[0], 0));
    HIPCHK(hipStreamWaitEvent(side, h->tr_events[0], 0));
    const StepLayout SL = step_layout(lv.c, F);
    const TrainOff t = train_off(lv.c, F);
    const float* p0 = h->tr_params + h->tr_level_off[lvl];
    float* img0 = h->arena + lv.dev[0].arena_off;
    // small tensors: down to the host (they parameterise the fp64 fold of ActNorm + 1x1), folded there, back up
    // (pinned buffers: rewritten by the next call only after it has waited for this level's stream, which the uploads below are on)
    const size_t small = t.K1;
    HIPCHK(hipMemcpy2DAsync(pin_sm(lvl), small * 4, p0, t.total * 4, small * 4, cfg.K, hipMemcpyDeviceToHost, ls));
    HIPCHK(hipMemcpy2DAsync(pin_b3(lvl), (size_t)lv.c * 4, p0 + t.b3, t.total * 4, (size_t)lv.c * 4, cfg.K, hipMemcpyDeviceToHost, ls));
    HIPCHK(hipEventRecord(h->tr_events[1 + L + lvl], ls));
    hipLaunchKernelGGL(k_repack_f32, dim3((unsigned)((h->tr_map_n[lvl] + 255) / 256), cfg.K), dim3(256), 0, side, (const int*)h->tr_map[lvl], h->tr_map_n[lvl],
                       p0, t.total, img0 + SL.K1p, SL.total);
    HIPCHK(hipEventRecord(h->tr_events[1 + 3 * L + lvl], side));
    hipLaunchKernelGGL(k_fold_bn, dim3((F + 255) / 256, cfg.K), dim3(256), 0, ls, p0 + t.b1, p0 + t.b2, p0 + t.bn, t.total, F, cfg.bn_eps, img0 + SL.ep, SL.total);
    LAUNCHCHK("k_repack_f32");
    // the fp16-split images, when the handle is in a split arithmetic (otherwise they are re-packed lazily by the host)
    refresh16[lvl] = h->precision != GLOWK_PREC_F32 && h->tr_map16[lvl];
    if (refresh16[lvl]) {
      const F16Codes q = f16_code_bases(lv.c, F);
      F16Prep fp;
      fp.params = p0; fp.param_stride = t.total; fp.ep = img0 + SL.ep; fp.img_stride = SL.total;
      fp.oK1 = t.K1; fp.oK2 = t.K2; fp.oK3 = t.K3; fp.ob1 = t.b1; fp.ob2 = t.b2; fp.c = lv.c; fp.F = F;
      fp.cA = q.A; fp.cB = q.B; fp.cC = q.C; fp.cD = q.D; fp.cE = q.E; fp.cG = q.G; fp.cTot = q.total;
      fp.src = h->tr16_src + (size_t)lvl * cfg.K * h->tr16_src_max; fp.S = h->tr16_S + (size_t)lvl * cfg.K * 6; fp.scales = h->tr16_scales + (size_t)lvl * cfg.K * 8;
      const int NMT = (9 * lv.c + 31) / 32;
      hipLaunchKernelGGL(k_f16_sources, dim3((unsigned)((q.total + 255) / 256), cfg.K), dim3(256), 0, ls, fp);
      hipLaunchKernelGGL(k_f16_absmax, dim3(6, cfg.K), dim3(1024), 0, ls, fp);
      hipLaunchKernelGGL(k_f16_consts, dim3((F + 32 * NMT + 255) / 256, cfg.K), dim3(256), 0, ls, fp, img0 + SL.epH, NMT);
      hipLaunchKernelGGL(k_f16_limits, dim3(2, cfg.K), dim3(512), 0, ls, fp);
      HIPCHK(hipMemcpyAsync(pin_sc(lvl), fp.scales, (size_t)cfg.K * 8 * 4, hipMemcpyDeviceToHost, ls));
      HIPCHK(hipEventRecord(h->tr_events[1 + 2 * L + lvl], ls));
      hipLaunchKernelGGL(k_repack_f16, dim3((unsigned)((h->tr_map16_n[lvl] + 1023) / 1024), cfg.K), dim3(256), 0, ls, (const int*)h->tr_map16[lvl],
                         h->tr_map16_n[lvl], fp, reinterpret_cast<unsigned short*>(img0 + SL.RHp), SL.total * 2);
      LAUNCHCHK("k_repack_f16");
    } else if (h->tr_map16[lvl]) {
      stale16 = true;
    }
  }
  const bool host_times = getenv("GLOWK_HOST_TIMES") != nullptr;
  double fold_us = 0.0;
  for (int lvl = 0; lvl < L; ++lvl) {
    Level& lv = h->levels[lvl];
    hipStream_t ls = h->tr_streams[lvl];
    HIPCHK(hipEventSynchronize(h->tr_events[1 + L + lvl]));      // the small tensors are down
    const auto ht0 = std::chrono::steady_clock::now();
    const StepLayout SL = step_layout(lv.c, F);
    const TrainOff t = train_off(lv.c, F);
    float* img0 = h->arena + lv.dev[0].arena_off;
    const size_t small = t.K1, tail = SL.total - SL.Afwd;
    std::vector<float> tmp(SL.total);
    float* blocks = pin_blk(lvl);
    for (int k = 0; k < cfg.K; ++k) {
      const float* src = pin_sm(lvl) + (size_t)k * small;
      const int c = lv.c;
      std::memcpy(lv.host[GLOWK_ACTNORM_LOG_SCALE][k].data(), src + t.als, c * 4);
      std::memcpy(lv.host[GLOWK_ACTNORM_SHIFT][k].data(), src + t.ash, c * 4);
      std::memcpy(lv.host[GLOWK_INV1X1_L][k].data(), src + t.L, (size_t)c * c * 4);
      std::memcpy(lv.host[GLOWK_INV1X1_LOG_S][k].data(), src + t.logS, c * 4);
      std::memcpy(lv.host[GLOWK_INV1X1_U][k].data(), src + t.U, (size_t)c * c * 4);
      std::memcpy(lv.host[GLOWK_CONV3_BIAS][k].data(), pin_b3(lvl) + (size_t)k * c, c * 4);
      std::string err;
      double ldc = 0;
      if (!pack_affine(cfg, lv, k, tmp.data(), &ldc, &err)) return fail("level " + std::to_string(lvl) + " step " + std::to_string(k) + ": " + err);
      h->ld_step[(size_t)lvl * cfg.K + k] = ldc;
      std::memcpy(blocks + (size_t)k * tail, tmp.data() + SL.Afwd, tail * 4);
    }
    fold_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ht0).count();
    HIPCHK(hipMemcpy2DAsync(img0 + SL.Afwd, SL.total * 4, blocks, tail * 4, tail * 4, cfg.K, hipMemcpyHostToDevice, ls));
    HIPCHK(hipEventRecord(h->tr_events[1 + lvl], ls));
    HIPCHK(hipStreamWaitEvent(s, h->tr_events[1 + lvl], 0));     // whatever the caller's stream runs next sees the refreshed images
    HIPCHK(hipStreamWaitEvent(s, h->tr_events[1 + 3 * L + lvl], 0));
  }
  for (int lvl = 0; lvl < L; ++lvl) {
    if (!refresh16[lvl]) continue;      // the kernels' scale arguments and range-guard limits live in the host-side step descriptors
    Level& lv = h->levels[lvl];
    const StepLayout SL = step_layout(lv.c, F);
    HIPCHK(hipEventSynchronize(h->tr_events[1 + 2 * L + lvl]));
    for (int k = 0; k < cfg.K; ++k) {
      StepDev& d = lv.dev[k];
      const float* q8 = pin_sc(lvl) + (size_t)k * 8;
      if (SL.slotH || SL.slotS) { d.sc1 = q8[0]; d.sc2 = q8[1]; d.sc3 = q8[2]; d.xlim_f = q8[6]; }
      if (SL.slotHB || SL.slotSB) { d.scb1 = q8[3]; d.scb2 = q8[4]; d.scb3 = q8[5]; d.xlim_b = q8[7]; }
    }
  }
  if (host_times) fprintf(stderr, "glowk_apply_gradients: host fold of ActNorm + 1x1 (all levels) %.0f us\n", fold_us);
  h->ld_const = 0.0;
  for (double v : h->ld_step) h->ld_const += v;
  if (cfg.learntop) {
    const size_t E = h->prior_loc.size();
    HIPCHK(hipMemcpyAsync(const_cast<float*>(h->d_loc), h->tr_params + h->tr_prior_off, E * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(const_cast<float*>(h->d_log_scale), h->tr_params + h->tr_prior_off + pad4(E), E * 4, hipMemcpyDeviceToDevice, s));
  }
  h->host_stale = true;
  h->split_stale = stale16;
  return 0;
}


}  // extern "C"
