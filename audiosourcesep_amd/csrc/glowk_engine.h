// glowk engine, shared between its translation units (round 4: glowk.hip was one 2 200-line file):
//   glowk.hip            weight store, launch sequencing of the forward / inverse / gradient passes, range guard, the inference ABI
//   glowk_training.hip   parameter gradients (weight-gradient batches, chain rule), optimizer, device-side image refresh: the training ABI
//   glowk_aux.hip        handle-free entry points: BASIS update kernel, device RNG, CRC-32C
// Everything here lives in namespace glowk_eng (external linkage: the functions declared at the bottom cross the units).
#pragma once
#include "../../include/glowk.h"
#include "glowk_kernels.h"
#include "glowk_pack.h"
#include "glowk_launch.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#include <string>
#include <thread>
#include <vector>

namespace glowk_eng {

extern thread_local std::string g_err;      // glowk_last_error(): glowk.hip

inline int fail(const std::string& m) {
  g_err = m;
  return GLOWK_ERR;
}

// every entry point that touches the GPU runs on the handle's device and hands the caller's current device back
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define HIPCHK(expr)                                                                           \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return glowk_eng::fail(std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

#define LAUNCHCHK(name)                                                                        \
  do {                                                                                         \
    hipError_t e_ = hipGetLastError();                                                         \
    if (e_ != hipSuccess) return glowk_eng::fail(std::string("launch ") + name + ": " + hipGetErrorString(e_)); \
  } while (0)

}  // namespace glowk_eng

struct glowk_handle {
  glowk_config cfg;
  int device = 0;
  int precision = GLOWK_PREC_F32;
  int range_policy = GLOWK_RANGE_ERROR;
  int64_t range_fallbacks = 0;
  int64_t family_launches[7] = {0, 0, 0, 0, 0, 0, 0};   // coupling-network launches by kernel family (glowk_launch.h: note_family)
  int64_t fused_steps = 0;      // flow steps that ran as ONE network + coupling kernel (net_and_couple)
  int* d_flag = nullptr;        // sticky range flag (device), written by k_couple / k_bwd_light
  int* h_flag = nullptr;        // pinned host word it is read back into
  unsigned* d_probe = nullptr;  // glowk_range_probe_begin .. end: [2][L][K] largest gathered network input (float bits) of the forward / backward launches
  std::vector<Level> levels;
  int Hl = 0, Wl = 0, Cl = 0;
  std::vector<float> prior_loc, prior_log_scale;
  bool finalized = false;
  double ld_const = 0.0;        // sum over steps of h*w*(sum log_scale + sum log_S)
  double ld_pre_const = 0.0;    // data-independent part of the preprocessing log-det
  std::vector<double> ld_step;  // per (level*K + step) constant
  float* arena = nullptr;       // packed weights
  const float* d_loc = nullptr;
  const float* d_log_scale = nullptr;
  // workspace
  int wsN = 0;
  float *bufA = nullptr, *bufB = nullptr, *bufP = nullptr, *bufZ = nullptr, *bufC = nullptr;   // bufP: 4 partials, pstride apart
  size_t pstride = 0;
  double* bufLd = nullptr;
  double* bufLdSlot = nullptr;  // [min(wsN, 2 CUs)][ld_slots_per_sample]
  double* bufStat = nullptr;    // [STAT_BLOCKS][32] partial sums + [32] means
  // input-gradient path: per-step saves of the forward pass (v, P, ReLU masks) and gradient scratch
  int saveN = 0;
  size_t cN = 0;                // tiles bufC holds
  float *saveV = nullptr, *saveP = nullptr, *bufGz = nullptr;   // saveP: every step's pre-tanh log_s inputs [Q][c/2] (CoupleArgs::o_save); the per-tap
                                                                // conv3 outputs P themselves are scratch (bufP) in the saving pass too
  unsigned short* saveM = nullptr;
  std::vector<size_t> offV, offP, offM;   // per forward-order step
  // training (glowk_param_grad / glowk_apply_gradients): device master copy of every parameter, optimizer state, scratch
  bool tr_active = false;        // tr_params holds the current parameters
  bool host_stale = false;       // ... and the host tensors of the conv / BatchNorm / prior parameters are behind it
  bool split_stale = false;      // ... and so are the f16 images (rebuilt by the host packer at the next split-precision call)
  float *tr_params = nullptr, *tr_m = nullptr, *tr_v = nullptr;
  size_t tr_n = 0;
  long tr_t = 0;                 // optimizer steps taken
  std::vector<size_t> tr_level_off;
  size_t tr_prior_off = 0;
  std::vector<int*> tr_map;      // per level: packed-image position -> position in the step's parameter block (k_repack_f32)
  std::vector<size_t> tr_map_n;
  std::vector<int*> tr_map16;    // per level: half position in [RHp, Afwd) -> source code | lo bit (k_repack_f16), or null
  std::vector<size_t> tr_map16_n;
  float* tr16_src = nullptr;     // [L][K][max cTot] scaled sources of a level's steps (a block per level: the levels refresh concurrently)
  int* tr16_S = nullptr;         // [L][K][6]
  float* tr16_scales = nullptr;  // [L][K][8]
  size_t tr16_src_max = 0;
  float* tr_pinned = nullptr;            // pinned host staging of glowk_apply_gradients (a pageable target would make every copy block the host)
  std::vector<size_t> tr_pin_off;        // per level: [scales K*8 | small K*K1off | b3 K*c | affine blocks K*tail]
  std::vector<hipStream_t> tr_streams;   // one per level: glowk_apply_gradients refreshes the levels' images side by side
  std::vector<hipEvent_t> tr_events;     // [0] fork, [1 + lvl] join
  std::vector<float> tr_bfac;    // per level: power of two the split training sweep scales g_o by (BwdArgs::go_scale), adapted after every
                                 // sweep from the largest |g_o| it saw (dynamic gradient scaling); 1 until the first sweep has run
  unsigned* tr_gmax = nullptr;   // [L][16] device: that maximum (float bits), per level, spread over 16 words
  float* h_gmax = nullptr;       // pinned host copy (read back with the sweep's one synchronisation)
  // glowk_param_grad: the host's ActNorm / 1x1 chain rule runs BESIDE the last level's weight-gradient GEMMs: a side stream brings the
  // per-step sums down as soon as the sweep has produced them and takes the results back up
  hipStream_t tr_side = nullptr;
  hipEvent_t tr_ev_sums = nullptr, tr_ev_up = nullptr;    // sums complete (main stream) / results uploaded (side stream)
  double* h_sums = nullptr;      // pinned [L * K][AFF_NOUT_MAX]
  float* h_up = nullptr;         // pinned: per level K x (the head of a step block: als | ash | L | logS | U)
  int trN = 0;
  float *trR1 = nullptr, *trR2 = nullptr, *trM1 = nullptr, *trM2 = nullptr, *trXcol = nullptr, *trGcol = nullptr, *trCpart = nullptr;
  float *trC1 = nullptr, *trC2 = nullptr, *trC3 = nullptr, *trGv = nullptr, *trGo = nullptr;
  int trNB = 1;                  // steps whose weight-gradient work runs as one batch: K (a level at a time: the planar arrays of all its
                                 // steps are kept until its sweep is over) when the memory is there and the per-step device blocks are
                                 // evenly spaced, else 1 (step by step)
  double *trAffPart = nullptr, *trAffSum = nullptr;
  float* trSmall = nullptr;      // staging of the small (ActNorm / 1x1 / conv3-bias) parameters, device side
  float* trKeep = nullptr;       // R1 | R2 of EVERY step, written by the saving forward pass itself when the memory is there (else recomputed per step)
  std::vector<size_t> trKeepOff; // per forward-order step: offset of its R1 (R2 follows at + F Q), in floats per tile
  int trKeepN = 0;
  // HIP-event profiler of k_net
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<int> ev_level;   // level of each (start, stop) pair
};

#define CDISPATCH(c, CALL)                                                                     \
  switch (c) {                                                                                 \
    case 4: { constexpr int CC = 4; CALL; } break;                                             \
    case 8: { constexpr int CC = 8; CALL; } break;                                             \
    case 16: { constexpr int CC = 16; CALL; } break;                                           \
    case 32: { constexpr int CC = 32; CALL; } break;                                           \
    default: return glowk_eng::fail("unsupported channel count " + std::to_string(c));         \
  }

namespace glowk_eng {

struct TrainCtx {
  float* grad;     // [tr_n] flat gradient vector (device, caller owned)
  float scale;     // every gradient is scale * d sum_n log_prob / d theta
  bool split;      // the sweep runs the fp16-split kernels (k_net_h3, MODE | 8): planar arrays in scaled units (StepGradArgs::scaled)
  hipEvent_t sums_ready = nullptr;   // recorded on the sweep's stream once every ActNorm / 1x1 sum (and the range flag, the gradient maxima) is final
};

// ---- defined in glowk.hip
int check_ready(glowk_handle* h, int N);
int ensure_save(glowk_handle* h, int N);
NetArgs net_args(glowk_handle* h, const Level& lv, const StepDev& sd, const float* vin, int in_stride, int in_off, int N);
int launch_net_raw(int c, int F, const NetArgs& a, int mode, hipStream_t s, bool dry = false);
int run_forward(glowk_handle* h, const float* x, int N, float* z_dst, hipStream_t s, bool save = false, bool keep_hidden = false);
int run_backward(glowk_handle* h, const float* x, const float* z, int N, float* dx, hipStream_t s, TrainCtx* tc = nullptr);
int launch_prior(glowk_handle* h, const float* z, int N, float* logp_dev, hipStream_t s);     // k_prior: log N(z) + the accumulated log-det
int ptr_device(const void* p);

// ---- defined in glowk_training.hip
int sync_host(glowk_handle* h);
int train_network_grads(glowk_handle* h, TrainCtx* tc, int lvl, int k0, int nb, const float* v, ptrdiff_t v_bs, const float* g_o, ptrdiff_t go_bs,
                        const float* R1, ptrdiff_t r_bs, const float* M1, const float* M2, ptrdiff_t m_bs, int N, hipStream_t s, float bfac = 1.0f);
int train_affine_sums(glowk_handle* h, int lvl, int k0, int nb, const float* v, ptrdiff_t v_bs, const float* gv, ptrdiff_t gv_bs, int N, hipStream_t s);

}  // namespace glowk_eng
