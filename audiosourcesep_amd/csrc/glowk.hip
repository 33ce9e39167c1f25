// glowk engine: weight store, host-side packing, launch sequencing and the C ABI of include/glowk.h.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC glowk.hip -o libglowk.so   (see __graft_entry__.build)
#include "../../include/glowk.h"
#include "glowk_kernels.h"
#include "glowk_light.h"
#include "glowk_basis.h"
#include "glowk_train.h"
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

namespace {

thread_local std::string g_err;

int fail(const std::string& m) {
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
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

#define LAUNCHCHK(name)                                                                        \
  do {                                                                                         \
    hipError_t e_ = hipGetLastError();                                                         \
    if (e_ != hipSuccess) return fail(std::string("launch ") + name + ": " + hipGetErrorString(e_)); \
  } while (0)

}  // namespace

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

namespace {

// ---- launch helpers (policy templates: glowk_launch.h) --------------------------------------------
}  // namespace

namespace glowk_detail {
int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
bool h3_shape16() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("GLOWK_H3_SHAPE"); v = (e && atoi(e) == 32) ? 0 : 1; }
  return v == 1;
}
void launch_fail(const std::string& m) { fail(m); }
unsigned long long* g_dbg_stamps = nullptr;    // glowk_debug_stamps
static EnvSwitches read_env() {
  auto on = [](const char* n) { return getenv(n) != nullptr; };
  return EnvSwitches{on("GLOWK_HALF_OFF"), on("GLOWK_HALF_FORCE"), on("GLOWK_FAM16_SMALL"), on("GLOWK_BWD_LIGHT_4"), on("GLOWK_COUPLE_PER_SAMPLE"),
                     on("GLOWK_COUPLE_4"), on("GLOWK_NO_FUSE"), on("GLOWK_WGRAD_PLAIN"), on("GLOWK_WGRAD_128"), on("GLOWK_CO_OFF"), on("GLOWK_Q_OFF"), on("GLOWK_CO_SPLIT_OFF")};
}
static EnvSwitches g_env = read_env();
const EnvSwitches& env() { return g_env; }
void reload_env() { g_env = read_env(); }
thread_local int g_family = 0;
thread_local bool g_co = false, g_q = false;
void note_family(int family) { g_family = family; g_co = false; g_q = false; }
void note_co() { g_co = true; }
void note_q() { g_q = true; }
// instantiated in glowk_net_inst.hip, one translation unit per (CI, NF)
#define GLOWK_EXTERN_NET(CI_, NF_) extern template int launch_net_t<CI_, NF_>(const NetArgs&, int, hipStream_t, bool);
GLOWK_EXTERN_NET(2, 16) GLOWK_EXTERN_NET(4, 16) GLOWK_EXTERN_NET(8, 16) GLOWK_EXTERN_NET(16, 16)
GLOWK_EXTERN_NET(2, 12) GLOWK_EXTERN_NET(4, 12) GLOWK_EXTERN_NET(8, 12) GLOWK_EXTERN_NET(16, 12)
GLOWK_EXTERN_NET(2, 8) GLOWK_EXTERN_NET(4, 8) GLOWK_EXTERN_NET(8, 8) GLOWK_EXTERN_NET(16, 8)
GLOWK_EXTERN_NET(2, 4) GLOWK_EXTERN_NET(4, 4) GLOWK_EXTERN_NET(8, 4) GLOWK_EXTERN_NET(16, 4)
#undef GLOWK_EXTERN_NET
}  // namespace glowk_detail

namespace {
using glowk_detail::num_cus;
using glowk_detail::launch_net_t;

// launch_net mode of the plain forward network for the handle's precision (glowk_launch.h: 0 exact fp32, 3 three-term split,
// 6 two-term split)
int fwd_mode(const glowk_handle* h) { return h->precision == GLOWK_PREC_F16X3 ? 3 : h->precision == GLOWK_PREC_F16X2 ? 6 : NET_FWD; }

// the light kernels raise the sticky range flag only for calls in a split arithmetic (in exact fp32 a non-finite value is the
// reference's own result, and the fp32 re-run of the FALLBACK policy must not re-arm it)
int* flagp(const glowk_handle* h) { return h->precision == GLOWK_PREC_F32 ? nullptr : h->d_flag; }

// normalisation target of the split backward kernels for a step whose static bound allows inputs up to xlim_b (scaled units):
// the largest power of two T with 2 T <= xlim_b, at most 2^10; 0 if even 2^-4 does not fit
float bwd_norm_target(float xlim_b) {
  if (!(xlim_b >= 0.125f)) return 0.0f;
  int e = 0;
  std::frexp(xlim_b * 0.5f, &e);                 // xlim_b / 2 = f 2^e, f in [0.5, 1): largest power of two <= it is 2^(e-1)
  return std::ldexp(1.0f, std::min(e - 1, 10));
}

int launch_net_raw(int c, int F, const NetArgs& a, int mode, hipStream_t s, bool dry = false) {
  glowk_detail::g_family = 0;     // (the fp32 kernels do not announce themselves)
  glowk_detail::g_co = false;
  glowk_detail::g_q = false;
#define NETCASE(CI_, NF_) if (c == 2 * CI_ && F == 32 * NF_) return launch_net_t<CI_, NF_>(a, mode, s, dry);
  NETCASE(2, 16) NETCASE(4, 16) NETCASE(8, 16) NETCASE(16, 16)
  NETCASE(2, 12) NETCASE(4, 12) NETCASE(8, 12) NETCASE(16, 12)
  NETCASE(2, 8) NETCASE(4, 8) NETCASE(8, 8) NETCASE(16, 8)
  NETCASE(2, 4) NETCASE(4, 4) NETCASE(8, 4) NETCASE(16, 4)
#undef NETCASE
  fail("unsupported (channels, n_filters) combination: c=" + std::to_string(c) + " F=" + std::to_string(F));
  return -1;
}

// np_out: number of partial P buffers the launch wrote (P + p * pstride), for the consumer
int launch_net(glowk_handle* h, int level, int c, int F, const NetArgs& a, hipStream_t s, int mode = NET_FWD, int* np_out = nullptr) {
  if (!h->profiling) {
    const int np = launch_net_raw(c, F, a, mode, s);
    if (np >= 0) { ++h->family_launches[glowk_detail::g_family]; if (glowk_detail::g_co) ++h->family_launches[5]; if (glowk_detail::g_q) ++h->family_launches[6]; }
    if (np_out) *np_out = np;
    return np < 0 ? 1 : 0;
  }
  while (h->ev_pool.size() < h->ev_used + 2) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    h->ev_pool.push_back(e);
  }
  hipEvent_t e0 = h->ev_pool[h->ev_used], e1 = h->ev_pool[h->ev_used + 1];
  HIPCHK(hipEventRecord(e0, s));
  const int np = launch_net_raw(c, F, a, mode, s);
  if (np >= 0) { ++h->family_launches[glowk_detail::g_family]; if (glowk_detail::g_co) ++h->family_launches[5]; if (glowk_detail::g_q) ++h->family_launches[6]; }
  HIPCHK(hipEventRecord(e1, s));
  h->ev_used += 2;
  h->ev_level.push_back(level);
  if (np_out) *np_out = np;
  return np < 0 ? 1 : 0;
}

#define CDISPATCH(c, CALL)                                                                     \
  switch (c) {                                                                                 \
    case 4: { constexpr int CC = 4; CALL; } break;                                             \
    case 8: { constexpr int CC = 8; CALL; } break;                                             \
    case 16: { constexpr int CC = 16; CALL; } break;                                           \
    case 32: { constexpr int CC = 32; CALL; } break;                                           \
    default: return fail("unsupported channel count " + std::to_string(c));                    \
  }

// light backward kernel of one step: 16 lanes per pixel where the batch is small and the level deep (few pixels, many channels and
// partial buffers: the gathers' dependent loads are what the launch waits for), one where the grid is large, else 4
int launch_bwd_light(int c, const BwdArgs& a, int N, hipStream_t s) {
  const bool wide = c >= 8 && (a.Q + 15) / 16 <= 8 * num_cus() && !glowk_detail::env().bwd_light_4;
  // large grids: one lane per pixel (the planar Pg gathers and the 16-byte rows of the [Q][C] arrays are then fully coalesced)
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool one = (a.Q + 255) / 256 >= 2 * num_cus() && al16(a.ghalf_in) && al16(a.v) && al16(a.osave) && al16(a.g_o) && al16(a.ghalf_out) &&
                   al16(a.gu_out) && al16(a.gv_out) && !glowk_detail::env().bwd_light_4;
  if (one) { CDISPATCH(c, hipLaunchKernelGGL((k_bwd_light<CC, 1>), dim3((a.Q + 255) / 256), dim3(256), 0, s, a)); }
  else if (wide) { CDISPATCH(c, hipLaunchKernelGGL((k_bwd_light<CC, 16>), dim3((a.Q + 15) / 16), dim3(256), 0, s, a)); }
  else { CDISPATCH(c, hipLaunchKernelGGL((k_bwd_light<CC, 4>), dim3((a.Q + 63) / 64), dim3(256), 0, s, a)); }
  LAUNCHCHK("k_bwd_light");
  return 0;
}

// log-det slots of the flat-grid coupling kernel (k_couple_flat): slot[n * stride + base + workgroup within the sample]
struct FlatLd { double* slot; int stride, base; };

int launch_couple(int c, const CoupleArgs& a, int N, hipStream_t s, const FlatLd* fl = nullptr, bool* flat_used = nullptr) {
  const int hw = a.h * a.w;
  if (N < 2 * num_cus() && hw % 64 == 0 && (!a.logdet || (fl && fl->slot)) && !glowk_detail::env().couple_per_sample) {
    // few samples: a flat grid over the pixels instead of one workgroup per sample (30 tiles: 30 workgroups on 256 CUs); sixteen
    // lanes per pixel where the level is deep (c >= 8) and small
    CoupleArgs b = a;
    b.logdet = nullptr;
    double* slots = a.logdet ? fl->slot : (double*)nullptr;
    const bool wide = c >= 8 && (a.Q + 15) / 16 <= 8 * num_cus() && !glowk_detail::env().couple_4;
    if (wide) { CDISPATCH(c, hipLaunchKernelGGL((k_couple_flat<CC, 16>), dim3((a.Q + 15) / 16), dim3(256), 0, s, b, slots, fl ? fl->stride : 0, fl ? fl->base : 0)); }
    else { CDISPATCH(c, hipLaunchKernelGGL((k_couple_flat<CC, 4>), dim3((a.Q + 63) / 64), dim3(256), 0, s, b, slots, fl ? fl->stride : 0, fl ? fl->base : 0)); }
    LAUNCHCHK("k_couple_flat");
    if (flat_used && a.logdet) *flat_used = true;
    return 0;
  }
  if (N >= 2 * num_cus()) {   // enough per-sample workgroups to fill the chip: one lane per pixel
    CDISPATCH(c, hipLaunchKernelGGL((k_couple<CC, false>), dim3(N), dim3(256), 0, s, a));
  } else {                    // four lanes per pixel, up to 256 pixels in flight per sample
    const int threads = hw >= 256 ? 1024 : hw > 64 ? 512 : 256;
    CDISPATCH(c, hipLaunchKernelGGL((k_couple<CC, true>), dim3(N), dim3(threads), 0, s, a));
  }
  LAUNCHCHK("k_couple");
  return 0;
}

// One flow step's coupling network + coupling.  Plain forward direction of the split arithmetics at the 4-channel level with a grid
// that fills the chip: ONE kernel (k_net_h3s<..., MODE | 16>: the per-tap conv3 outputs never leave the workgroup) plus
// k_couple_edge for the pixel rows whose 3 x 3 neighbourhood straddles two workgroups and the per-sample log-det; otherwise k_net
// (P to HBM) + k_couple.  na: the network launch (na.P / pstride as for launch_net); ca: the coupling as k_couple takes it.
bool fuse_geometry_ok(int h, int w, int pxw = 256) {     // pxw: pixels per workgroup of the fused kernel
  const int hw = h * w;
  if (w < 4 || w > FUSE_EW || (w & (w - 1)) || pxw % w) return false;
  if (hw % pxw == 0) return true;
  return hw >= 32 && hw < pxw && (hw & (hw - 1)) == 0;
}

int net_and_couple(glowk_handle* h, int lvl, int c, int F, NetArgs na, CoupleArgs ca, int N, hipStream_t s, int mode, const FlatLd* fl = nullptr,
                   bool* flat_used = nullptr) {
  const bool no_fuse = glowk_detail::env().no_fuse;            // (A/B timing and the fused-vs-unfused parity test, which calls glowk_reload_env)
  const int hw = ca.h * ca.w;
  if (!no_fuse && c == 4 && (mode == 3 || mode == 6 || mode == 4) && na.RSp && ca.vin && !ca.log_s_out && ca.out && na.in_stride == 4 &&
      ca.out_stride % 4 == 0 && ca.out_off % 4 == 0 && fuse_geometry_ok(ca.h, ca.w) && na.P == h->bufP) {
    // (scratch sized for the co-resident form's 128-pixel workgroups when it may be taken: twice the edge slots)
    const size_t pxw = (na.co && fuse_geometry_ok(ca.h, ca.w, CO_PX)) ? CO_PX : 256;
    if (pxw == 256) na.co = 0;
    const size_t wgs = ((size_t)ca.Q + pxw - 1) / pxw;
    const size_t edge_floats = (wgs * 4 * FUSE_EW * 4 + 3) & ~(size_t)3;
    const size_t nld = wgs * (pxw / 32);                                  // one fp64 log-det partial per wave (32 pixels), idle waves of a ragged last workgroup included
    if ((edge_floats + 2 * nld + 4) <= 4 * h->pstride) {                // (the scratch lives in the P buffers the fused launch does not use)
      na.fuse = 1;
      na.fz_b3 = ca.b3; na.fz_A = ca.A; na.fz_b = ca.b; na.fz_out = ca.out; na.fz_out_stride = ca.out_stride; na.fz_out_off = ca.out_off;
      na.fz_inverse = ca.inverse;
      na.fz_osave = ca.o_save;
      na.fz_edge = h->bufP;
      na.fz_ldpart = ca.logdet ? reinterpret_cast<double*>(h->bufP + edge_floats) : nullptr;
    }
  }
  int np = 1;
  if (int rc = launch_net(h, lvl, c, F, na, s, mode, &np)) return rc;
  if (np != 100 && np != 101) {
    ca.P = na.P; ca.np = np; ca.pstride = na.pstride;
    return launch_couple(c, ca, N, s, fl, flat_used);
  }
  ++h->fused_steps;
  const int pxw = np == 101 ? CO_PX : 256;
  if (hw > pxw || ca.logdet) {
    EdgeArgs ea;
    ea.vin = ca.vin; ea.edge = na.fz_edge; ea.ldpart = na.fz_ldpart; ea.A = ca.A; ea.b = ca.b; ea.out = ca.out; ea.out_stride = ca.out_stride;
    ea.out_off = ca.out_off; ea.inverse = ca.inverse; ea.logdet = ca.logdet; ea.osave = ca.o_save; ea.h = ca.h; ea.w = ca.w; ea.flag = ca.flag; ea.pxw = pxw;
    hipLaunchKernelGGL(k_couple_edge, dim3(N), dim3(256), 0, s, ea);
    LAUNCHCHK("k_couple_edge");
  }
  return 0;
}

PreArgs pre_args(const glowk_config& cfg) {
  PreArgs p;
  p.minval = cfg.minval; p.maxval = cfg.maxval; p.alpha = cfg.alpha; p.use_logit = cfg.use_logit;
  return p;
}

// ---- device memory of a handle: ONE description of every buffer, used by the allocators and by glowk_workspace_bytes ----
struct WsSizes {      // forward / inverse workspace for N tiles (bytes)
  size_t act, P, ld;  // bufA, bufB, bufZ (each) | bufP (4 partial per-tap buffers) | bufLd
  size_t slots;       // bufLdSlot: per-sample log-det slots of the flat-grid coupling kernel (small batches only)
  size_t total() const { return 3 * act + P + ld + slots; }
};
// log-det slots per sample: one per flow step and 16 pixels of its level
int ld_slots_per_sample(const glowk_handle* h) {
  int n = 0;
  for (const Level& lv : h->levels) n += h->cfg.K * ((lv.h * lv.w + 15) / 16);
  return n;
}
WsSizes ws_sizes(const glowk_handle* h, size_t N) {
  const size_t E = (size_t)h->cfg.H * h->cfg.W * h->cfg.C;
  return WsSizes{N * E * 4, 4 * 9 * N * E * 4, N * 8, std::min(N, (size_t)2 * num_cus()) * ld_slots_per_sample(h) * 8};
}
// pixel blocks of the ReLU-mask arrays of a level with Q pixels (one 16-bit entry per lane, hidden block and pixel block of a wave):
// 32-pixel blocks of whole 256-pixel workgroups -- or 16-pixel blocks of 128-pixel workgroups where the launch policy may pick the
// half-wave form of the 16x16x32 family (glowk_launch.h: use_half -- small grids, and the 32-channel level at any size)
// MONOTONE in Q (round-3 advisor): the per-step offsets (offM) are laid out once for the largest batch a handle has seen and a later,
// smaller call computes its own mask2 = mask1 + blocks inside that layout -- so a smaller Q must never need MORE blocks.  The
// half-wave form needs ceil(Q / 128) * 8 blocks up to the largest Q the policy picks it for (Qh = 256 * floor(CUs / 8) pixels) and
// the 256-pixel form ceil(Q / 256) * 8 beyond: the bound is the larger of the latter and the former capped at Qh.
size_t mask_blocks(const Level& lv, size_t Q) {
  const size_t b256 = ((Q + 255) / 256) * 8;
  if (lv.c == 32) return ((Q + 127) / 128) * 8;
  const size_t Qh = (size_t)256 * (size_t)(num_cus() / 8);
  return std::max(b256, ((std::min(Q, Qh) + 127) / 128) * 8);
}

struct SaveSizes {    // input-gradient path for N tiles: per-step saves + scratch
  size_t v, p, m;     // floats of saveV, floats of saveP, shorts of saveM
  int np;             // (1: kept for the size formula)
  size_t gz, c;       // bytes of bufGz, bufC
  size_t total() const { return v * 4 + (size_t)np * p * 4 + m * 2 + gz + c; }
};
NetArgs net_args(glowk_handle* h, const Level& lv, const StepDev& sd, const float* vin, int in_stride, int in_off, int N);
SaveSizes save_sizes(const glowk_handle* h, size_t N, std::vector<size_t>* offV = nullptr, std::vector<size_t>* offP = nullptr,
                     std::vector<size_t>* offM = nullptr) {
  const int K = h->cfg.K, L = h->cfg.L, NF = h->cfg.F / 32;
  SaveSizes S{0, 0, 0, 1, 0, 0};
  if (offV) { offV->assign((size_t)L * K, 0); offP->assign((size_t)L * K, 0); offM->assign((size_t)L * K, 0); }
  for (int lvl = 0; lvl < L; ++lvl) {
    const Level& lv = h->levels[lvl];
    const size_t Q = N * lv.h * lv.w;
    const size_t blocks = mask_blocks(lv, Q);
    for (int j = 0; j < K; ++j) {
      const size_t sidx = (size_t)lvl * K + j;
      if (offV) { (*offV)[sidx] = S.v; (*offP)[sidx] = S.p; (*offM)[sidx] = S.m; }
      S.v += Q * lv.c;
      S.p += Q * (lv.c / 2);
      S.m += 2 * blocks * NF * 64;   // mask1 then mask2
    }
  }
  const size_t E = (size_t)h->cfg.H * h->cfg.W * h->cfg.C;
  S.gz = N * E * 4;
  S.c = N * E * 4;
  return S;
}

int ensure_flag(glowk_handle* h) {
  if (h->d_flag) return 0;
  HIPCHK(hipMalloc(&h->d_flag, 16));
  HIPCHK(hipMemset(h->d_flag, 0, 16));
  HIPCHK(hipHostMalloc(&h->h_flag, 16));
  h->h_flag[0] = 0;
  return 0;
}

int ensure_ws(glowk_handle* h, int N) {
  if (int rc = ensure_flag(h)) return rc;
  if (N <= h->wsN) return 0;
  HIPCHK(hipDeviceSynchronize());
  if (h->bufA) { hipFree(h->bufA); hipFree(h->bufB); hipFree(h->bufP); hipFree(h->bufZ); hipFree(h->bufLd); hipFree(h->bufLdSlot); }
  h->bufA = h->bufB = h->bufP = h->bufZ = nullptr; h->bufLd = nullptr; h->bufLdSlot = nullptr; h->wsN = 0;
  const WsSizes W = ws_sizes(h, (size_t)N);
  HIPCHK(hipMalloc(&h->bufA, W.act));
  HIPCHK(hipMalloc(&h->bufB, W.act));
  h->pstride = W.P / 16;          // floats per partial
  HIPCHK(hipMalloc(&h->bufP, W.P));
  HIPCHK(hipMalloc(&h->bufZ, W.act));
  HIPCHK(hipMalloc(&h->bufLd, W.ld));
  HIPCHK(hipMalloc(&h->bufLdSlot, W.slots));
  h->wsN = N;
  return 0;
}

// scratch tensor of the gradient path / the ActNorm initialisation (one level tensor)
int ensure_c(glowk_handle* h, int N) {
  if ((size_t)N <= h->cN) return 0;
  HIPCHK(hipDeviceSynchronize());
  if (h->bufC) hipFree(h->bufC);
  h->bufC = nullptr; h->cN = 0;
  HIPCHK(hipMalloc(&h->bufC, (size_t)N * h->cfg.H * h->cfg.W * h->cfg.C * 4));
  h->cN = (size_t)N;
  return 0;
}

// pixel and element indices of a call are 32-bit in the light kernels: one call takes at most 2^28 elements (1 GiB of tiles,
// 22 GB of workspace); larger batches are the caller's loop (the Python mirror chunks by glowk_max_tiles)
int max_tiles(const glowk_handle* h) {
  const long long E = (long long)h->cfg.H * h->cfg.W * h->cfg.C;
  const long long m = (1LL << 28) / E;
  return m < 1 ? 1 : (int)m;
}

extern "C" int glowk_finalize_weights(glowk_handle* h);

int check_batch(const glowk_handle* h, int N) {
  if (N <= 0) return fail("batch size must be positive");
  if (N > max_tiles(h)) return fail("batch of " + std::to_string(N) + " tiles exceeds glowk_max_tiles() = " + std::to_string(max_tiles(h)) +
                                    " (32-bit indices within one call): split the batch");
  return 0;
}

int check_ready(glowk_handle* h, int N) {
  if (!h) return fail("null handle");
  if (h->finalized && h->split_stale && h->precision != GLOWK_PREC_F32)   // trained since the f16 images were packed: re-pack (host, ~0.3 s)
    if (int rc = glowk_finalize_weights(h)) return rc;
  if (!h->finalized) return fail("glowk_finalize_weights has not been called");
  if (int rc = check_batch(h, N)) return rc;
  return ensure_ws(h, N);
}

NetArgs net_args(glowk_handle* h, const Level& lv, const StepDev& sd, const float* vin, int in_stride, int in_off, int N) {
  NetArgs a;
  a.vin = vin; a.in_stride = in_stride; a.in_off = in_off;
  a.Q = N * lv.h * lv.w; a.h = lv.h; a.w = lv.w;
  a.K1p = sd.K1p; a.ep = sd.ep; a.R0p = sd.R0p; a.mask1 = nullptr; a.mask2 = nullptr; a.P = h->bufP;
  a.RHp = sd.RHp; a.RSp = sd.RSp; a.fam16 = (sd.RSp && sd.RSBp) ? 1 : 0; a.eph = sd.epH; a.pstride = h->pstride; a.max_np = 4; a.sc1 = sd.sc1; a.sc2 = sd.sc2; a.sc3 = sd.sc3;
  a.flag = flagp(h); a.xlim = sd.xlim_f; a.st1 = nullptr; a.st2 = nullptr;
  a.bnorm = 1.0f;
  a.fuse = 0; a.co = glowk_detail::env().co_off ? 0 : 1; a.fz_osave = nullptr; a.fz_b3 = nullptr; a.fz_A = nullptr; a.fz_b = nullptr; a.fz_out = nullptr; a.fz_out_stride = 0; a.fz_out_off = 0; a.fz_inverse = 0;
  a.fz_edge = nullptr; a.fz_ldpart = nullptr;
  a.dbg = glowk_detail::g_dbg_stamps;
  a.xmax_out = h->d_probe ? h->d_probe + ((&lv - h->levels.data()) * h->cfg.K + (&sd - lv.dev.data())) : nullptr;
  return a;
}

// per-step save buffers of the input-gradient path, forward order index sidx = level*K + (K-1-k)
int ensure_save(glowk_handle* h, int N) {
  if (int rc = ensure_c(h, N)) return rc;
  const SaveSizes need = save_sizes(h, (size_t)N);
  if (N <= h->saveN) return 0;
  const int Na = std::max(N, h->saveN);
  HIPCHK(hipDeviceSynchronize());
  if (h->saveV) { hipFree(h->saveV); hipFree(h->saveP); hipFree(h->saveM); hipFree(h->bufGz); }
  h->saveV = h->saveP = h->bufGz = nullptr; h->saveM = nullptr; h->saveN = 0;
  SaveSizes S = save_sizes(h, (size_t)Na, &h->offV, &h->offP, &h->offM);
  HIPCHK(hipMalloc(&h->saveV, S.v * 4));
  HIPCHK(hipMalloc(&h->saveP, S.p * 4));
  HIPCHK(hipMalloc(&h->saveM, S.m * 2));
  HIPCHK(hipMalloc(&h->bufGz, S.gz));
  h->saveN = Na;
  return 0;
}

// data -> latent (+ log-det accumulated in h->bufLd); z_dst [N,Hl,Wl,Cl].  save: keep every step's coupling input v,
// per-tap conv3 outputs P and the two ReLU masks for run_backward.
int run_forward(glowk_handle* h, const float* x, int N, float* z_dst, hipStream_t s, bool save = false, bool keep_hidden = false) {
  const glowk_config& cfg = h->cfg;
  const int K = cfg.K, L = cfg.L, NF = cfg.F / 32;
  float* cur = save ? h->saveV + h->offV[0] : h->bufA;
  float* oth = h->bufB;
  // small batches: the coupling kernels run on a flat pixel grid and leave their log-det shares in per-workgroup slots (k_couple_flat)
  const int nslots = ld_slots_per_sample(h);
  const bool flat_ok = N < 2 * num_cus() && h->bufLdSlot;
  bool flat_used = false;
  int slot_base = 0;
  if (flat_ok) HIPCHK(hipMemsetAsync(h->bufLdSlot, 0, (size_t)N * nslots * 8, s));
  {
    const Level& lv = h->levels[0];
    const StepDev& first = lv.dev[K - 1];
    CDISPATCH(lv.c, hipLaunchKernelGGL((k_in<CC>), dim3(N), dim3(256), 0, s, x, cfg.H, cfg.W, pre_args(cfg), 1,
                                       first.Afwd, first.bfwd, cur, h->bufLd, h->ld_const + h->ld_pre_const));
    LAUNCHCHK("k_in");
  }
  for (int lvl = 0; lvl < L; ++lvl) {
    const Level& lv = h->levels[lvl];
    const size_t Q = (size_t)N * lv.h * lv.w;
    const size_t blocks = mask_blocks(lv, Q);
    for (int k = K - 1; k >= 0; --k) {   // tfb.Chain applies right to left: step K-1 first (flow_glow.py:51-52)
      const StepDev& sd = lv.dev[k];
      const size_t sidx = (size_t)lvl * K + (K - 1 - k);
      NetArgs na = net_args(h, lv, sd, cur, lv.c, lv.c / 2, N);
      if (save) {
        na.mask1 = h->saveM + h->offM[sidx];
        na.mask2 = na.mask1 + blocks * NF * 64;
      }
      int np = 1;   // the f16x3 kernels leave P as np partial sums (one per pass over the hidden width)
      if (keep_hidden) {   // training, exact fp32: this launch also leaves relu(conv1 + b1), relu(conv2 + b2) planar for the weight gradients
        na.st1 = h->trKeep + h->trKeepOff[sidx] * (size_t)N;
        na.st2 = na.st1 + (size_t)cfg.F * Q;
      }
      // the plain forward direction and the saving pass of the split arithmetics: network + coupling may run as one kernel
      const bool plain = !keep_hidden && (!save || h->precision != GLOWK_PREC_F32);
      const int plain_mode = save ? 4 : fwd_mode(h);
      if (!plain) {
        if (int rc = launch_net(h, lvl, lv.c, cfg.F, na, s, keep_hidden ? (h->precision == GLOWK_PREC_F32 ? 9 : 10) : NET_FWD_SAVE, &np)) return rc;
      }
      CoupleArgs ca;
      ca.vin = cur; ca.P = na.P; ca.np = np; ca.pstride = na.pstride; ca.b3 = sd.b3; ca.logdet = h->bufLd; ca.log_s_out = nullptr; ca.t_out = nullptr;
      ca.o_save = save ? h->saveP + h->offP[sidx] : nullptr;
      ca.Q = (int)Q; ca.h = lv.h; ca.w = lv.w; ca.inverse = 0; ca.flag = flagp(h);
      float* next = save && k > 0 ? h->saveV + h->offV[sidx + 1] : oth;
      if (k > 0) {
        ca.A = lv.dev[k - 1].Afwd; ca.b = lv.dev[k - 1].bfwd;
        ca.out = next; ca.out_stride = lv.c; ca.out_off = 0;
      } else if (lvl < L - 1) {
        ca.A = nullptr; ca.b = nullptr; ca.out = oth; ca.out_stride = lv.c; ca.out_off = 0;
      } else {
        ca.A = nullptr; ca.b = nullptr; ca.out = z_dst; ca.out_stride = h->Cl; ca.out_off = lv.z_off;
      }
      const FlatLd fl{flat_ok ? h->bufLdSlot : nullptr, nslots, slot_base};
      slot_base += (lv.h * lv.w + 15) / 16;
      if (plain) { if (int rc = net_and_couple(h, lvl, lv.c, cfg.F, na, ca, N, s, plain_mode, &fl, &flat_used)) return rc; }
      else if (int rc = launch_couple(lv.c, ca, N, s, &fl, &flat_used)) return rc;
      if (k > 0) {
        if (save) cur = next; else std::swap(cur, oth);
      } else if (lvl < L - 1) {
        // block output sits in oth; split + squeeze + first ActNorm/1x1 of the next block
        float* nv = save ? h->saveV + h->offV[sidx + 1] : cur;
        const StepDev& nfirst = h->levels[lvl + 1].dev[K - 1];
        CDISPATCH(lv.c, hipLaunchKernelGGL((k_split<CC>), dim3(N), dim3(256), 0, s, (const float*)oth, lv.h, lv.w, z_dst, h->Hl * h->Wl, h->Cl,
                                           lv.z_off, lv.Cz, nfirst.Afwd, nfirst.bfwd, nv));
        LAUNCHCHK("k_split");
        cur = nv;
        if (!save) oth = (cur == h->bufA) ? h->bufB : h->bufA;
      }
    }
  }
  if (flat_used) {
    hipLaunchKernelGGL(k_ld_fold, dim3(N), dim3(64), 0, s, h->bufLd, (const double*)h->bufLdSlot, nslots);
    LAUNCHCHK("k_ld_fold");
  }
  return 0;
}

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
  const int TM = big8 ? 256 : split ? 128 : big ? 128 : 64, TN = split ? (big ? 128 : 64) : TM;
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
    if (big8 && vec) hipLaunchKernelGGL((k_wgrad_h3<2, 2, 4, 2, true>), grid, dim3(512), 0, s, a);
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

struct TrainCtx {
  float* grad;     // [tr_n] flat gradient vector (device, caller owned)
  float scale;     // every gradient is scale * d sum_n log_prob / d theta
  bool split;      // the sweep runs the fp16-split kernels (k_net_h3, MODE | 8): planar arrays in scaled units (StepGradArgs::scaled)
};

// Weight gradients of the coupling networks of steps k0 .. k0 + nb - 1 of a level (k = backward order), as ONE batch of launches.
// Entry b reads the saved coupling input v + b v_bs [Q][c], the gradient wrt the network output g_o + b go_bs [Q][c], R1 + b r_bs
// (R2 follows at + F Q) and M1 / M2 + b m_bs, all planar [F][Q].
int train_network_grads(glowk_handle* h, TrainCtx* tc, int lvl, int k0, int nb, const float* v, ptrdiff_t v_bs, const float* g_o, ptrdiff_t go_bs,
                        const float* R1, ptrdiff_t r_bs, const float* M1, const float* M2, ptrdiff_t m_bs, int N, hipStream_t s, float bfac = 1.0f) {
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

// d sum_n log_prob[n] / dx, after run_forward(save = true) on the same x; z is that run's latent.
// tc != null: the training sweep -- exact fp32 kernels, and every step also leaves its weight gradients in tc->grad.
int run_backward(glowk_handle* h, const float* x, const float* z, int N, float* dx, hipStream_t s, TrainCtx* tc = nullptr) {
  const glowk_config& cfg = h->cfg;
  const int K = cfg.K, L = cfg.L, NF = cfg.F / 32;
  const int E = h->Hl * h->Wl * h->Cl;
  {
    const size_t total = (size_t)N * E;
    hipLaunchKernelGGL(k_prior_grad, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, total, E, h->d_loc, h->d_log_scale, h->bufGz);
    LAUNCHCHK("k_prior_grad");
  }
  float* gh_a = h->bufA;    // [g_va, g_yb] ping-pong
  float* gh_b = h->bufB;
  float* g_o = h->bufC;     // gradient wrt the network output
  float* Pg = h->bufP;      // per-tap partial input gradients
  float* gu = h->bufZ;      // block-level results (g_u of a block / g_y of the previous block's last step)
  const float* gy_src = nullptr;   // upstream gradient wrt the current block's output o (set per block)
  for (int lvl = L - 1; lvl >= 0; --lvl) {
    const Level& lv = h->levels[lvl];
    const int Q = N * lv.h * lv.w;
    const size_t blocks = mask_blocks(lv, (size_t)Q);
    int npg = 1;                    // partials of Pg the previous network launch of this level left in bufP
    // split training sweep: g_o (and with it everything linear in it: the backward network, its stored hiddens, the im2col of g_o,
    // the weight-gradient GEMMs) is carried times a power of two sized on the previous sweep's gradient magnitudes
    const float bfac = (tc && tc->split && !h->tr_bfac.empty()) ? h->tr_bfac[lvl] : 1.0f;
    // training sweep: per-step slots of g_o, g_v, M1 / M2 when the level's weight-gradient work runs as one batch after its sweep
    const bool level_batch = tc && h->trNB == K && K > 1;
    const size_t go_slot = level_batch ? (size_t)Q * lv.c : 0, m_slot = level_batch ? (size_t)cfg.F * Q : 0;
    const ptrdiff_t v_bs = K > 1 ? (ptrdiff_t)h->offV[(size_t)lvl * K] - (ptrdiff_t)h->offV[(size_t)lvl * K + 1] : 0;   // saved inputs: forward order
    for (int k = 0; k < K; ++k) {   // reverse of the forward order K-1 .. 0
      const StepDev& sd = lv.dev[k];
      const size_t sidx = (size_t)lvl * K + (K - 1 - k);
      BwdArgs ba;
      ba.Q = Q; ba.h = lv.h; ba.w = lv.w; ba.flag = flagp(h);
      ba.go_scale = bfac; ba.pg_scale = 1.0f / bfac; ba.gmax = (tc && h->tr_gmax) ? h->tr_gmax + 16 * lvl : nullptr;
      ba.v = h->saveV + h->offV[sidx]; ba.osave = h->saveP + h->offP[sidx];
      float* go_k = tc ? h->trGo + (size_t)k * go_slot : g_o;
      ba.g_o = go_k; ba.ghalf_out = gh_b; ba.gu_out = nullptr;
      if (k == 0) {
        // gradient wrt the block output: the latent slice itself (last block) or what k_bwd_split assembled
        ba.ghalf_in = nullptr; ba.Pg = nullptr; ba.npg = 1; ba.pgstride = 0; ba.A = nullptr;
        if (lvl == L - 1) { ba.gv_direct = h->bufGz; ba.gvd_stride = h->Cl; ba.gvd_off = lv.z_off; }
        else              { ba.gv_direct = gy_src;   ba.gvd_stride = lv.c;  ba.gvd_off = 0; }
      } else {
        // merge step k-1's network gradient, go through its fused ActNorm + 1x1, then this step's coupling
        ba.ghalf_in = gh_a; ba.Pg = Pg; ba.npg = npg; ba.pgstride = h->pstride; ba.gv_direct = nullptr; ba.gvd_stride = 0; ba.gvd_off = 0;
        ba.A = lv.dev[k - 1].Afwd;
      }
      ba.gv_out = (tc && k > 0) ? h->trGv + (size_t)(k - 1) * go_slot : nullptr;
      if (int rc = launch_bwd_light(lv.c, ba, N, s)) return rc;
      if (tc && k > 0 && !level_batch)   // g_v of step k-1 is complete: its ActNorm + 1x1 gradient sums
        if (int rc = train_affine_sums(h, lvl, k - 1, 1, h->saveV + h->offV[sidx + 1], 0, h->trGv, 0, N, s)) return rc;
      std::swap(gh_a, gh_b);   // gh_a now holds this step's [g_va, g_yb]
      NetArgs na = net_args(h, lv, sd, go_k, lv.c, 0, N);
      na.K1p = sd.K3bp; na.R0p = sd.RBp; na.P = Pg;
      na.mask1 = h->saveM + h->offM[sidx];
      na.mask2 = na.mask1 + blocks * NF * 64;
      const bool h3b = (tc ? tc->split : h->precision != GLOWK_PREC_F32) && (sd.RHBp || sd.RSBp);   // (c = 32: 16x16x32 image only)
      if (h3b) {
        na.RHp = sd.RHBp; na.RSp = sd.RSBp; na.eph = nullptr; na.sc1 = sd.scb1; na.sc2 = sd.scb2; na.sc3 = sd.scb3; na.xlim = sd.xlim_b;
        // the backward network is linear, so the kernels normalise every pixel's gradient vector to [T, 2T) (a power of two, exact;
        // glowk_kernels.h: pixel_norm): T = the largest power of two for which the host's worst-case bound (xlim_b: L1 norms of the
        // folded transposed weights) rules an fp16 overflow out.  Weights so large that even T = 2^-4 is not safe (the low bits of
        // the smaller components would go) are handled like a tripped guard: the call's range policy decides.
        na.bnorm = tc ? 1.0f : bwd_norm_target(sd.xlim_b);      // (training: one scale per launch instead, see bfac)
        if (na.bnorm == 0.0f) {
          na.bnorm = 1.0f;
          if (na.flag) HIPCHK(hipMemsetAsync(na.flag, 1, sizeof(int), s));
        }
      }
      if (na.xmax_out) na.xmax_out += (size_t)cfg.L * cfg.K;     // (range probe: the backward launches' half)
      if (tc) { na.st1 = h->trM2 + (size_t)k * m_slot; na.st2 = h->trM1 + (size_t)k * m_slot; }
      if (int rc = launch_net(h, lvl, lv.c, cfg.F, na, s, tc ? (tc->split ? 11 : 8) : h3b ? 5 : NET_BWD, &npg)) return rc;
      if (tc && !level_batch) {
        // R1 / R2: kept by the saving forward pass, or recomputed now from the saved input (the P output of that launch goes to a
        // scratch partial of bufP)
        const float* R1 = h->trR1;
        if (h->trKeep && N <= h->trKeepN) R1 = h->trKeep + h->trKeepOff[sidx] * (size_t)N;   // (blocks are per tile: they scale with the batch)
        else {
          NetArgs nf = net_args(h, lv, sd, h->saveV + h->offV[sidx], lv.c, lv.c / 2, N);
          nf.P = h->bufP + 2 * h->pstride; nf.st1 = h->trR1; nf.st2 = h->trR1 + (size_t)cfg.F * Q;
          if (launch_net_raw(lv.c, cfg.F, nf, 7, s) < 0) return 1;
        }
        if (int rc = train_network_grads(h, tc, lvl, k, 1, h->saveV + h->offV[sidx], 0, go_k, 0, R1, 0, h->trM1, h->trM2, 0, N, s, bfac)) return rc;
      }
    }
    // first forward step of the block (k = K-1): merge, then through its ActNorm + 1x1 -> g_u of the squeezed block input
    {
      BwdArgs ba;
      ba.Q = Q; ba.h = lv.h; ba.w = lv.w; ba.flag = flagp(h);
      ba.go_scale = 1.0f; ba.pg_scale = 1.0f / bfac; ba.gmax = nullptr;
      ba.ghalf_in = gh_a; ba.Pg = Pg; ba.npg = npg; ba.pgstride = h->pstride; ba.gv_direct = nullptr; ba.gvd_stride = 0; ba.gvd_off = 0;
      ba.A = lv.dev[K - 1].Afwd;
      ba.v = nullptr; ba.osave = nullptr; ba.g_o = nullptr; ba.ghalf_out = nullptr; ba.gu_out = g_o;   // reuse g_o as g_u
      ba.gv_out = tc ? h->trGv + (size_t)(K - 1) * go_slot : nullptr;
      if (int rc = launch_bwd_light(lv.c, ba, N, s)) return rc;
      if (tc && !level_batch)
        if (int rc = train_affine_sums(h, lvl, K - 1, 1, h->saveV + h->offV[(size_t)lvl * K], 0, h->trGv, 0, N, s)) return rc;
    }
    if (level_batch) {
      // the level's sweep is over: weight gradients of its K steps in one batch of launches.  Step k (backward order) saved its input at
      // forward position K-1-k (v_bs < 0) and kept R1 / R2 there; M1 / M2, g_o, g_v sit in slot k.
      const float* v0 = h->saveV + h->offV[(size_t)lvl * K + (K - 1)];
      const float* R10 = h->trKeep + h->trKeepOff[(size_t)lvl * K + (K - 1)] * (size_t)N;
      const ptrdiff_t r_bs = -(ptrdiff_t)(2 * (size_t)cfg.F * Q);
      if (int rc = train_network_grads(h, tc, lvl, 0, K, v0, v_bs, h->trGo, (ptrdiff_t)go_slot, R10, r_bs, h->trM1, h->trM2, (ptrdiff_t)m_slot, N, s, bfac)) return rc;
      if (int rc = train_affine_sums(h, lvl, 0, K, v0, v_bs, h->trGv, (ptrdiff_t)go_slot, N, s)) return rc;
    }
    if (lvl > 0) {
      const Level& pv = h->levels[lvl - 1];
      CDISPATCH(pv.c, hipLaunchKernelGGL((k_bwd_split<CC>), dim3(N), dim3(256), 0, s, (const float*)h->bufGz, h->Hl * h->Wl, h->Cl, pv.z_off,
                                         pv.Cz, (const float*)g_o, pv.h, pv.w, gu));
      LAUNCHCHK("k_bwd_split");
      gy_src = gu;
    } else {
      CDISPATCH(lv.c, hipLaunchKernelGGL((k_bwd_in<CC>), dim3(N), dim3(256), 0, s, (const float*)g_o, x, lv.h, lv.w, pre_args(cfg), dx));
      LAUNCHCHK("k_bwd_in");
    }
  }
  return 0;
}

// latent -> data
int run_inverse(glowk_handle* h, const float* z, int N, float* x, hipStream_t s) {
  const glowk_config& cfg = h->cfg;
  const int K = cfg.K, L = cfg.L;
  float* cur = h->bufA;
  float* oth = h->bufB;
  for (int lvl = L - 1; lvl >= 0; --lvl) {
    const Level& lv = h->levels[lvl];
    const float* unext = (lvl == L - 1) ? nullptr : cur;
    CDISPATCH(lv.c, hipLaunchKernelGGL((k_unsplit<CC>), dim3(N), dim3(256), 0, s, z, h->Hl * h->Wl, h->Cl, lv.z_off, lv.Cz, unext,
                                       lv.h, lv.w, oth));
    LAUNCHCHK("k_unsplit");
    std::swap(cur, oth);
    for (int k = 0; k < K; ++k) {   // Chain.inverse: step 0 first
      const StepDev& sd = lv.dev[k];
      CoupleArgs ca;
      ca.vin = cur; ca.P = h->bufP; ca.np = 1; ca.pstride = h->pstride; ca.b3 = sd.b3; ca.logdet = nullptr; ca.log_s_out = nullptr; ca.t_out = nullptr; ca.o_save = nullptr;
      ca.Q = N * lv.h * lv.w; ca.h = lv.h; ca.w = lv.w; ca.inverse = 1; ca.flag = flagp(h);
      ca.A = sd.Ainv; ca.b = sd.binv; ca.out = oth; ca.out_stride = lv.c; ca.out_off = 0;
      if (int rc = net_and_couple(h, lvl, lv.c, cfg.F, net_args(h, lv, sd, cur, lv.c, lv.c / 2, N), ca, N, s, fwd_mode(h))) return rc;
      std::swap(cur, oth);
    }
  }
  const Level& l0 = h->levels[0];
  CDISPATCH(l0.c, hipLaunchKernelGGL((k_out<CC>), dim3(N), dim3(256), 0, s, cur, l0.h, l0.w, pre_args(cfg), 1, x));
  LAUNCHCHK("k_out");
  return 0;
}

// device that owns a device pointer (the handle-free entry points launch there); the current device if HIP cannot tell
int ptr_device(const void* p) {
  hipPointerAttribute_t a;
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (hipPointerGetAttributes(&a, p) == hipSuccess && a.device >= 0) return a.device;
  (void)hipGetLastError();
  return cur;
}

// The range guard around one compute call (include/glowk.h: glowk_range_policy).  `run` issues the call's launches.
template <class Run>
int guarded(glowk_handle* h, hipStream_t s, Run&& run) {
  if (int rc = run()) return rc;
  if (h->precision == GLOWK_PREC_F32 || h->range_policy == GLOWK_RANGE_IGNORE) return 0;
  HIPCHK(hipMemcpyAsync(h->h_flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (!h->h_flag[0]) return 0;
  h->h_flag[0] = 0;
  HIPCHK(hipMemsetAsync(h->d_flag, 0, sizeof(int), s));
  if (h->range_policy == GLOWK_RANGE_ERROR) {
    g_err = "a hidden activation left the fp16 range of the split arithmetic (|a| >= 16 376) or the input is not finite: "
            "the outputs of this call are not usable -- use GLOWK_PREC_F32 or GLOWK_RANGE_FALLBACK";
    return GLOWK_ERR_RANGE;
  }
  const int prec = h->precision;   // FALLBACK: the same call on the exact kernels, in-process
  h->precision = GLOWK_PREC_F32;
  ++h->range_fallbacks;
  const int rc = run();
  h->precision = prec;
  return rc;
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

}  // namespace

// =================================================================================================
extern "C" {

int glowk_version(void) { return GLOWK_VERSION; }
void glowk_reload_env(void) { glowk_detail::reload_env(); }

int glowk_debug_stamps(unsigned long long* out, int n) {
  if (n < 0 || n > 64 || (n && !out)) return fail("glowk_debug_stamps: n must be in [0, 64]");
  if (!glowk_detail::g_dbg_stamps) {
    HIPCHK(hipMalloc(&glowk_detail::g_dbg_stamps, 64 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(glowk_detail::g_dbg_stamps, 0, 64 * sizeof(unsigned long long)));
  }
  HIPCHK(hipDeviceSynchronize());
  if (n) HIPCHK(hipMemcpy(out, glowk_detail::g_dbg_stamps, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
}
const char* glowk_last_error(void) { return g_err.c_str(); }

int glowk_create(const glowk_config* cfg, int device, glowk_handle** out) {
  if (!cfg || !out) return fail("null argument");
  if (cfg->L < 2 || cfg->L > 4) return fail("L should be 2, 3 or 4");  // flow_builder.py:76-77
  const int s = 1 << cfg->L;
  if (cfg->H <= 0 || cfg->W <= 0 || cfg->C <= 0 || cfg->H % s || cfg->W % s) return fail("H and W must be positive multiples of 2^L");
  if (cfg->K <= 0) return fail("K must be positive");
  if (cfg->F != 128 && cfg->F != 256 && cfg->F != 384 && cfg->F != 512) return fail("n_filters must be 128, 256, 384 or 512 (the instantiated coupling-network widths)");
  glowk_handle* h = new glowk_handle();
  h->cfg = *cfg;
  h->device = device;
  h->Hl = cfg->H / s; h->Wl = cfg->W / s; h->Cl = cfg->C * s * s;
  int hh = cfg->H, ww = cfg->W, cc = cfg->C, off = 0, rem = h->Cl;
  for (int l = 0; l < cfg->L; ++l) {
    Level lv;
    hh /= 2; ww /= 2; cc *= 4;
    lv.h = hh; lv.w = ww; lv.c = cc;
    if (l < cfg->L - 1) {
      lv.z_width = rem / 2; lv.z_off = off; off += lv.z_width; rem -= lv.z_width;
      lv.Cz = (hh * ww * (cc / 2)) / (h->Hl * h->Wl);
    } else {
      lv.z_width = rem; lv.z_off = off; lv.Cz = 0;
    }
    for (int t = 0; t < GLOWK_NUM_STEP_TENSORS; ++t) {
      lv.host[t].resize(cfg->K);
      for (int k = 0; k < cfg->K; ++k) lv.host[t][k].assign(step_tensor_size(*cfg, lv, t), 0.0f);
    }
    lv.dev.resize(cfg->K);
    h->levels.push_back(std::move(lv));
    cc /= 2;
  }
  const size_t E = (size_t)h->Hl * h->Wl * h->Cl;
  h->prior_loc.assign(E, 0.0f);
  h->prior_log_scale.assign(E, 0.0f);
  *out = h;
  return 0;
}

int glowk_destroy(glowk_handle* h) {
  if (!h) return 0;
  DeviceGuard dg(h->device);
  if (h->arena) hipFree(h->arena);
  if (h->d_flag) hipFree(h->d_flag);
  if (h->d_probe) hipFree(h->d_probe);
  if (h->tr_gmax) hipFree(h->tr_gmax);
  if (h->h_gmax) hipHostFree(h->h_gmax);
  if (h->h_flag) hipHostFree(h->h_flag);
  {
    void* tr[] = {h->tr_params, h->tr_m, h->tr_v, h->trR1, h->trR2, h->trM1, h->trM2, h->trXcol, h->trGcol, h->trCpart, h->trC1, h->trC2, h->trC3,
                  h->trGv, h->trGo, h->trAffPart, h->trAffSum, h->trSmall, h->trKeep, h->tr16_src, h->tr16_S, h->tr16_scales};
    for (void* p : tr) if (p) hipFree(p);
    for (int* m : h->tr_map) if (m) hipFree(m);
    for (int* m : h->tr_map16) if (m) hipFree(m);
  }
  if (h->bufA) { hipFree(h->bufA); hipFree(h->bufB); hipFree(h->bufP); hipFree(h->bufZ); hipFree(h->bufLd); hipFree(h->bufLdSlot); }
  if (h->bufC) hipFree(h->bufC);
  if (h->bufStat) hipFree(h->bufStat);
  if (h->saveV) { hipFree(h->saveV); hipFree(h->saveP); hipFree(h->saveM); hipFree(h->bufGz); }
  for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
  for (hipEvent_t e : h->tr_events) hipEventDestroy(e);
  if (h->tr_pinned) hipHostFree(h->tr_pinned);
  for (hipStream_t t : h->tr_streams) hipStreamDestroy(t);
  delete h;
  return 0;
}

size_t glowk_tensor_size(const glowk_handle* h, int level, int tensor_id) {
  if (!h) return 0;
  if (tensor_id == GLOWK_PRIOR_LOC || tensor_id == GLOWK_PRIOR_LOG_SCALE) return h->prior_loc.size();
  if (level < 0 || level >= (int)h->levels.size() || tensor_id < 0 || tensor_id >= GLOWK_NUM_STEP_TENSORS) return 0;
  return step_tensor_size(h->cfg, h->levels[level], tensor_id);
}

static std::vector<float>* locate(glowk_handle* h, int level, int step, int id) {
  if (id == GLOWK_PRIOR_LOC) return &h->prior_loc;
  if (id == GLOWK_PRIOR_LOG_SCALE) return &h->prior_log_scale;
  if (level < 0 || level >= (int)h->levels.size() || id < 0 || id >= GLOWK_NUM_STEP_TENSORS) return nullptr;
  if (step < 0 || step >= h->cfg.K) return nullptr;
  return &h->levels[level].host[id][step];
}

int glowk_set_tensor(glowk_handle* h, int level, int step, int tensor_id, const float* host, size_t n) {
  if (!h || !host) return fail("null argument");
  if (h->host_stale) { DeviceGuard dg(h->device); if (int rc = sync_host(h)) return rc; }
  h->tr_active = false;      // the device master copy of a training run no longer holds the current parameters
  std::vector<float>* v = locate(h, level, step, tensor_id);
  if (!v) return fail("no such tensor");
  if (v->size() != n) return fail("tensor size mismatch: expected " + std::to_string(v->size()) + ", got " + std::to_string(n));
  std::memcpy(v->data(), host, n * sizeof(float));
  h->finalized = false;
  return 0;
}

int glowk_get_tensor(const glowk_handle* h, int level, int step, int tensor_id, float* host, size_t n) {
  if (!h || !host) return fail("null argument");
  if (h->host_stale) { DeviceGuard dg(h->device); if (int rc = sync_host(const_cast<glowk_handle*>(h))) return rc; }
  const std::vector<float>* v = locate(const_cast<glowk_handle*>(h), level, step, tensor_id);
  if (!v) return fail("no such tensor");
  if (v->size() != n) return fail("tensor size mismatch");
  std::memcpy(host, v->data(), n * sizeof(float));
  if (tensor_id == GLOWK_INV1X1_P_INV) {   // never loaded: what the reference initialises it to, inv(P)
    bool set = false;
    for (size_t i = 0; i < n; ++i) set |= host[i] != 0.0f;
    if (!set) {
      const int c = h->levels[level].c;
      const std::vector<float>& P = h->levels[level].host[GLOWK_INV1X1_P][step];
      Mat Pm(P.begin(), P.end()), Pi;
      if (invert(Pm, c, Pi))
        for (size_t i = 0; i < n; ++i) host[i] = (float)Pi[i];
    }
  }
  return 0;
}

int glowk_finalize_weights(glowk_handle* h) {
  if (!h) return fail("null handle");
  if (h->host_stale) { DeviceGuard dg0(h->device); if (int rc = sync_host(h)) return rc; }
  h->split_stale = false;
  const glowk_config& cfg = h->cfg;
  for (const Level& lv : h->levels)
    if (lv.c != 4 && lv.c != 8 && lv.c != 16 && lv.c != 32)
      return fail("unsupported channel count " + std::to_string(lv.c) + " (this build: 4, 8, 16, 32)");
  size_t total = 0;
  for (const Level& lv : h->levels) total += step_layout(lv.c, cfg.F).total * cfg.K;
  const size_t E = h->prior_loc.size();
  const size_t prior_off = total;
  total += 2 * pad4(E);
  std::vector<float> stage(total, 0.0f);
  h->ld_step.assign((size_t)cfg.L * cfg.K, 0.0);
  h->ld_const = 0.0;
  // every step packs into its own block of the staging arena: steps are packed by a few host threads
  std::vector<PackJob> jobs;
  std::vector<size_t> offs;
  {
    size_t o = 0;
    for (size_t l = 0; l < h->levels.size(); ++l)
      for (int k = 0; k < cfg.K; ++k) {
        jobs.push_back(PackJob{l, k, o, 0.0, {1, 1, 1, 1, 1, 1, 0, 0}, std::string(), false});
        offs.push_back(o);
        o += step_layout(h->levels[l].c, cfg.F).total;
      }
  }
  {
    unsigned nthr = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) nthr = std::min<unsigned>(nthr ? nthr : 1, (unsigned)CPU_COUNT(&set));
    pack_all_steps(cfg, h->levels, stage.data(), jobs, std::min<unsigned>(nthr, 16u));
  }
  for (const PackJob& jb : jobs) {
    if (!jb.ok) return fail("level " + std::to_string(jb.l) + " step " + std::to_string(jb.k) + ": " + jb.err);
    StepDev& d = h->levels[jb.l].dev[jb.k];
    h->ld_step[jb.l * cfg.K + jb.k] = jb.ldc;
    d.sc1 = jb.sc[0]; d.sc2 = jb.sc[1]; d.sc3 = jb.sc[2];
    d.scb1 = jb.sc[3]; d.scb2 = jb.sc[4]; d.scb3 = jb.sc[5];
    d.xlim_f = jb.sc[6]; d.xlim_b = jb.sc[7];
    h->ld_const += jb.ldc;
  }
  std::memcpy(stage.data() + prior_off, h->prior_loc.data(), E * 4);
  std::memcpy(stage.data() + prior_off + pad4(E), h->prior_log_scale.data(), E * 4);
  // data-independent preprocessing log-det (flow_tfp_bijectors.py:390-396)
  const double npx = (double)cfg.H * cfg.W * cfg.C;
  h->ld_pre_const = -npx * std::log((double)cfg.maxval - (double)cfg.minval);
  if (cfg.use_logit) h->ld_pre_const += npx * std::log(1.0 - 2.0 * (double)cfg.alpha);

  DeviceGuard dg(h->device);
  HIPCHK(hipDeviceSynchronize());
  if (h->arena) { hipFree(h->arena); h->arena = nullptr; }
  HIPCHK(hipMalloc(&h->arena, total * 4));
  HIPCHK(hipMemcpy(h->arena, stage.data(), total * 4, hipMemcpyHostToDevice));
  size_t idx = 0;
  for (Level& lv : h->levels) {
    const StepLayout SL = step_layout(lv.c, cfg.F);
    for (int k = 0; k < cfg.K; ++k) {
      StepDev& d = lv.dev[k];
      d.arena_off = offs[idx];
      const float* base = h->arena + offs[idx++];
      d.K1p = base + SL.K1p; d.ep = base + SL.ep;
      d.R0p = reinterpret_cast<const float4*>(base + SL.R0p);
      d.RHp = SL.slotH ? reinterpret_cast<const float4*>(base + SL.RHp) : nullptr;
      d.epH = (SL.slotH || SL.slotS) ? base + SL.epH : nullptr;
      d.RSp = SL.slotS ? reinterpret_cast<const float4*>(base + SL.RSp) : nullptr;
      d.RSBp = SL.slotSB ? reinterpret_cast<const float4*>(base + SL.RSBp) : nullptr;
      d.RHBp = SL.slotHB ? reinterpret_cast<const float4*>(base + SL.RHBp) : nullptr;
      d.K3bp = base + SL.K3bp;
      d.RBp = reinterpret_cast<const float4*>(base + SL.RBp);
      d.Afwd = base + SL.Afwd; d.bfwd = base + SL.bfwd; d.Ainv = base + SL.Ainv; d.binv = base + SL.binv; d.b3 = base + SL.b3;
    }
  }
  h->d_loc = cfg.learntop ? h->arena + prior_off : nullptr;
  h->d_log_scale = cfg.learntop ? h->arena + prior_off + pad4(E) : nullptr;
  h->finalized = true;
  return 0;
}

namespace {
constexpr int STAT_BLOCKS = 256;

// one flow step on a materialised tensor: u (cur) -> ActNorm+1x1 -> tmp; network; coupling -> cur
int run_step_inplace(glowk_handle* h, int lvl, int k, float* cur, float* tmp, int Nl, hipStream_t s) {
  const Level& lv = h->levels[lvl];
  const StepDev& sd = lv.dev[k];
  const int Q = Nl * lv.h * lv.w;
  CDISPATCH(lv.c, hipLaunchKernelGGL((k_affine<CC>), dim3((Q + 255) / 256), dim3(256), 0, s, (const float*)cur, Q, sd.Afwd, sd.bfwd, tmp));
  LAUNCHCHK("k_affine");
  NetArgs na = net_args(h, lv, sd, tmp, lv.c, lv.c / 2, Nl);
  if (int rc = launch_net(h, lvl, lv.c, h->cfg.F, na, s)) return rc;
  CoupleArgs ca;
  ca.o_save = nullptr;
  ca.vin = tmp; ca.P = h->bufP; ca.np = 1; ca.pstride = 0; ca.b3 = sd.b3; ca.A = nullptr; ca.b = nullptr;
  ca.out = cur; ca.out_stride = lv.c; ca.out_off = 0;
  ca.logdet = nullptr; ca.log_s_out = nullptr; ca.t_out = nullptr;
  ca.Q = Q; ca.h = lv.h; ca.w = lv.w; ca.inverse = 0; ca.flag = nullptr;
  return launch_couple(lv.c, ca, Nl, s);
}

// re-fold ActNorm+1x1 of one step after its ActNorm tensors changed and upload the small affine block
int refresh_step_affine(glowk_handle* h, int lvl, int k, hipStream_t s) {
  Level& lv = h->levels[lvl];
  const StepLayout SL = step_layout(lv.c, h->cfg.F);
  std::vector<float> tmp(SL.total - SL.Afwd + SL.Afwd, 0.0f);   // pack_affine indexes from the block start
  std::string err;
  double ldc = 0;
  if (!pack_affine(h->cfg, lv, k, tmp.data(), &ldc, &err)) return fail(err);
  h->ld_step[(size_t)lvl * h->cfg.K + k] = ldc;
  HIPCHK(hipMemcpyAsync(h->arena + lv.dev[k].arena_off + SL.Afwd, tmp.data() + SL.Afwd, (SL.total - SL.Afwd) * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}
}  // namespace

int glowk_actnorm_data_init(glowk_handle* h, const float* x_dev, int N, int runtime_order, int raw_minibatch_quirk, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (int rc = check_ready(h, N)) return rc;
  if (!x_dev) return fail("null tensor");
  hipStream_t s = (hipStream_t)stream;
  const glowk_config& cfg = h->cfg;
  const int K = cfg.K, L = cfg.L;
  const size_t E = (size_t)cfg.H * cfg.W * cfg.C;
  if (int rc = ensure_c(h, N)) return rc;
  if (h->host_stale) if (int rc = sync_host(h)) return rc;
  h->tr_active = false;
  if (!h->bufStat) HIPCHK(hipMalloc(&h->bufStat, (size_t)(STAT_BLOCKS + 1) * 32 * 8));
  // y = SpecPreprocessing.forward(minibatch) (flow_builder.py:121), kept in bufZ
  hipLaunchKernelGGL(k_pre_only, dim3(N), dim3(256), 0, s, x_dev, (int)E, pre_args(cfg), 0, h->bufZ, (float*)nullptr, 0.0);
  LAUNCHCHK("k_pre_only");
  PreArgs nopre = {0, 1, 0, 0};
  float* blk_in = h->bufC;      // input of the current block (squeezed), kept for the block forward
  float* cur = h->bufA;
  float* tmp = h->bufB;
  const bool quirk = raw_minibatch_quirk && L > 2;
  std::vector<double> part((size_t)STAT_BLOCKS * 32), mean(32), var(32);
  for (int lvl = 0; lvl < L; ++lvl) {
    Level& lv = h->levels[lvl];
    int Nl = N;
    if (lvl == 0) {
      CDISPATCH(lv.c, hipLaunchKernelGGL((k_in<CC>), dim3(N), dim3(256), 0, s, (const float*)h->bufZ, cfg.H, cfg.W, nopre, 0,
                                         (const float*)nullptr, (const float*)nullptr, blk_in, (double*)nullptr, 0.0));
      LAUNCHCHK("k_in");
    } else if (quirk) {
      // Squeeze([2h,2w,c/4]).forward(raw minibatch): reshape(-1, ...) reinterprets it as more, smaller samples
      Nl = (int)((size_t)N * E / ((size_t)lv.h * lv.w * lv.c));
      CDISPATCH(lv.c, hipLaunchKernelGGL((k_in<CC>), dim3(Nl), dim3(256), 0, s, (const float*)h->bufZ, 2 * lv.h, 2 * lv.w, nopre, 0,
                                         (const float*)nullptr, (const float*)nullptr, blk_in, (double*)nullptr, 0.0));
      LAUNCHCHK("k_in");
    } else {
      // second half of the previous block's output (held in cur), squeezed (flow_glow.py:96-97)
      const Level& pv = h->levels[lvl - 1];
      CDISPATCH(pv.c, hipLaunchKernelGGL((k_split<CC>), dim3(N), dim3(256), 0, s, (const float*)cur, pv.h, pv.w, (float*)nullptr, 0, 0, 0, 1,
                                         (const float*)nullptr, (const float*)nullptr, blk_in));
      LAUNCHCHK("k_split");
    }
    const int Q = Nl * lv.h * lv.w;
    HIPCHK(hipMemcpyAsync(cur, blk_in, (size_t)Q * lv.c * 4, hipMemcpyDeviceToDevice, s));
    for (int idx = 0; idx < K; ++idx) {
      const int k = runtime_order ? K - 1 - idx : idx;
      double* d_part = h->bufStat;
      double* d_mean = h->bufStat + (size_t)STAT_BLOCKS * 32;
      for (int pass = 0; pass < 2; ++pass) {
        CDISPATCH(lv.c, hipLaunchKernelGGL((k_chan_stats<CC>), dim3(STAT_BLOCKS), dim3(256), 0, s, (const float*)cur, Q,
                                           pass ? (const double*)d_mean : (const double*)nullptr, d_part));
        LAUNCHCHK("k_chan_stats");
        HIPCHK(hipMemcpyAsync(part.data(), d_part, (size_t)STAT_BLOCKS * lv.c * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (int c = 0; c < lv.c; ++c) {
          double t = 0;
          for (int b = 0; b < STAT_BLOCKS; ++b) t += part[(size_t)b * lv.c + c];
          (pass ? var : mean)[c] = t / Q;
        }
        if (!pass) HIPCHK(hipMemcpyAsync(d_mean, mean.data(), lv.c * 8, hipMemcpyHostToDevice, s));
      }
      for (int c = 0; c < lv.c; ++c) {
        // std_init = reduce_std + 1e-8 (the reference adds it in fp32); scale = 1/std; shift = -mean/std
        const double sd_ = (double)((float)std::sqrt(var[c]) + 1e-8f);
        lv.host[GLOWK_ACTNORM_LOG_SCALE][k][c] = (float)std::log(1.0 / sd_);
        lv.host[GLOWK_ACTNORM_SHIFT][k][c] = (float)(-mean[c] / sd_);
      }
      if (int rc = refresh_step_affine(h, lvl, k, s)) return rc;
      if (int rc = run_step_inplace(h, lvl, k, cur, tmp, Nl, s)) return rc;   // minibatch_updated = step.forward(...)
    }
    if (lvl < L - 1 && !quirk && !runtime_order) {
      // glow_block.forward(minibatch): the block as tfb.Chain applies it (K-1 .. 0) on the block's own input
      HIPCHK(hipMemcpyAsync(cur, blk_in, (size_t)Q * lv.c * 4, hipMemcpyDeviceToDevice, s));
      for (int k = K - 1; k >= 0; --k)
        if (int rc = run_step_inplace(h, lvl, k, cur, tmp, Nl, s)) return rc;
    }
  }
  h->ld_const = 0.0;
  for (double v : h->ld_step) h->ld_const += v;
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

int glowk_set_precision(glowk_handle* h, int precision) {
  if (!h) return fail("null handle");
  if (precision != GLOWK_PREC_F32 && precision != GLOWK_PREC_F16X3 && precision != GLOWK_PREC_F16X2) return fail("unknown precision mode");
  h->precision = precision;
  return 0;
}

int glowk_get_precision(const glowk_handle* h) { return h ? h->precision : -1; }

int glowk_set_range_policy(glowk_handle* h, int policy) {
  if (!h) return fail("null handle");
  if (policy != GLOWK_RANGE_IGNORE && policy != GLOWK_RANGE_ERROR && policy != GLOWK_RANGE_FALLBACK) return fail("unknown range policy");
  h->range_policy = policy;
  return 0;
}

int glowk_get_range_policy(const glowk_handle* h) { return h ? h->range_policy : -1; }

int glowk_range_status(glowk_handle* h, int* tripped, int64_t* fallbacks, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (fallbacks) *fallbacks = h->range_fallbacks;
  if (tripped) {
    *tripped = 0;
    if (h->d_flag) {
      hipStream_t s = (hipStream_t)stream;
      HIPCHK(hipMemcpyAsync(h->h_flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      if (h->h_flag[0]) {
        *tripped = 1;
        h->h_flag[0] = 0;
        HIPCHK(hipMemsetAsync(h->d_flag, 0, sizeof(int), s));
      }
    }
  }
  return 0;
}

int glowk_range_probe_begin(glowk_handle* h) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  const size_t n = (size_t)2 * h->cfg.L * h->cfg.K;
  if (!h->d_probe) HIPCHK(hipMalloc(&h->d_probe, n * sizeof(unsigned)));
  HIPCHK(hipMemset(h->d_probe, 0, n * sizeof(unsigned)));
  return 0;
}

int glowk_range_probe_end(glowk_handle* h, float* fwd_ratio, float* bwd_ratio, void* stream) {
  if (!h) return fail("null handle");
  if (!h->d_probe) return fail("glowk_range_probe_end without glowk_range_probe_begin");
  DeviceGuard dg(h->device);
  const int LK = h->cfg.L * h->cfg.K;
  std::vector<float> v((size_t)2 * LK);
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(hipMemcpy(v.data(), h->d_probe, v.size() * sizeof(float), hipMemcpyDeviceToHost));
  hipFree(h->d_probe);
  h->d_probe = nullptr;
  float rf = 0.f, rb = 0.f;
  for (int l = 0; l < h->cfg.L; ++l)
    for (int k = 0; k < h->cfg.K; ++k) {
      const StepDev& sd = h->levels[l].dev[k];
      const int i = l * h->cfg.K + k;
      if (sd.xlim_f > 0.f) rf = std::max(rf, v[i] / sd.xlim_f);
      // (the backward kernels normalise their inputs per pixel: what can fail is the STATIC requirement 2 * 2^-4 <= xlim_b;
      //  reported as that ratio -- the recorded gradient magnitudes v[LK + i] no longer matter for the range)
      if (sd.xlim_b > 0.f && v[LK + i] > 0.f) rb = std::max(rb, getenv("GLOWK_PROBE_RAW_BWD") ? v[LK + i] / sd.xlim_b : 0.125f / sd.xlim_b);   // (raw: the training sweep's uniformly scaled inputs)
    }
  if (fwd_ratio) *fwd_ratio = rf;
  if (bwd_ratio) *bwd_ratio = rb;
  return 0;
}

size_t glowk_workspace_bytes(const glowk_handle* h, int N, int with_grad) {
  if (!h || N <= 0) return 0;
  size_t b = ws_sizes(h, (size_t)N).total() + 16 /* range flag */;
  if (with_grad) b += save_sizes(h, (size_t)N).total();
  return b;
}

int glowk_max_tiles(const glowk_handle* h) { return h ? max_tiles(h) : 0; }

int glowk_reserve(glowk_handle* h, int N, int with_grad) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (int rc = check_batch(h, N)) return rc;
  if (int rc = ensure_ws(h, N)) return rc;
  if (with_grad) {
    if (!h->finalized) return fail("glowk_finalize_weights must run before reserving the gradient path (its size depends on the launch policy)");
    return ensure_save(h, N);
  }
  return 0;
}

int glowk_forward(glowk_handle* h, const float* x_dev, int N, float* z_dev, float* logdet_dev, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (int rc = check_ready(h, N)) return rc;
  if (!x_dev || !z_dev) return fail("null tensor");
  hipStream_t s = (hipStream_t)stream;
  return guarded(h, s, [&]() -> int {
    if (int rc = run_forward(h, x_dev, N, z_dev, s)) return rc;
    if (logdet_dev) {
      hipLaunchKernelGGL(k_prior, dim3(N), dim3(256), 0, s, (const float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr,
                         (const double*)h->bufLd, (float*)nullptr, logdet_dev);
      LAUNCHCHK("k_prior(logdet)");
    }
    return 0;
  });
}

int glowk_inverse(glowk_handle* h, const float* z_dev, int N, float* x_dev, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (int rc = check_ready(h, N)) return rc;
  if (!x_dev || !z_dev) return fail("null tensor");
  hipStream_t s = (hipStream_t)stream;
  return guarded(h, s, [&]() -> int { return run_inverse(h, z_dev, N, x_dev, s); });
}

int glowk_log_prob(glowk_handle* h, const float* x_dev, int N, float* logp_dev, float* z_dev, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (int rc = check_ready(h, N)) return rc;
  if (!x_dev || !logp_dev) return fail("null tensor");
  hipStream_t s = (hipStream_t)stream;
  float* z = z_dev ? z_dev : h->bufZ;
  return guarded(h, s, [&]() -> int {
    if (int rc = run_forward(h, x_dev, N, z, s)) return rc;
    hipLaunchKernelGGL(k_prior, dim3(N), dim3(256), 0, s, (const float*)z, h->Hl * h->Wl * h->Cl, h->d_loc, h->d_log_scale,
                       (const double*)h->bufLd, logp_dev, (float*)nullptr);
    LAUNCHCHK("k_prior");
    return 0;
  });
}

int glowk_log_prob_sum(glowk_handle* h, const float* x_dev, int N, float* logp_dev, float* z_dev, double* sum_dev, int accumulate, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (int rc = check_ready(h, N)) return rc;
  if (!x_dev || !logp_dev || !sum_dev) return fail("null tensor");
  hipStream_t s = (hipStream_t)stream;
  float* z = z_dev ? z_dev : h->bufZ;
  // The reduction runs AFTER the range guard has accepted the call (round-3 advisor): inside the guarded lambda a tripped call's
  // rejected partial sum would already sit in *sum_dev when the fp32 re-run of the FALLBACK policy adds its own (accumulate != 0:
  // double-counted or NaN), and under the ERROR policy *sum_dev would be left modified by a call that reports failure.
  const int rc = guarded(h, s, [&]() -> int {
    if (int rc = run_forward(h, x_dev, N, z, s)) return rc;
    hipLaunchKernelGGL(k_prior, dim3(N), dim3(256), 0, s, (const float*)z, h->Hl * h->Wl * h->Cl, h->d_loc, h->d_log_scale,
                       (const double*)h->bufLd, logp_dev, (float*)nullptr);
    LAUNCHCHK("k_prior");
    return 0;
  });
  if (rc) return rc;
  hipLaunchKernelGGL(k_sum_f64, dim3(1), dim3(1024), 0, s, (const float*)logp_dev, (size_t)N, sum_dev, accumulate, 1.0);
  LAUNCHCHK("k_sum_f64");
  return 0;
}

int glowk_sum_f64(const float* v_dev, size_t n, double* out_dev, int accumulate, double scale, void* stream) {
  if (!out_dev || (!v_dev && n)) return fail("null tensor");
  DeviceGuard dg(ptr_device(out_dev));
  hipLaunchKernelGGL(k_sum_f64, dim3(1), dim3(1024), 0, (hipStream_t)stream, v_dev, n, out_dev, accumulate, scale);
  LAUNCHCHK("k_sum_f64");
  return 0;
}

int glowk_log_prob_grad(glowk_handle* h, const float* x_dev, int N, float* logp_dev, float* dx_dev, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (int rc = check_ready(h, N)) return rc;
  if (!x_dev || !logp_dev || !dx_dev) return fail("null tensor");
  if (int rc = ensure_save(h, N)) return rc;
  hipStream_t s = (hipStream_t)stream;
  float* z = h->bufGz;   // the latent lives in bufGz until the prior gradient overwrites it in place
  return guarded(h, s, [&]() -> int {
    if (int rc = run_forward(h, x_dev, N, z, s, true)) return rc;
    hipLaunchKernelGGL(k_prior, dim3(N), dim3(256), 0, s, (const float*)z, h->Hl * h->Wl * h->Cl, h->d_loc, h->d_log_scale,
                       (const double*)h->bufLd, logp_dev, (float*)nullptr);
    LAUNCHCHK("k_prior");
    return run_backward(h, x_dev, z, N, dx_dev, s);
  });
}

int glowk_sample(glowk_handle* h, const float* eps_dev, int N, float* x_dev, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (int rc = check_ready(h, N)) return rc;
  if (!eps_dev || !x_dev) return fail("null tensor");
  hipStream_t s = (hipStream_t)stream;
  const int E = h->Hl * h->Wl * h->Cl;
  const size_t total = (size_t)N * E;
  return guarded(h, s, [&]() -> int {
    hipLaunchKernelGGL(k_prior_sample, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, eps_dev, total, E, h->d_loc, h->d_log_scale,
                       h->bufZ);
    LAUNCHCHK("k_prior_sample");
    return run_inverse(h, h->bufZ, N, x_dev, s);
  });
}

int glowk_prior_log_prob(glowk_handle* h, const float* z_dev, int N, float* logp_dev, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (int rc = check_ready(h, N)) return rc;
  if (!z_dev || !logp_dev) return fail("null tensor");
  hipLaunchKernelGGL(k_prior, dim3(N), dim3(256), 0, (hipStream_t)stream, z_dev, h->Hl * h->Wl * h->Cl, h->d_loc, h->d_log_scale,
                     (const double*)nullptr, logp_dev, (float*)nullptr);
  LAUNCHCHK("k_prior");
  return 0;
}

int64_t glowk_fused_steps(const glowk_handle* h) { return h ? h->fused_steps : -1; }

int glowk_kernel_families(const glowk_handle* h, int64_t* out7) {
  if (!h || !out7) return fail("null argument");
  for (int i = 0; i < 7; ++i) out7[i] = h->family_launches[i];
  return 0;
}

int glowk_profile_begin(glowk_handle* h) {
  if (!h) return fail("null handle");
  h->profiling = true;
  h->ev_used = 0;
  h->ev_level.clear();
  return 0;
}

int glowk_profile_end(glowk_handle* h, glowk_profile* out) {
  if (!h || !out) return fail("null argument");
  std::memset(out, 0, sizeof(*out));
  h->profiling = false;
  DeviceGuard dg(h->device);
  HIPCHK(hipDeviceSynchronize());
  for (size_t i = 0; i < h->ev_level.size(); ++i) {
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev_pool[2 * i], h->ev_pool[2 * i + 1]));
    const int l = h->ev_level[i];
    if (l >= 0 && l < 4) { out->net_ms[l] += ms; out->net_launches[l] += 1; }
  }
  h->ev_used = 0;
  h->ev_level.clear();
  return 0;
}

int glowk_squeeze(const float* x_dev, int N, int H, int W, int C, float* y_dev, void* stream) {
  if (!x_dev || !y_dev) return fail("null tensor");
  if (N <= 0 || H <= 0 || W <= 0 || H % 2 || W % 2) return fail("squeeze needs even H and W");  // flow_tfp_bijectors.py:165-166
  if ((long long)N * H * W * C > (1LL << 28)) return fail("squeeze: more than 2^28 elements in one call; split the batch");
  DeviceGuard dg(ptr_device(x_dev));
  PreArgs p = {0, 1, 0, 0};
  const int c = 4 * C;
  CDISPATCH(c, hipLaunchKernelGGL((k_in<CC>), dim3(N), dim3(256), 0, (hipStream_t)stream, x_dev, H, W, p, 0, (const float*)nullptr,
                                  (const float*)nullptr, y_dev, (double*)nullptr, 0.0));
  LAUNCHCHK("k_in(squeeze)");
  return 0;
}

int glowk_unsqueeze(const float* y_dev, int N, int hh, int ww, int c4, float* x_dev, void* stream) {
  if (!x_dev || !y_dev) return fail("null tensor");
  if (N <= 0 || hh <= 0 || ww <= 0 || c4 % 4) return fail("unsqueeze needs a channel count divisible by 4");
  if ((long long)N * hh * ww * c4 > (1LL << 28)) return fail("unsqueeze: more than 2^28 elements in one call; split the batch");
  DeviceGuard dg(ptr_device(y_dev));
  PreArgs p = {0, 1, 0, 0};
  CDISPATCH(c4, hipLaunchKernelGGL((k_out<CC>), dim3(N), dim3(256), 0, (hipStream_t)stream, y_dev, hh, ww, p, 0, x_dev));
  LAUNCHCHK("k_out(unsqueeze)");
  return 0;
}

int glowk_preprocess_forward(glowk_handle* h, const float* x_dev, int N, float* y_dev, float* logdet_dev, void* stream) {
  if (!h || !x_dev || !y_dev) return fail("null argument");
  DeviceGuard dg(h->device);
  if (int rc = check_batch(h, N)) return rc;
  const int E = h->cfg.H * h->cfg.W * h->cfg.C;
  double ldc = -(double)E * std::log((double)h->cfg.maxval - (double)h->cfg.minval);
  if (h->cfg.use_logit) ldc += (double)E * std::log(1.0 - 2.0 * (double)h->cfg.alpha);
  hipLaunchKernelGGL(k_pre_only, dim3(N), dim3(256), 0, (hipStream_t)stream, x_dev, E, pre_args(h->cfg), 0, y_dev, logdet_dev, ldc);
  LAUNCHCHK("k_pre_only");
  return 0;
}

int glowk_preprocess_inverse(glowk_handle* h, const float* y_dev, int N, float* x_dev, void* stream) {
  if (!h || !x_dev || !y_dev) return fail("null argument");
  DeviceGuard dg(h->device);
  if (int rc = check_batch(h, N)) return rc;
  const int E = h->cfg.H * h->cfg.W * h->cfg.C;
  hipLaunchKernelGGL(k_pre_only, dim3(N), dim3(256), 0, (hipStream_t)stream, y_dev, E, pre_args(h->cfg), 1, x_dev, (float*)nullptr, 0.0);
  LAUNCHCHK("k_pre_only");
  return 0;
}

int glowk_step_forward(glowk_handle* h, int level, int step, const float* u_dev, int N, float* y_dev, float* logdet_dev, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (int rc = check_ready(h, N)) return rc;
  if (level < 0 || level >= h->cfg.L || step < 0 || step >= h->cfg.K) return fail("no such step");
  if (!u_dev || !y_dev) return fail("null tensor");
  hipStream_t s = (hipStream_t)stream;
  const Level& lv = h->levels[level];
  const StepDev& sd = lv.dev[step];
  const int Q = N * lv.h * lv.w;
  return guarded(h, s, [&]() -> int {
    CDISPATCH(lv.c, hipLaunchKernelGGL((k_affine<CC>), dim3((Q + 255) / 256), dim3(256), 0, s, u_dev, Q, sd.Afwd, sd.bfwd, h->bufA));
    LAUNCHCHK("k_affine");
    int np = 1;
    if (int rc = launch_net(h, level, lv.c, h->cfg.F, net_args(h, lv, sd, h->bufA, lv.c, lv.c / 2, N), s, fwd_mode(h), &np)) return rc;
    if (logdet_dev) {
      // logdet accumulator starts at the step's constant h*w*(sum log_scale + sum log_S)
      std::vector<double> init(N, h->ld_step[(size_t)level * h->cfg.K + step]);
      HIPCHK(hipMemcpyAsync(h->bufLd, init.data(), (size_t)N * 8, hipMemcpyHostToDevice, s));
      HIPCHK(hipStreamSynchronize(s));
    }
    CoupleArgs ca;
    ca.o_save = nullptr;
    ca.vin = h->bufA; ca.P = h->bufP; ca.np = np; ca.pstride = h->pstride; ca.b3 = sd.b3; ca.A = nullptr; ca.b = nullptr;
    ca.out = y_dev; ca.out_stride = lv.c; ca.out_off = 0;
    ca.logdet = logdet_dev ? h->bufLd : nullptr; ca.log_s_out = nullptr; ca.t_out = nullptr;
    ca.Q = Q; ca.h = lv.h; ca.w = lv.w; ca.inverse = 0; ca.flag = flagp(h);
    if (int rc = launch_couple(lv.c, ca, N, s)) return rc;
    if (logdet_dev) {
      hipLaunchKernelGGL(k_prior, dim3(N), dim3(256), 0, s, (const float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr,
                         (const double*)h->bufLd, (float*)nullptr, logdet_dev);
      LAUNCHCHK("k_prior(logdet)");
    }
    return 0;
  });
}

int glowk_step_inverse(glowk_handle* h, int level, int step, const float* y_dev, int N, float* u_dev, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (int rc = check_ready(h, N)) return rc;
  if (level < 0 || level >= h->cfg.L || step < 0 || step >= h->cfg.K) return fail("no such step");
  if (!u_dev || !y_dev) return fail("null tensor");
  hipStream_t s = (hipStream_t)stream;
  const Level& lv = h->levels[level];
  const StepDev& sd = lv.dev[step];
  return guarded(h, s, [&]() -> int {
    int np = 1;
    if (int rc = launch_net(h, level, lv.c, h->cfg.F, net_args(h, lv, sd, y_dev, lv.c, lv.c / 2, N), s, fwd_mode(h), &np)) return rc;
    CoupleArgs ca;
    ca.o_save = nullptr;
    ca.vin = y_dev; ca.P = h->bufP; ca.np = np; ca.pstride = h->pstride; ca.b3 = sd.b3; ca.A = sd.Ainv; ca.b = sd.binv;
    ca.out = u_dev; ca.out_stride = lv.c; ca.out_off = 0;
    ca.logdet = nullptr; ca.log_s_out = nullptr; ca.t_out = nullptr;
    ca.Q = N * lv.h * lv.w; ca.h = lv.h; ca.w = lv.w; ca.inverse = 1; ca.flag = flagp(h);
    return launch_couple(lv.c, ca, N, s);
  });
}

int glowk_coupling_net(glowk_handle* h, int level, int step, const float* xb_dev, int N, float* log_s_dev, float* t_dev, void* stream) {
  if (!h) return fail("null handle");
  DeviceGuard dg(h->device);
  if (int rc = check_ready(h, N)) return rc;
  if (level < 0 || level >= h->cfg.L || step < 0 || step >= h->cfg.K) return fail("no such step");
  if (!xb_dev || !log_s_dev || !t_dev) return fail("null tensor");
  hipStream_t s = (hipStream_t)stream;
  const Level& lv = h->levels[level];
  const StepDev& sd = lv.dev[step];
  return guarded(h, s, [&]() -> int {
    int np = 1;
    if (int rc = launch_net(h, level, lv.c, h->cfg.F, net_args(h, lv, sd, xb_dev, lv.c / 2, 0, N), s, fwd_mode(h), &np)) return rc;
    CoupleArgs ca;
    ca.o_save = nullptr;
    ca.vin = nullptr; ca.P = h->bufP; ca.np = np; ca.pstride = h->pstride; ca.b3 = sd.b3; ca.A = nullptr; ca.b = nullptr;
    ca.out = nullptr; ca.out_stride = 0; ca.out_off = 0;
    ca.logdet = nullptr; ca.log_s_out = log_s_dev; ca.t_out = t_dev;
    ca.Q = N * lv.h * lv.w; ca.h = lv.h; ca.w = lv.w; ca.inverse = 0; ca.flag = flagp(h);
    return launch_couple(lv.c, ca, N, s);
  });
}

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
    hipLaunchKernelGGL(k_prior, dim3(N), dim3(256), 0, s, (const float*)z, E, h->d_loc, h->d_log_scale, (const double*)h->bufLd, logp_dev, (float*)nullptr);
    LAUNCHCHK("k_prior");
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
  TrainCtx tc{grad_dev, scale, split};
  // (the input gradient falls out of the sweep as well; the trainer has no use for it: it lands in the block-level scratch,
  //  which is free again by the time the last kernel of the sweep writes it)
  if (int rc = run_backward(h, x_dev, z, N, h->bufZ, s, &tc)) return rc;
  // ActNorm / 1x1: the per-step sums come back once, the c x c chain rule runs on the host in fp64, the results go up in one copy
  const size_t steps = (size_t)cfg.L * cfg.K;
  std::vector<double> sums(steps * AFF_NOUT_MAX);
  HIPCHK(hipMemcpyAsync(sums.data(), h->trAffSum, sums.size() * 8, hipMemcpyDeviceToHost, s));
  if (split) HIPCHK(hipMemcpyAsync(h->h_flag, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(h->h_gmax, h->tr_gmax, sizeof(float) * 64, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (split && h->h_flag[0]) {       // the range guard of the split arithmetic fired somewhere in the sweep: its gradients are not usable
    h->h_flag[0] = 0;
    HIPCHK(hipMemsetAsync(h->d_flag, 0, sizeof(int), s));
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
  for (int lvl = 0; lvl < cfg.L; ++lvl) {
    const Level& lv = h->levels[lvl];
    const TrainOff t = train_off(lv.c, cfg.F);
    const size_t small = t.K1;            // [als | ash | L | logS | U] (padded) sit at the head of the step block
    std::vector<float> out((size_t)cfg.K * small, 0.0f);
    for (int k = 0; k < cfg.K; ++k)
      affine_chain_rule(lv, k, sums.data() + ((size_t)lvl * cfg.K + k) * AFF_NOUT_MAX, N, (double)scale, out.data() + (size_t)k * small, t);
    HIPCHK(hipMemcpy2DAsync(grad_dev + h->tr_level_off[lvl], t.total * 4, out.data(), small * 4, small * 4, cfg.K, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));      // (out is a local)
  }
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
  //      bit for bit the host-packed ones): each level runs on a stream of its own, the host joins them once for the c x c algebra ----
  if (h->tr_streams.empty()) {
    h->tr_streams.resize(cfg.L); h->tr_events.resize(cfg.L + 1);
    for (hipStream_t& t : h->tr_streams) HIPCHK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking));
    for (hipEvent_t& e : h->tr_events) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    size_t tot = 0;
    h->tr_pin_off.assign(cfg.L + 1, 0);
    for (int lvl = 0; lvl < cfg.L; ++lvl) {
      const StepLayout SL = step_layout(h->levels[lvl].c, F);
      const TrainOff t = train_off(h->levels[lvl].c, F);
      h->tr_pin_off[lvl] = tot;
      tot += (size_t)cfg.K * (8 + t.K1 + h->levels[lvl].c + (SL.total - SL.Afwd));
    }
    h->tr_pin_off[cfg.L] = tot;
    HIPCHK(hipHostMalloc((void**)&h->tr_pinned, tot * 4, hipHostMallocDefault));
  }
  HIPCHK(hipEventRecord(h->tr_events[0], s));
  // pinned staging of level lvl: scales | small tensors | conv3 biases | folded affine blocks
  auto pin_sc = [&](int lvl) { return h->tr_pinned + h->tr_pin_off[lvl]; };
  auto pin_sm = [&](int lvl) { return pin_sc(lvl) + (size_t)cfg.K * 8; };
  auto pin_b3 = [&](int lvl) { return pin_sm(lvl) + (size_t)cfg.K * train_off(h->levels[lvl].c, F).K1; };
  auto pin_blk = [&](int lvl) { return pin_b3(lvl) + (size_t)cfg.K * h->levels[lvl].c; };
  std::vector<char> refresh16(cfg.L, 0);
  for (int lvl = 0; lvl < cfg.L; ++lvl) {
    Level& lv = h->levels[lvl];
    hipStream_t ls = h->tr_streams[lvl];
    HIPCHK(hipStreamWaitEvent(ls, h->tr_events[0], 0));
    const StepLayout SL = step_layout(lv.c, F);
    const TrainOff t = train_off(lv.c, F);
    const float* p0 = h->tr_params + h->tr_level_off[lvl];
    float* img0 = h->arena + lv.dev[0].arena_off;
    hipLaunchKernelGGL(k_repack_f32, dim3((unsigned)((h->tr_map_n[lvl] + 255) / 256), cfg.K), dim3(256), 0, ls, (const int*)h->tr_map[lvl], h->tr_map_n[lvl],
                       p0, t.total, img0 + SL.K1p, SL.total);
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
      hipLaunchKernelGGL(k_repack_f16, dim3((unsigned)((h->tr_map16_n[lvl] + 1023) / 1024), cfg.K), dim3(256), 0, ls, (const int*)h->tr_map16[lvl],
                         h->tr_map16_n[lvl], fp, reinterpret_cast<unsigned short*>(img0 + SL.RHp), SL.total * 2);
      LAUNCHCHK("k_repack_f16");
      HIPCHK(hipMemcpyAsync(pin_sc(lvl), fp.scales, (size_t)cfg.K * 8 * 4, hipMemcpyDeviceToHost, ls));
    } else if (h->tr_map16[lvl]) {
      stale16 = true;
    }
    // small tensors: down to the host (they parameterise the fp64 fold of ActNorm + 1x1), folded, back up
    const size_t small = t.K1;
    HIPCHK(hipMemcpy2DAsync(pin_sm(lvl), small * 4, p0, t.total * 4, small * 4, cfg.K, hipMemcpyDeviceToHost, ls));
    HIPCHK(hipMemcpy2DAsync(pin_b3(lvl), (size_t)lv.c * 4, p0 + t.b3, t.total * 4, (size_t)lv.c * 4, cfg.K, hipMemcpyDeviceToHost, ls));
  }
  for (int lvl = 0; lvl < cfg.L; ++lvl) {
    Level& lv = h->levels[lvl];
    hipStream_t ls = h->tr_streams[lvl];
    HIPCHK(hipStreamSynchronize(ls));
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
      if (refresh16[lvl]) {      // the kernels' scale arguments and range-guard limits live in the host-side step descriptors
        StepDev& d = lv.dev[k];
        const float* q8 = pin_sc(lvl) + (size_t)k * 8;
        if (SL.slotH || SL.slotS) { d.sc1 = q8[0]; d.sc2 = q8[1]; d.sc3 = q8[2]; d.xlim_f = q8[6]; }
        if (SL.slotHB || SL.slotSB) { d.scb1 = q8[3]; d.scb2 = q8[4]; d.scb3 = q8[5]; d.xlim_b = q8[7]; }
      }
    }
    // (pinned source, rewritten only by the next call, which first waits for this level's stream above)
    HIPCHK(hipMemcpy2DAsync(img0 + SL.Afwd, SL.total * 4, blocks, tail * 4, tail * 4, cfg.K, hipMemcpyHostToDevice, ls));
    HIPCHK(hipEventRecord(h->tr_events[1 + lvl], ls));
    HIPCHK(hipStreamWaitEvent(s, h->tr_events[1 + lvl], 0));     // whatever the caller's stream runs next sees the refreshed images
  }
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

namespace {
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
}  // namespace

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
