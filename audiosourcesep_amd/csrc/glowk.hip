// glowk engine core: weight store, launch sequencing and the inference part of the C ABI of include/glowk.h.
// Build: __graft_entry__.build (hipcc --offload-arch=gfx950; one object per translation unit, linked into libglowk.so)
#include "glowk_engine.h"
#include "glowk_light.h"

namespace glowk_eng {
thread_local std::string g_err;
}
using namespace glowk_eng;

namespace glowk_eng {

// ---- launch helpers (policy templates: glowk_launch.h) --------------------------------------------
}  // namespace glowk_eng

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
                     on("GLOWK_COUPLE_4"), on("GLOWK_NO_FUSE"), on("GLOWK_WGRAD_PLAIN"), on("GLOWK_WGRAD_128"), on("GLOWK_CO_OFF"), on("GLOWK_Q_OFF"), on("GLOWK_CO_SPLIT_OFF"), on("GLOWK_CO_TRAIN_OFF"), on("GLOWK_WGRAD_16_OFF"), on("GLOWK_CO8_OFF"), on("GLOWK_CO_MID_OFF")};
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

namespace glowk_eng {
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

int launch_net_raw(int c, int F, const NetArgs& a, int mode, hipStream_t s, bool dry) {
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
int run_forward(glowk_handle* h, const float* x, int N, float* z_dst, hipStream_t s, bool save, bool keep_hidden) {
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

// d sum_n log_prob[n] / dx, after run_forward(save = true) on the same x; z is that run's latent.
// tc != null: the training sweep -- exact fp32 kernels, and every step also leaves its weight gradients in tc->grad.
int run_backward(glowk_handle* h, const float* x, const float* z, int N, float* dx, hipStream_t s, TrainCtx* tc) {
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
      // (the ActNorm / 1x1 sums first: after the last level's the host has all it needs for its share of the step, which then runs beside
      //  this level's GEMMs -- glowk_param_grad)
      if (int rc = train_affine_sums(h, lvl, 0, K, v0, v_bs, h->trGv, (ptrdiff_t)go_slot, N, s)) return rc;
      if (lvl == 0 && tc->sums_ready) { HIPCHK(hipEventRecord(tc->sums_ready, s)); tc->sums_ready = nullptr; }
      if (int rc = train_network_grads(h, tc, lvl, 0, K, v0, v_bs, h->trGo, (ptrdiff_t)go_slot, R10, r_bs, h->trM1, h->trM2, (ptrdiff_t)m_slot, N, s, bfac)) return rc;
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
  if (tc && tc->sums_ready) { HIPCHK(hipEventRecord(tc->sums_ready, s)); tc->sums_ready = nullptr; }     // (step-by-step mode: the sums end the sweep)
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

// log_prob (or the log-det alone) from the latent and the accumulated log-det: the prior kernel, for callers outside this unit
int launch_prior(glowk_handle* h, const float* z, int N, float* logp_dev, hipStream_t s) {
  hipLaunchKernelGGL(k_prior, dim3(N), dim3(256), 0, s, z, h->Hl * h->Wl * h->Cl, h->d_loc, h->d_log_scale, (const double*)h->bufLd, logp_dev, (float*)nullptr);
  LAUNCHCHK("k_prior");
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


}  // namespace glowk_eng

// =================================================================================================
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
  if (h->tr_ev_sums) hipEventDestroy(h->tr_ev_sums);
  if (h->tr_ev_up) hipEventDestroy(h->tr_ev_up);
  if (h->tr_side) hipStreamDestroy(h->tr_side);
  if (h->h_sums) hipHostFree(h->h_sums);
  if (h->h_up) hipHostFree(h->h_up);
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

namespace glowk_eng {
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
}  // namespace glowk_eng

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


}  // extern "C"
