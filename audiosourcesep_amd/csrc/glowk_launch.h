// Launch policy of the coupling-network kernels (which kernel family, how many passes, split or not), as templates over the
// level shape.  The heavy kernels are instantiated ONLY through launch_net_t<CI, NF>; each (CI, NF) pair is explicitly
// instantiated in its own translation unit (glowk_net_inst.hip, compiled once per pair), so the ~100 kernel instances build
// in parallel; glowk.hip sees extern template declarations.
#pragma once
#include "glowk_kernels.h"
#include "glowk_co.h"
#include "glowk_q.h"

#include <cstdlib>
#include <string>

namespace glowk_detail {

int num_cus();                          // compute units of the current device (queried once); glowk.hip
void note_co();                         // ... and that it took the co-resident form (after note_family)
void note_q();                          // ... or the small-grid form with all conv1 blocks first (glowk_q.h)
void note_family(int family);           // which kernel family a (non-dry) launch took: 0 k_net_f32, 1 k_net_h3 (32x32x16), 2 k_net_h3s
                                        // (16x16x32), 3 its half-wave form, 4 the fused network + coupling kernel; glowk.hip
bool h3_shape16();                      // GLOWK_H3_SHAPE=32 keeps the forward pass on the 32x32x16 kernel (A/B timing); glowk.hip
void launch_fail(const std::string&);   // sets glowk_last_error(); glowk.hip

// Diagnostic switches (A/B timing, parity tests of one launch form against another): environment variables, read ONCE -- when the
// library is loaded and again by glowk_reload_env() -- not per launch (round-3 verdict: six getenv() scans per flow step sat on the
// latency-bound path, ~1 200 per 30-tile gradient call).  glowk.hip owns the instance.
struct EnvSwitches {
  bool half_off;            // GLOWK_HALF_OFF: never the half-wave form where another one exists
  bool half_force;          // GLOWK_HALF_FORCE: the half-wave form of the plain forward network at every grid size
  bool fam16_small;         // GLOWK_FAM16_SMALL: the 16x16x32 family of the gradient path at every grid size
  bool bwd_light_4;         // GLOWK_BWD_LIGHT_4: k_bwd_light with four lanes per pixel whatever the grid
  bool couple_per_sample;   // GLOWK_COUPLE_PER_SAMPLE: k_couple (one workgroup per sample) instead of the flat grid
  bool couple_4;            // GLOWK_COUPLE_4: k_couple_flat with four lanes per pixel whatever the level
  bool no_fuse;             // GLOWK_NO_FUSE: network + coupling as two kernels at the 4-channel level
  bool wgrad_plain;         // GLOWK_WGRAD_PLAIN: the weight-gradient GEMM's plain (not fenced) round
  bool wgrad_128;           // GLOWK_WGRAD_128: 128 x 128 tiles in the exact weight-gradient GEMM
  bool co_off;              // GLOWK_CO_OFF: never the co-resident (two workgroups per CU) form of the forward network
  bool q_off;               // GLOWK_Q_OFF: never the all-conv1-first small-grid form (glowk_q.h)
  bool co_split_off;        // GLOWK_CO_SPLIT_OFF: never the one-pass-per-workgroup (small-grid) form of k_net_h3c
  bool co_train_off;        // GLOWK_CO_TRAIN_OFF: the training sweep stays on the 32x32x16 family (A/B timing)
  bool wgrad_16_off;        // GLOWK_WGRAD_16_OFF: no 16-wave / 256 x 256 form of the split weight-gradient GEMM (A/B timing)
  bool co8_off;             // GLOWK_CO8_OFF: no co-resident form at the 8-channel level (A/B timing)
  bool co_mid_off;          // GLOWK_CO_MID_OFF: the co-resident form only on grids of >= 4 workgroups per CU (or <= 1: SPLIT), as first built
};
const EnvSwitches& env();

// k_net_h3 / k_net_h3s launch forms.  NP = passes over the hidden width (2, or 4 where the shape needs the registers);
// when NP workgroups per 256 pixels still fit the CUs in one round the passes become workgroups of their own (SPLIT):
// 1/NP of the latency per launch.  Returns the number of partial P buffers the launch writes (= NP), 0 if no instance fits.
// dry: decide only (the consumers of P need the same answer).
template <int KIN, int MOUT, int NF, int MODE>
int launch_h3(const NetArgs& a, hipStream_t s, bool dry) {
  constexpr bool F2 = RingH<KIN, MOUT, NF, MODE, 2>::FITS, F4 = RingH<KIN, MOUT, NF, MODE, 4>::FITS;
  const int wgs = (a.Q + 255) / 256, cus = num_cus();
  if constexpr (F4) {
    if (a.max_np >= 4 && (4 * wgs <= cus || !F2)) {
      const bool split = 4 * wgs <= cus;
      if (!dry) {
        if (split) hipLaunchKernelGGL((k_net_h3<KIN, MOUT, NF, MODE, 4, true>), dim3(wgs, 4), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((k_net_h3<KIN, MOUT, NF, MODE, 4, false>), dim3(wgs), dim3(512), 0, s, a);
        note_family(1);
      }
      return 4;
    }
  }
  if constexpr (F2) {
    if (!dry) {
      if (2 * wgs <= cus) hipLaunchKernelGGL((k_net_h3<KIN, MOUT, NF, MODE, 2, true>), dim3(wgs, 2), dim3(512), 0, s, a);
      else hipLaunchKernelGGL((k_net_h3<KIN, MOUT, NF, MODE, 2, false>), dim3(wgs), dim3(512), 0, s, a);
      note_family(1);
    }
    return 2;
  }
  return 0;
}

// does the runtime place two workgroups of this co-resident instance on a CU?  (asked once per instance: one device type per process)
template <class Kernel>
inline bool co_two_per_cu(Kernel kernel) {
  static int blocks = -1;
  if (blocks < 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, 0) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    blocks = n;
  }
  return blocks >= 2;
}

template <int KIN, int MOUT, int NF, int MODE>
int launch_h3s(const NetArgs& a, hipStream_t s, bool dry) {
  constexpr bool F2 = RingS<KIN, MOUT, NF, MODE, 2>::FITS, F4 = RingS<KIN, MOUT, NF, MODE, 4>::FITS;
  const int wgs = (a.Q + 255) / 256, cus = num_cus();
  if constexpr (F4) {
    if (a.max_np >= 4 && (4 * wgs <= cus || !F2)) {
      const bool split = 4 * wgs <= cus;
      if (!dry) {
        if (split) hipLaunchKernelGGL((k_net_h3s<KIN, MOUT, NF, MODE, 4, true>), dim3(wgs, 4), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((k_net_h3s<KIN, MOUT, NF, MODE, 4, false>), dim3(wgs), dim3(512), 0, s, a);
        note_family(2);
      }
      return 4;
    }
  }
  if constexpr (F2) {
    const bool split = 2 * wgs <= cus;
    // the coupling fused into the kernel (glowk_kernels.h: fused_couple) where the caller asks for it (NetArgs::fuse: plain forward
    // direction, geometry checked by the host), the level has four channels and both passes run in one workgroup: answers 100 =
    // "no P was written, the step's output is in place (but for the rows k_couple_edge finishes)"
    // the co-resident form (glowk_co.h: four-wave / 128-pixel workgroups, two to a CU), where the caller allows it (NetArgs::co) and it
    // has an instance: on grids that fill the chip with it both passes in one workgroup -- fused (101: forward modes at the 4-channel
    // level) or writing P once (1) --, on small grids (2 x Q/128 workgroups fit two to a CU) one pass per workgroup (2 partial P buffers)
    if constexpr (RingC<KIN, MOUT, NF, MODE>::FITS) {
      const int wgc = (a.Q + CO_PX - 1) / CO_PX;
      const bool co_ok = a.co && !env().co_off && !(RingC<KIN, MOUT, NF, MODE>::NMT == 5 && env().co8_off);
      // (grids in between -- more 128-pixel workgroups than CUs, fewer than four per CU, e.g. BASIS' 30 mixture tiles at the reference's
      //  96 x 64: the eight-wave kernel would run one two-pass workgroup on 70 % of the CUs; this form runs its workgroups two to a CU in
      //  one round up to 2 x CUs, two rounds up to 4 x CUs.  GLOWK_CO_MID_OFF: the round-4 rule "four per CU or none", for A/B timing)
      if (co_ok && (wgc >= 4 * cus || (wgc > cus && !env().co_mid_off))) {
        if constexpr (RingC<KIN, MOUT, NF, MODE | 16>::FITS) {
          // (the form only pays with TWO workgroups per CU -- 2 x 78.8 KB of LDS, 2 x 4 x 248 VGPRs: ask the runtime once per instance, and
          //  keep the eight-wave kernel where a driver / device leaves room for one)
          if (a.fuse && co_two_per_cu(k_net_h3c<KIN, MOUT, NF, MODE | 16, false>)) {
            if (!dry) { hipLaunchKernelGGL((k_net_h3c<KIN, MOUT, NF, MODE | 16, false>), dim3(wgc), dim3(256), 0, s, a); note_family(4); note_co(); }
            return 101;
          }
        }
        if (!a.fuse && co_two_per_cu(k_net_h3c<KIN, MOUT, NF, MODE, false>)) {
          if (!dry) { hipLaunchKernelGGL((k_net_h3c<KIN, MOUT, NF, MODE, false>), dim3(wgc), dim3(256), 0, s, a); note_family(2); note_co(); }
          return RingC<KIN, MOUT, NF, MODE>::MERGE ? 1 : 2;
        }
      }
      if (co_ok && !env().co_split_off && split && wgc <= cus && a.max_np >= 2 && co_two_per_cu(k_net_h3c<KIN, MOUT, NF, MODE, true>)) {
        if (!dry) { hipLaunchKernelGGL((k_net_h3c<KIN, MOUT, NF, MODE, true>), dim3(wgc, 2), dim3(256), 0, s, a); note_family(2); note_co(); }
        return 2;
      }
    }
    if constexpr ((MODE == NET_FWD || MODE == NET_FWD2 || MODE == NET_FWD_SAVE) && MOUT == 36) {
      if constexpr (RingS<KIN, MOUT, NF, MODE | 16, 2>::FITS && RingS<KIN, MOUT, NF, MODE | 16, 2>::MERGE) {
        if (a.fuse && !split) {
          if (!dry) { hipLaunchKernelGGL((k_net_h3s<KIN, MOUT, NF, MODE | 16, 2, false>), dim3(wgs), dim3(512), 0, s, a); note_family(4); }
          return 100;
        }
      }
    }
    if (!dry) {
      if (split) hipLaunchKernelGGL((k_net_h3s<KIN, MOUT, NF, MODE, 2, true>), dim3(wgs, 2), dim3(512), 0, s, a);
      else hipLaunchKernelGGL((k_net_h3s<KIN, MOUT, NF, MODE, 2, false>), dim3(wgs), dim3(512), 0, s, a);
      note_family(2);
    }
    return (!split && RingS<KIN, MOUT, NF, MODE, 2>::MERGE) ? 1 : 2;     // (merged: the two passes' sums leave the kernel as one buffer)
  }
  return 0;
}

// The 16x16x32 family with ONE 16-pixel half per wave (MODE | 32: 128-pixel workgroups, always four passes).  Twice the
// workgroups at half the work per phase: chosen where the 256-pixel workgroups with their passes as workgroups of their own
// still leave half the CUs idle (latency-bound grids: the deeper levels at the reference's batch sizes of 30 / 32 tiles), and for
// shapes whose small-conv fragments only fit the registers at one half per wave (the 32-channel level's backward network: K = 288).
inline bool half_wave_grid(const NetArgs& a) {
  if (env().half_off) return false;                 // (A/B timing and diagnostics)
  return 8 * ((a.Q + 255) / 256) <= num_cus();
}

template <int KIN, int MOUT, int NF, int MODE>
int launch_h3s_half(const NetArgs& a, hipStream_t s, bool dry) {
  if constexpr (RingS<KIN, MOUT, NF, MODE | 32, 4>::FITS) {
    if (a.max_np < 4) return 0;
    const int wgs = (a.Q + 127) / 128;
    // passes as workgroups of their own, each alone on its CU: the form with all conv1 blocks first (glowk_q.h), where it has an instance
    if constexpr (RingQ<KIN, MOUT, NF, MODE>::FITS) {
      if (4 * wgs <= num_cus() && !env().q_off) {
        if (!dry) { hipLaunchKernelGGL((k_net_h3q<KIN, MOUT, NF, MODE>), dim3(wgs, 4), dim3(512), 0, s, a); note_family(3); note_q(); }
        return 4;
      }
    }
    if (!dry) {
      if (4 * wgs <= num_cus()) hipLaunchKernelGGL((k_net_h3s<KIN, MOUT, NF, MODE | 32, 4, true>), dim3(wgs, 4), dim3(512), 0, s, a);
      else hipLaunchKernelGGL((k_net_h3s<KIN, MOUT, NF, MODE | 32, 4, false>), dim3(wgs), dim3(512), 0, s, a);
      note_family(3);
    }
    return 4;
  }
  return 0;
}

// the saving forward pass and the backward pass of a level must agree on the form (their ReLU-mask layouts differ: one entry per
// pixel block of a wave): both have a half-wave instance
template <int CI, int NF>
constexpr bool half_ok() {
  return RingS<CI, 18 * CI, NF, NET_FWD_SAVE | 32, 4>::FITS && RingS<2 * CI, 9 * CI, NF, NET_BWD | 32, 4>::FITS;
}
// ... and for this level the half-wave form is the ONLY split form of the gradient path (no 256-pixel instance of the backward network)
template <int CI, int NF>
constexpr bool half_only() {
  return half_ok<CI, NF>() && !(RingS<2 * CI, 9 * CI, NF, NET_BWD, 2>::FITS || RingS<2 * CI, 9 * CI, NF, NET_BWD, 4>::FITS) &&
         !(RingH<2 * CI, 9 * CI, NF, NET_BWD, 2>::FITS || RingH<2 * CI, 9 * CI, NF, NET_BWD, 4>::FITS);
}
// the same for the training sweep (MODE | 8: the launches also store their hidden tensors)
template <int CI, int NF>
constexpr bool half_train_ok() {
  return RingS<CI, 18 * CI, NF, (NET_FWD_SAVE | 8) | 32, 4>::FITS && RingS<2 * CI, 9 * CI, NF, (NET_BWD | 8) | 32, 4>::FITS;
}
template <int CI, int NF>
constexpr bool half_train_only() {
  return half_train_ok<CI, NF>() && !((RingH<CI, 18 * CI, NF, (NET_FWD_SAVE | 8), 2>::FITS || RingH<CI, 18 * CI, NF, (NET_FWD_SAVE | 8), 4>::FITS) &&
                                      (RingH<2 * CI, 9 * CI, NF, (NET_BWD | 8), 2>::FITS || RingH<2 * CI, 9 * CI, NF, (NET_BWD | 8), 4>::FITS));
}
template <int CI, int NF>
inline bool use_half_train(const NetArgs& a) {
  if constexpr (!half_train_ok<CI, NF>()) return false;
  return a.fam16 && h3_shape16() && a.max_np >= 4 && (half_train_only<CI, NF>() || half_wave_grid(a));
}

template <int CI, int NF>
inline bool use_half(const NetArgs& a) {
  if constexpr (!half_ok<CI, NF>()) return false;
  return a.fam16 && h3_shape16() && a.max_np >= 4 && (half_only<CI, NF>() || half_wave_grid(a));
}

// a level's saving forward pass and its backward pass run in ONE kernel family (their ReLU-mask layouts differ): the
// 16x16x32 family if both have an instance that works whatever the batch size (NP = 4 if NP = 2 does not fit needs room
// for four partial buffers, which the save buffers may not have)
// ... and only where the grid fills the chip (measured: +3.3 % at 1024 tiles, -2.5 % at 30, where the launches are split
// into passes and latency-bound).  Both launches of a level see the same pixel count, so they decide alike.
// (re-measured in round 3 with GLOWK_FAM16_SMALL=1 -- the 16x16x32 family at every grid size: 8.72 vs 8.72 ms for the gradient of 30
//  tiles, within +-1 % at 8 ... 128 tiles: no reason to change the rule the fuzz runs validated)
inline bool big_grid(const NetArgs& a) { return 2 * ((a.Q + 255) / 256) > num_cus() || env().fam16_small; }

// ... or where both launches take the one-pass-per-workgroup co-resident form (k_net_h3c<..., SPLIT>: four-wave workgroups two to a CU
// instead of one eight-wave pass-workgroup per CU): the same question for the saving and the backward launch of a level, same answer
template <int CI, int NF>
inline bool co_split_grad(const NetArgs& a) {
  if constexpr (RingC<CI, 18 * CI, NF, NET_FWD_SAVE>::FITS && RingC<2 * CI, 9 * CI, NF, NET_BWD>::FITS) {
    const int wgc = (a.Q + CO_PX - 1) / CO_PX, cus = num_cus();
    return a.co && !env().co_off && !env().co_split_off && a.fam16 && h3_shape16() && a.max_np >= 2 && 2 * ((a.Q + 255) / 256) <= cus && wgc <= cus &&
           co_two_per_cu(k_net_h3c<CI, 18 * CI, NF, NET_FWD_SAVE, true>) && co_two_per_cu(k_net_h3c<2 * CI, 9 * CI, NF, NET_BWD, true>);
  }
  return false;
}

// The training sweep of a level in the co-resident form (k_net_h3c<..., MODE | 8>: the launches also store their hidden tensors): where
// both the saving forward and the backward network have an instance, every workgroup is full (the kernels count their stores: Q % 128
// == 0) and the grid is one of the two the form is built for -- >= 4 workgroups per CU with both passes in a workgroup, or <= 1 per CU
// with a workgroup per pass.  Same question, same answer for the two launches of a level (their ReLU-mask layouts must agree).
template <int CI, int NF>
inline bool co_train(const NetArgs& a) {
  if constexpr (RingC<CI, 18 * CI, NF, (NET_FWD_SAVE | 8)>::FITS && RingC<2 * CI, 9 * CI, NF, (NET_BWD | 8)>::FITS) {
    const int wgc = (a.Q + CO_PX - 1) / CO_PX, cus = num_cus();
    if (!(a.co && !env().co_off && !env().co_train_off && a.fam16 && h3_shape16() && a.Q % CO_PX == 0 && a.max_np >= 2)) return false;
    if (wgc >= 4 * cus || (wgc > cus && !env().co_mid_off))
      return co_two_per_cu(k_net_h3c<CI, 18 * CI, NF, (NET_FWD_SAVE | 8), false>) && co_two_per_cu(k_net_h3c<2 * CI, 9 * CI, NF, (NET_BWD | 8), false>);
    if (wgc <= cus && !env().co_split_off)
      return co_two_per_cu(k_net_h3c<CI, 18 * CI, NF, (NET_FWD_SAVE | 8), true>) && co_two_per_cu(k_net_h3c<2 * CI, 9 * CI, NF, (NET_BWD | 8), true>);
  }
  return false;
}
// (after co_train said yes; returns the number of partial P buffers)
template <int KIN, int MOUT, int NF, int MODE>
int launch_co_train(const NetArgs& a, hipStream_t s, bool dry) {
  if constexpr (RingC<KIN, MOUT, NF, MODE>::FITS) {
    const int wgc = (a.Q + CO_PX - 1) / CO_PX;
    const bool split = wgc <= num_cus();
    if (!dry) {
      if (split) hipLaunchKernelGGL((k_net_h3c<KIN, MOUT, NF, MODE, true>), dim3(wgc, 2), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_net_h3c<KIN, MOUT, NF, MODE, false>), dim3(wgc), dim3(256), 0, s, a);
      note_family(2); note_co();
    }
    return split ? 2 : 1;
  }
  return 0;
}

template <int CI, int NF>
constexpr bool fam16_ok() {
  return RingS<CI, 18 * CI, NF, NET_FWD_SAVE, 2>::FITS && (RingS<2 * CI, 9 * CI, NF, NET_BWD, 2>::FITS || RingS<2 * CI, 9 * CI, NF, NET_BWD, 4>::FITS);
}

// returns the number of partial P buffers written (>= 1), or -1 on error
template <int CI, int NF>
int launch_net_t(const NetArgs& a, int mode, hipStream_t s, bool dry) {
  const int ntiles = (a.Q + 127) / 128;
  int np = 0;
  switch (mode) {
    case NET_FWD:      if (!dry) hipLaunchKernelGGL((k_net_f32<CI, 18 * CI, NF, NET_FWD>), dim3(ntiles), dim3(256), 0, s, a); break;
    case NET_FWD_SAVE: if (!dry) hipLaunchKernelGGL((k_net_f32<CI, 18 * CI, NF, NET_FWD_SAVE>), dim3(ntiles), dim3(256), 0, s, a); break;
    case NET_BWD:      if (!dry) hipLaunchKernelGGL((k_net_f32<2 * CI, 9 * CI, NF, NET_BWD>), dim3(ntiles), dim3(256), 0, s, a); break;
    // training (exact fp32 only): the same two kernels, also storing their hidden tensors planar (NetArgs::st1 / st2)
    case 7:            if (!dry) hipLaunchKernelGGL((k_net_f32<CI, 18 * CI, NF, NET_FWD, true>), dim3(ntiles), dim3(256), 0, s, a); break;
    case 8:            if (!dry) hipLaunchKernelGGL((k_net_f32<2 * CI, 9 * CI, NF, NET_BWD, true>), dim3(ntiles), dim3(256), 0, s, a); break;
    // training in the split arithmetic: the 32x32x16 family's saving forward and backward launches, storing their hiddens (MODE | 8)
    case 10:   // (dry: returns 0 when the shape has no instance -- glowk_param_grad asks before it chooses the arithmetic of the sweep)
      if (use_half_train<CI, NF>(a)) np = launch_h3s_half<CI, 18 * CI, NF, (NET_FWD_SAVE | 8)>(a, s, dry);
      if (!np && co_train<CI, NF>(a)) np = launch_co_train<CI, 18 * CI, NF, (NET_FWD_SAVE | 8)>(a, s, dry);
      if (!np && a.RHp) np = launch_h3<CI, 18 * CI, NF, (NET_FWD_SAVE | 8)>(a, s, dry);
      if (!np) { if (dry) return 0; launch_fail("no split-arithmetic training instance for this shape"); return -1; }
      break;
    case 11:
      if (use_half_train<CI, NF>(a)) np = launch_h3s_half<2 * CI, 9 * CI, NF, (NET_BWD | 8)>(a, s, dry);
      if (!np && co_train<CI, NF>(a)) np = launch_co_train<2 * CI, 9 * CI, NF, (NET_BWD | 8)>(a, s, dry);
      if (!np && a.RHp) np = launch_h3<2 * CI, 9 * CI, NF, (NET_BWD | 8)>(a, s, dry);
      if (!np) { if (dry) return 0; launch_fail("no split-arithmetic training instance for this shape"); return -1; }
      break;
    case 9:            if (!dry) hipLaunchKernelGGL((k_net_f32<CI, 18 * CI, NF, NET_FWD_SAVE, true>), dim3(ntiles), dim3(256), 0, s, a); break;   // saving forward pass that keeps its hiddens
    case 3:   // f16x3 arithmetic: forward / forward with saves / backward; shapes without an instance run the exact fp32 kernel
      // (GLOWK_HALF_FORCE=1: the half-wave form at every grid size -- two 128-pixel workgroups per CU; an experiment)
      if (a.RSp && h3_shape16() && ((half_wave_grid(a) && !a.fuse) || env().half_force)) np = launch_h3s_half<CI, 18 * CI, NF, NET_FWD>(a, s, dry);
      if (!np && a.RSp && h3_shape16()) np = launch_h3s<CI, 18 * CI, NF, NET_FWD>(a, s, dry);
      if (!np && a.RHp) np = launch_h3<CI, 18 * CI, NF, NET_FWD>(a, s, dry);
      if (!np && !dry) hipLaunchKernelGGL((k_net_f32<CI, 18 * CI, NF, NET_FWD>), dim3(ntiles), dim3(256), 0, s, a);
      break;
    case 6:   // two-term forward (GLOWK_PREC_F16X2): 16x16x32 kernel only; without an instance, the three-term forms of case 3
      if (a.RSp && h3_shape16()) np = launch_h3s<CI, 18 * CI, NF, NET_FWD2>(a, s, dry);
      if (!np && a.RSp && h3_shape16()) np = launch_h3s<CI, 18 * CI, NF, NET_FWD>(a, s, dry);
      if (!np && a.RHp) np = launch_h3<CI, 18 * CI, NF, NET_FWD>(a, s, dry);
      if (!np && !dry) hipLaunchKernelGGL((k_net_f32<CI, 18 * CI, NF, NET_FWD>), dim3(ntiles), dim3(256), 0, s, a);
      break;
    case 4:
      if (use_half<CI, NF>(a)) np = launch_h3s_half<CI, 18 * CI, NF, NET_FWD_SAVE>(a, s, dry);
      if constexpr (fam16_ok<CI, NF>()) {
        if (!np && a.fam16 && h3_shape16() && (big_grid(a) || co_split_grad<CI, NF>(a))) np = launch_h3s<CI, 18 * CI, NF, NET_FWD_SAVE>(a, s, dry);
      }
      if (!np && a.RHp) np = launch_h3<CI, 18 * CI, NF, NET_FWD_SAVE>(a, s, dry);
      if (!np && !dry) hipLaunchKernelGGL((k_net_f32<CI, 18 * CI, NF, NET_FWD_SAVE>), dim3(ntiles), dim3(256), 0, s, a);
      break;
    case 5:
      if (use_half<CI, NF>(a)) np = launch_h3s_half<2 * CI, 9 * CI, NF, NET_BWD>(a, s, dry);
      if constexpr (fam16_ok<CI, NF>()) {
        if (!np && a.fam16 && h3_shape16() && (big_grid(a) || co_split_grad<CI, NF>(a))) np = launch_h3s<2 * CI, 9 * CI, NF, NET_BWD>(a, s, dry);
      }
      if (!np && a.RHp) np = launch_h3<2 * CI, 9 * CI, NF, NET_BWD>(a, s, dry);
      if (!np && !dry) hipLaunchKernelGGL((k_net_f32<2 * CI, 9 * CI, NF, NET_BWD>), dim3(ntiles), dim3(256), 0, s, a);
      break;
    default: launch_fail("bad k_net mode"); return -1;
  }
  if (!dry) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { launch_fail(std::string("k_net: ") + hipGetErrorString(e)); return -1; }
  }
  return np ? np : 1;
}

}  // namespace glowk_detail
