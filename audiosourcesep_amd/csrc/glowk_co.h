// k_net_h3c: the CO-RESIDENT form of the forward coupling network on v_mfma_f32_16x16x32_f16 (round-3 verdict, item 2).
//
// k_net_h3s runs ONE 8-wave / 256-pixel workgroup per CU (255 VGPRs, 105-150 KB of LDS): its gather prologue, the phase barriers of
// all eight waves and the fused coupling tail overlap nothing.  Here a workgroup is FOUR waves / 128 pixels -- one wave per SIMD,
// each wave keeping the two 16-pixel halves of k_net_h3s (same A-fragment reuse, same accumulators, same arithmetic order: results
// are bit for bit those of k_net_h3s) -- and TWO such workgroups share a CU (2 x <= 80 KB of LDS, 2 x 4 waves x 256 VGPRs), each with
// its own barrier domain and its own weight ring, started at different times by the dispatcher: while one gathers, waits at a barrier
// or runs its coupling tail, the other's waves own the matrix pipe.
//
// Ring of 16-KiB UNITS (half a 32-KiB chunk of the NP = 2 image: the row blocks of one half of the pass's hidden half -- the image
// itself is the one k_net_h3s reads).  Per hidden block i of a pass the ops are
//     X_i    conv1 of block i + ReLU + split                      (reads K[i & 1])
//     Ya_i   conv2 contribution of block i to row blocks 0 .. NFH-1        (reads M0 = unit 2 i)
//     Yb_i   ... to row blocks NFH .. 2 NFH - 1                            (reads M1 = unit 2 i + 1)
// then Z_0 .. Z_{NMT-1} (conv3, one unit of NFH tiles each: D, M0, M1).  All four waves run the same op; every op ends with a
// workgroup barrier.  A unit's DMA is issued in the op that follows the barrier after its slot's last reader and has at least
// one X or two Y halves to land:
//     X_i  issues unit 2 i + 1 -> M1      Ya_i issues conv1 operands of block i + 2 -> K[i & 1]      Yb_i issues unit 2 i + 2 -> M0
//     Z_0  issues conv3 unit 2 -> M1      Z_1  issues the next pass's conv3 unit 0 -> D              Z_2  issues the next pass's unit 0 -> M0
// (conv3 unit 0 sits in a slot of its own, D, because nothing separates Yb_{NF-1} from Z_0; conv3 unit 1 -> M0 is issued by Yb_{NF-1}
// as "unit 2 NF").  The waits are counted: vmcnt counts in issue order, so "everything but the n youngest DMA instructions of this
// wave has landed" is exactly what each op boundary needs (4 pieces per wave and unit, 1 per conv1 block); the barrier publishes it.
// The slots are distinct static __shared__ objects with static roles (glowk_kernels.h, rule 1): M0 always holds the first halves.
//
// Modes (MODE & 7): the plain forward network (NET_FWD, NET_FWD2), the forward network of the gradient path (NET_FWD_SAVE: ReLU masks
// stored from X and Z) and the backward network (NET_BWD, 4-channel level: 18 output rows = two conv3 units; the forward pass's masks
// of the workgroup's four 32-pixel blocks in LDS; every pixel's gradient normalised by a power of two) -- the mask layout is that of the
// eight-wave 16x16x32 kernels (one entry per 32-pixel block of a wave), so either form can follow the other.  MODE | 16: the coupling
// fused in (forward modes).  SPLIT: grid.y = 2 and every workgroup runs ONE pass (small grids: 2 x Q/128 workgroups of four waves,
// two to a CU; each pass leaves its own partial P).  MODE | 8 (NET_FWD_SAVE, NET_BWD): the training sweep's launches, which also store
// their hidden tensors planar (NetArgs::st1 / st2) -- lane n of a wave then holds the ADJACENT pixels 2 n, 2 n + 1 (not n, 16 + n), so
// that the values leave as 8-byte pairs, one full 128-byte line per hidden row and instruction; the stores are counted into the op-end
// waits (they stay in flight across a barrier), which is why the host takes the form only where every wave is full (Q % 128 == 0).
#pragma once
#include "glowk_kernels.h"

constexpr int CO_PX = 128;          // pixels per workgroup
constexpr int CO_PSTR = 132;        // fused coupling: floats per LDS row of P (4 * 132 = 16 mod 64 banks: the four row quarters of a tile apart)

template <int KIN, int MOUT, int NF, int MODE>
struct RingC {
  static constexpr int MODE7 = MODE & 7;
  using S = RingS<KIN, MOUT, NF, MODE7, 2>;            // per-wave tiling, weight image, epilogue constants: those of the two-pass form
  static constexpr bool FUSE = (MODE & 16) != 0;
  static constexpr bool SAVE = MODE7 == NET_FWD_SAVE, BWD = MODE7 == NET_BWD;
  static constexpr bool MERGE = S::NMT <= 3;           // two passes in one workgroup: pass 0's sums of P stay in registers, P is written once
                                                       // (five row blocks, the 8-channel level: 40 registers the kernel does not have -- two partial buffers)
  static constexpr bool M0LAST = S::NMT % 3 == 2;      // the pass's last conv3 unit sits in M0 (units cycle D, M0, M1): the next pass asks for its unit 0 itself
  static constexpr bool STORE = (MODE & 8) != 0;       // training: the launch also leaves its hidden tensors planar (NetArgs::st1 / st2)
  static constexpr int MODEX = MODE & 15;              // what h3s_X sees: the mode and the store bit
  static constexpr int NFH = S::NFH, NRB = S::NRB, NMT = S::NMT, KS = S::KS;
  static constexpr int UNITP = S::MAINP / 2;            // 1-KiB pieces per unit
  static constexpr int UNIT4 = S::MAIN4 / 2;            // float4 per unit
  static constexpr int PPW = UNITP / 4;                 // DMA instructions per wave and unit
  static constexpr int NG = NFH / 2;                    // groups of 12 MFMAs (two row blocks x two pixel halves x three terms) per Y half
  static constexpr int K1PW = (S::K1P + 3) / 4;         // DMA instructions per wave and conv1 block
  static constexpr int EPN = BWD ? 0 : S::EPN;          // (the backward network has no epilogue constants)
  static constexpr int MASKN = BWD ? 2 * 4 * NF * 64 : 0;   // backward: LDS copy of the masks [mask1 | mask2][wave][hidden block][lane], entries
  static constexpr size_t LDS_BYTES = (size_t)3 * UNIT4 * 16 + (size_t)2 * S::K14 * 16 + (size_t)EPN * 4 + (size_t)MASKN * 2 +
                                      (FUSE ? (size_t)36 * CO_PSTR * 4 + CO_PX * 16 : 0);
  static constexpr bool FITS = (MODE7 == NET_FWD || MODE7 == NET_FWD2 || SAVE || BWD) && !(MODE & 32) && (!STORE || ((SAVE || BWD) && !FUSE)) && S::NGRP == 1 && (NMT == 2 || NMT == 3 || (NMT == 5 && !STORE && !BWD)) &&
                               (!FUSE || (MOUT == 36 && !BWD)) && KS <= 3 && NF % 4 == 0 && NFH >= 2 && NFH % 2 == 0 && UNITP % 4 == 0 && PPW == NG &&
                               2 * LDS_BYTES <= 160 * 1024 + 1;   // (diagnostic paddings aside)
  // training: vector-memory stores an op issues per wave (8 pairs of values per activated hidden block from X and from Z; a saving launch adds
  // the block's ReLU mask).  The host only takes the form where every wave is full (Q % CO_PX == 0), so the counts are static.
  // (lane n of a wave holds the adjacent pixels 2 n, 2 n + 1 there, and the values leave as 8-byte pairs: full 128-byte lines)
  static constexpr int XST = STORE ? 8 : 0;
  static constexpr int z_acts(int z) { int n = 0; for (int t = z * NFH; t < (z + 1) * NFH; ++t) n += (t % NMT == 0) ? 1 : 0; return n; }
  static constexpr int z_st(int z) { return STORE ? z_acts(z) * (8 + (SAVE ? 1 : 0)) : 0; }
  __device__ static const float4* main_unit(const float4* img, int pass, int i, int half) { return S::main_chunk(img, pass, i) + (size_t)half * UNIT4; }
  __device__ static const float4* out_unit(const float4* img, int pass, int z) { return S::out_chunk(img, pass, 0) + (size_t)z * UNIT4; }
};

// end of an op without a unit to wait for: this wave's LDS reads have retired (a DMA into the slot they read may follow the barrier)
__device__ __forceinline__ void co_bar() {
#ifndef GLOWK_EXP_CONOLGKM
  __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0) alone
#endif
  h3_barrier();
}

// end of an op: all but the n youngest vector-memory operations of this wave are done, its LDS reads have retired, then the workgroup
// barrier.  (lgkmcnt(0): the first build waited on vmcnt alone and lost ~0.2 % of the waves' results under a co-resident partner -- an
// LDS read still in flight when the wave passed the barrier after which another wave's DMA refills the slot)
template <int N>
__device__ __forceinline__ void co_end() {
#ifdef GLOWK_EXP_CONOLGKM
  __builtin_amdgcn_s_waitcnt((N & 15) | 0x0F70 | ((N >> 4) << 14));   // vmcnt(N); expcnt / lgkmcnt: no wait
#else
  __builtin_amdgcn_s_waitcnt((N & 15) | 0x0070 | ((N >> 4) << 14));   // vmcnt(N) and lgkmcnt(0)
#endif
  h3_barrier();
#ifdef GLOWK_EXP_COSLEEP   // (diagnostic build: a pause between the publishing barrier and the first read of the published unit)
  __builtin_amdgcn_s_sleep(GLOWK_EXP_COSLEEP);
#endif
}
// ... with `more` extra operations (the op's stores: a saving launch's ReLU mask) allowed to stay in flight, decided per wave at run time
template <int N>
__device__ __forceinline__ void co_end_st(bool more) {
  if (more) co_end<N + 1>();
  else co_end<N>();
}

struct CoCtx {
  float4 *m0, *m1, *d;      // unit slots (distinct static __shared__ arrays)
  const float4* img;
  int w4;
  unsigned voff;
};

// one Y half: conv2 contribution of the current hidden block to row blocks H NFH .. H NFH + NFH - 1 (unit in `slot`), the A fragments of
// the next group read while the current one computes (as h3s_Y), and this wave's share of one DMA -- a unit (one piece per group) or
// the conv1 operands of a later block (first group) -- issued between the groups
// (H = 0: the conv1 operands, src = block image, dst = its K slot; H = 1: a unit, dst = M0)
template <int NFH, int MODE7, int H, int TAG, int K1P>
__device__ __forceinline__ void co_Y(const float4* slot, const h8 (&bh)[2], const h8 (&bl)[2], f32x4 (&acc2)[2 * NFH][2], int lane,
                                     const float4* src, float4* dst, int w4, unsigned voff) {
  const h8* buf = reinterpret_cast<const h8*>(slot) + lane;
  const char* ub = uniform_ptr(src);
  constexpr int NG = NFH / 2;
  h8 A[2][4];
  auto load = [&](h8 (&d)[4], int gi) {
    d[0] = buf[((2 * gi) * 2 + 0) * 64];            // row block 2gi hi, lo; row block 2gi+1 hi, lo
    d[1] = buf[((2 * gi) * 2 + 1) * 64];
    d[2] = buf[((2 * gi + 1) * 2 + 0) * 64];
    d[3] = buf[((2 * gi + 1) * 2 + 1) * 64];
  };
#ifdef GLOWK_EXP_COPRIO    // (A/B build: a wave in its matrix ops outranks its SIMD partner -- the other workgroup's wave, possibly in the VALU-heavy X op:
                           //  62.42 -> 62.74 ms per 1024-tile log_prob, +0.5 %: not kept)
  __builtin_amdgcn_s_setprio(GLOWK_EXP_COPRIO);
#endif
  load(A[0], 0);
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int o0 = H * NFH + 2 * gi, o1 = o0 + 1;
    if (gi + 1 < NG) load(A[(gi + 1) & 1], gi + 1);
    if constexpr (H == 1) {
      const int piece = gi * 4 + w4;
#ifdef GLOWK_EXP_COHALFDMA   // (diagnostic build, wrong results: every other unit piece is not fetched -- is the L2 -> LDS path what bounds the form?)
      if (gi & 1)
#endif
      glds16(reinterpret_cast<const float4*>(ub + (size_t)piece * 1024 + voff), dst + piece * 64);
      asm volatile("; dma site %0" ::"n"(TAG * 16 + gi));
    } else if (gi == 0) {
      stage4<K1P, TAG * 16 + 15>(src, dst, w4, voff);
    }
    __builtin_amdgcn_sched_barrier(0);
    const h8 (&af)[4] = A[gi & 1];
    acc2[o0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], bh[0], acc2[o0][0], 0, 0, 0);
    acc2[o0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], bh[1], acc2[o0][1], 0, 0, 0);
    acc2[o1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[3], bh[0], acc2[o1][0], 0, 0, 0);
    acc2[o1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[3], bh[1], acc2[o1][1], 0, 0, 0);
    if (MODE7 != NET_FWD2) {
      acc2[o0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bl[0], acc2[o0][0], 0, 0, 0);
      acc2[o0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bl[1], acc2[o0][1], 0, 0, 0);
      acc2[o1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2], bl[0], acc2[o1][0], 0, 0, 0);
      acc2[o1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2], bl[1], acc2[o1][1], 0, 0, 0);
    }
    acc2[o0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bh[0], acc2[o0][0], 0, 0, 0);
    acc2[o0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bh[1], acc2[o0][1], 0, 0, 0);
    acc2[o1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2], bh[0], acc2[o1][0], 0, 0, 0);
    acc2[o1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2], bh[1], acc2[o1][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#ifdef GLOWK_EXP_COPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// Z: conv3 unit z of a pass = NFH tiles (16 rows x one hidden block) in (hidden block, row block) order.  Two passes in one workgroup:
// pass 0 keeps its sums (bias included) in `keep`, the last pass adds them and writes the rows -- to HBM (P, one buffer) or, fused, to
// the workgroup's LDS copy; SOLO (one pass per workgroup): every pass writes its own partial P.
// DMA duties with a next pass in the workgroup (NEXT): three units (NMT = 3: D, M0, M1): Z_0 -> conv3 unit 2 into M1 (always), Z_1 -> the
// next pass's conv3 unit 0 into D, Z_2 -> its unit 0 into M0 (its X_0 then asks for unit 1 -> M1); two units (NMT = 2: D, M0): Z_0 -> the
// next pass's unit 1 into M1, Z_1 -> its conv3 unit 0 into D (its X_0 asks for unit 0 -> M0 and waits for it).
template <int KIN, int MOUT, int NF, int MODE, int PASS, bool SOLO, int Z>
__device__ __forceinline__ void co_Z(const NetArgs& a, const H3Ctx& hc, const CoCtx& c, const float* epl, float* pl,
                                     f32x4 (&acc2)[(RingC<KIN, MOUT, NF, MODE>::NRB)][2], f32x4 (&acc3)[(RingC<KIN, MOUT, NF, MODE>::NMT)][2], h8 (&bh)[2],
                                     h8 (&bl)[2], const int (&q)[2], const bool (&qok)[2], int lane, int kq,
                                     f32x4 (&keep)[(RingC<KIN, MOUT, NF, MODE>::NMT)][2]) {
  using G = RingC<KIN, MOUT, NF, MODE>;
  constexpr int NFH = G::NFH, NMT = G::NMT, M3 = MOUT, MODE7 = G::MODE7;
  constexpr bool NEXT = !SOLO && PASS == 0;            // another pass follows in this workgroup
  constexpr bool LAST = !SOLO && PASS == 1;            // ... this is the second: it completes pass 0's sums
  if constexpr (NMT == 3) {
    if constexpr (Z == 0) stage4<G::UNITP, 70>(G::out_unit(c.img, PASS, 2), c.m1, c.w4, c.voff);
    if constexpr (Z == 1 && NEXT) stage4<G::UNITP, 71>(G::out_unit(c.img, PASS + 1, 0), c.d, c.w4, c.voff);
    if constexpr (Z == 2 && NEXT) stage4<G::UNITP, 72>(G::main_unit(c.img, PASS + 1, 0, 0), c.m0, c.w4, c.voff);
  } else if constexpr (NMT == 2) {
    if constexpr (Z == 0 && NEXT) stage4<G::UNITP, 75>(G::main_unit(c.img, PASS + 1, 0, 1), c.m1, c.w4, c.voff);
    if constexpr (Z == 1 && NEXT) stage4<G::UNITP, 76>(G::out_unit(c.img, PASS + 1, 0), c.d, c.w4, c.voff);
  } else {      // five units (the 8-channel level): Z_z asks for conv3 unit z + 2 into the slot Z_{z-1} read; then the next pass's unit 1 -> M1, conv3 unit 0 -> D
    static_assert(NMT == 5, "conv3 units");
    if constexpr (Z == 0) stage4<G::UNITP, 80>(G::out_unit(c.img, PASS, 2), c.m1, c.w4, c.voff);
    if constexpr (Z == 1) stage4<G::UNITP, 81>(G::out_unit(c.img, PASS, 3), c.d, c.w4, c.voff);
    if constexpr (Z == 2) stage4<G::UNITP, 82>(G::out_unit(c.img, PASS, 4), c.m0, c.w4, c.voff);
    if constexpr (Z == 3 && NEXT) stage4<G::UNITP, 83>(G::main_unit(c.img, PASS + 1, 0, 1), c.m1, c.w4, c.voff);
    if constexpr (Z == 4 && NEXT) stage4<G::UNITP, 84>(G::out_unit(c.img, PASS + 1, 0), c.d, c.w4, c.voff);
  }
  const float4* slot = Z % 3 == 0 ? c.d : Z % 3 == 1 ? c.m0 : c.m1;
  const h8* buf = reinterpret_cast<const h8*>(slot) + lane;
  const float* pb = epl + NF * 32;
#pragma unroll
  for (int tp = 0; tp < NFH; ++tp) {
    const int t = Z * NFH + tp;
    const int fo = t / NMT, ml = t % NMT;
    if (ml == 0) {
      unsigned mask = 0, bits = 0;
      if (G::BWD) mask = hc.mkl[((size_t)(threadIdx.x >> 6) * NF + PASS * NFH + fo) * 64 + lane];        // mask1: the ReLU after conv1
      if constexpr (G::STORE)
#ifdef GLOWK_EXP_TILEDST
        bits = h3s_act_pair<MODE7, true>(acc2[2 * fo][0], acc2[2 * fo + 1][0], acc2[2 * fo][1], acc2[2 * fo + 1][1], a.sc2, mask, bh, bl,
                                         uniform_fptr(a.st2 + ((size_t)blockIdx.x * NF * 32 + (size_t)(PASS * NFH + fo) * 32 * GLOWK_EXP_TILEDST) * 128),
                                         ((unsigned)(4 * kq) * 128u + (unsigned)(q[0] & 127)) * 4u, 128u * 4u);
#else
        bits = h3s_act_pair<MODE7, true>(acc2[2 * fo][0], acc2[2 * fo + 1][0], acc2[2 * fo][1], acc2[2 * fo + 1][1], a.sc2, mask, bh, bl,
                                         uniform_fptr(a.st2 + (size_t)(PASS * NFH + fo) * 32 * a.Q), ((unsigned)(4 * kq) * (unsigned)a.Q + (unsigned)q[0]) * 4u,
                                         (unsigned)a.Q * 4u);
#endif
      else {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) bits |= h3s_act<MODE7, false>(acc2[2 * fo][hf], acc2[2 * fo + 1][hf], a.sc2, mask >> (8 * hf), bh[hf], bl[hf]) << (8 * hf);
      }
      if (G::SAVE && hc.wok) a.mask2[(hc.wblk * NF + PASS * NFH + fo) * 64 + lane] = (unsigned short)bits;
    }
    if (fo == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc3[ml][0][r] = 0.0f; acc3[ml][1][r] = 0.0f; }
    }
    const h8 ah = buf[(tp * 2 + 0) * 64], al = buf[(tp * 2 + 1) * 64];
    acc3[ml][0] = mfma3s<MODE7 == NET_FWD2>(ah, al, bh[0], bl[0], acc3[ml][0]);
    acc3[ml][1] = mfma3s<MODE7 == NET_FWD2>(ah, al, bh[1], bl[1], acc3[ml][1]);
    if (fo == NFH - 1) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = ml * 16 + 4 * kq + r;
          const float part = G::BWD ? acc3[ml][hf][r] * (a.sc3 * hc.ub[hf]) : PASS == 0 ? fmaf(acc3[ml][hf][r], a.sc3, pb[m]) : acc3[ml][hf][r] * a.sc3;
          if constexpr (NEXT && G::MERGE) { keep[ml][hf][r] = part; continue; }
          const float val = (LAST && G::MERGE) ? part + keep[ml][hf][r] : part;
          if constexpr (G::FUSE) {
            if (m < M3) pl[m * CO_PSTR + (int)(threadIdx.x >> 6) * 32 + 16 * hf + (lane & 15)] = val;
#ifdef GLOWK_EXP_COCHECK
            keep[ml][hf][r] = val;      // (diagnostic build: the kernel re-reads what it wrote)
#endif
          } else {
            float* Pp = a.P + ((SOLO || !G::MERGE) ? (size_t)PASS * a.pstride : (size_t)0);
            if (m < M3 && qok[hf]) Pp[(size_t)m * a.Q + q[hf]] = val;
          }
        }
    }
  }
  // next op and what it reads: Z_1 <- conv3 unit 1 (issued by Yb_{NF-1}); Z_2 <- conv3 unit 2 (issued by Z_0); X_0 of the next pass <- conv1
  // operands that landed long ago.  A saving launch's mask stores ride in between: it waits for everything (three short ops per pass).
  // A training launch's hidden stores stay in flight (vmcnt counts in issue order: what the next op reads is older than the count allowed
  // to remain -- Z_2's unit was asked for by Z_0, BEFORE Z_0's stores; the counter holds 63).
  constexpr bool ISSUED = NMT == 3 ? (Z == 0 || NEXT) : NMT == 2 ? (Z <= 1 && NEXT) : (Z <= 2 || NEXT);     // this op issued a unit
  if constexpr (Z + 1 < NMT) {
    if constexpr (G::STORE) {
      constexpr int after = (ISSUED ? G::PPW : 0) + G::z_st(Z) + ((NMT == 3 && Z == 1) ? G::z_st(0) : 0);
      co_end<(after < 63 ? after : 63)>();
    } else if constexpr (G::SAVE) co_end<0>();
    else co_end<(ISSUED ? G::PPW : 0)>();
  } else co_bar();
}

template <int KIN, int MOUT, int NF, int MODE, int PASS, bool SOLO>
__device__ __forceinline__ void co_pass(const NetArgs& a, const H3Ctx& hc, const CoCtx& c, const float* epl, float* pl,
                                        const h8 (&xh)[(RingC<KIN, MOUT, NF, MODE>::KS)][2], const h8 (&xl)[(RingC<KIN, MOUT, NF, MODE>::KS)][2],
                                        const int (&q)[2], const bool (&qok)[2], int lane, int kq, f32x4 (&keep)[(RingC<KIN, MOUT, NF, MODE>::NMT)][2]) {
  using G = RingC<KIN, MOUT, NF, MODE>;
  using S = typename G::S;
  constexpr int NRB = G::NRB, NFH = G::NFH, MODE7 = G::MODE7;
  constexpr int f2base = PASS * NFH * 32;
  constexpr bool SECOND = !SOLO && PASS == 1;          // units 0 / 1 were not loaded by the prologue
  f32x4 acc2[NRB][2];
#pragma unroll
  for (int ob = 0; ob < NRB; ++ob)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float b = G::BWD ? 0.0f : epl[f2base + ob * 16 + 4 * kq + r];   // conv2 bias (scaled)
      acc2[ob][0][r] = b;
      acc2[ob][1][r] = b;
    }
  const bool st = G::SAVE && PASS == 0 && hc.wok;      // this wave stores a ReLU mask from every X of this pass (h3s_X)
  constexpr int XST = PASS == 0 ? G::XST : 0;          // ... and a training launch the block's 16 values per lane
  h8 bh[2], bl[2];
#pragma nounroll
  for (int i0 = 0; i0 < NF; i0 += 2) {
    // ---- hidden block i0 (conv1 operands in K0).  Its second unit (-> M1): the prologue's in a first pass' block 0; with two conv3 units
    //      a second pass' block 0 finds unit 1 requested by Z_0 and asks for unit 0 (-> M0) itself
    if (i0 > 0 || (SECOND && !G::M0LAST)) {
#ifdef GLOWK_EXP_COHALFDMA
      stage4<G::UNITP / 2, 73>(G::main_unit(c.img, PASS, i0, 1), c.m1, c.w4, c.voff);
#else
      stage4<G::UNITP, 73>(G::main_unit(c.img, PASS, i0, 1), c.m1, c.w4, c.voff);
#endif
    }
    if constexpr (SECOND && G::M0LAST) {
      if (i0 == 0) stage4<G::UNITP, 77>(G::main_unit(c.img, PASS, 0, 0), c.m0, c.w4, c.voff);
    }
    if (PASS == 0 && i0 == 2) GLOWK_STAMP(a, 2);
    h3s_X<KIN, MOUT, NF, G::MODEX, 2, 0, PASS, G::STORE>(a, hc, i0, xh, xl, lane, bh, bl);
    if (PASS == 0 && i0 == 2) GLOWK_STAMP(a, 3);
    if (SECOND && G::M0LAST && i0 == 0) co_end<0>();            // (the unit this very op asked for)
    else {
#ifdef GLOWK_EXP_COHALFDMA
      co_end_st<G::PPW / 2>(st);
#else
      co_end_st<G::PPW + XST>(st);                    // unit 2 i0 (M0) landed; this op's unit (and its stores) may still be in flight
#endif
    }
    if (PASS == 0 && i0 == 2) GLOWK_STAMP(a, 4);
    co_Y<NFH, MODE7, 0, 1, S::K1P>(c.m0, bh, bl, acc2, lane, hc.k1img + (size_t)((i0 + 2) % NF) * S::K14, hc.k1s0, c.w4, c.voff);
    if (PASS == 0 && i0 == 2) GLOWK_STAMP(a, 5);
    co_end_st<G::K1PW + XST>(st);                       // unit 2 i0 + 1 (M1) landed (asked for before X's stores: they may stay in flight)
    if (PASS == 0 && i0 == 2) GLOWK_STAMP(a, 6);
    co_Y<NFH, MODE7, 1, 2, S::K1P>(c.m1, bh, bl, acc2, lane, G::main_unit(c.img, PASS, i0 + 1, 0), c.m0, c.w4, c.voff);
    if (PASS == 0 && i0 == 2) GLOWK_STAMP(a, 7);
    co_bar();                                           // (M1 free for the next X's unit)
    if (PASS == 0 && i0 == 2) GLOWK_STAMP(a, 8);
    // ---- hidden block i0 + 1 (conv1 operands in K1)
#ifdef GLOWK_EXP_COHALFDMA
    stage4<G::UNITP / 2, 74>(G::main_unit(c.img, PASS, i0 + 1, 1), c.m1, c.w4, c.voff);
#else
    stage4<G::UNITP, 74>(G::main_unit(c.img, PASS, i0 + 1, 1), c.m1, c.w4, c.voff);
#endif
    h3s_X<KIN, MOUT, NF, G::MODEX, 2, 1, PASS, G::STORE>(a, hc, i0 + 1, xh, xl, lane, bh, bl);
#ifdef GLOWK_EXP_COHALFDMA
    co_end_st<G::PPW / 2>(st);
#else
    co_end_st<G::PPW + XST>(st);
#endif
    co_Y<NFH, MODE7, 0, 3, S::K1P>(c.m0, bh, bl, acc2, lane, hc.k1img + (size_t)((i0 + 3) % NF) * S::K14, hc.k1s1, c.w4, c.voff);
    co_end_st<G::K1PW + XST>(st);
    // (after the last block: conv3 unit 1 takes the place of "unit 2 NF")
    co_Y<NFH, MODE7, 1, 4, S::K1P>(c.m1, bh, bl, acc2, lane, i0 + 2 < NF ? G::main_unit(c.img, PASS, i0 + 2, 0) : G::out_unit(c.img, PASS, 1), c.m0,
                                   c.w4, c.voff);
    co_bar();
  }
  f32x4 acc3[G::NMT][2];
  GLOWK_STAMP(a, 9 + (PASS == 0 ? 0 : 1));                  // (9: pass 0's blocks done, 10: pass 1's)
  co_Z<KIN, MOUT, NF, MODE, PASS, SOLO, 0>(a, hc, c, epl, pl, acc2, acc3, bh, bl, q, qok, lane, kq, keep);
  co_Z<KIN, MOUT, NF, MODE, PASS, SOLO, 1>(a, hc, c, epl, pl, acc2, acc3, bh, bl, q, qok, lane, kq, keep);
  if constexpr (G::NMT >= 3) co_Z<KIN, MOUT, NF, MODE, PASS, SOLO, 2>(a, hc, c, epl, pl, acc2, acc3, bh, bl, q, qok, lane, kq, keep);
  if constexpr (G::NMT >= 4) co_Z<KIN, MOUT, NF, MODE, PASS, SOLO, 3>(a, hc, c, epl, pl, acc2, acc3, bh, bl, q, qok, lane, kq, keep);
  if constexpr (G::NMT >= 5) co_Z<KIN, MOUT, NF, MODE, PASS, SOLO, 4>(a, hc, c, epl, pl, acc2, acc3, bh, bl, q, qok, lane, kq, keep);
}

template <int KIN, int MOUT, int NF, int MODE, bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_net_h3c(NetArgs a) {
  using G = RingC<KIN, MOUT, NF, MODE>;
  using S = typename G::S;
  constexpr int KS = G::KS;
  constexpr int SGN = G::BWD ? -1 : 1;                 // backward gathers at q - d(tap)
  static_assert(G::FITS, "shape");
  static_assert(!(G::FUSE && SPLIT), "the fused coupling needs both passes in one workgroup");

  const int tid = threadIdx.x;
  GLOWK_STAMP(a, 0);
  // (aligned 1024: the LDS layout is sorted by alignment first, so the DMA targets take the lowest addresses and the fused form's
  //  copy of P the highest)
  __shared__ __attribute__((aligned(1024))) float4 slotM0[G::UNIT4];
  __shared__ __attribute__((aligned(1024))) float4 slotM1[G::UNIT4];
  __shared__ __attribute__((aligned(1024))) float4 slotD[G::UNIT4];
  __shared__ __attribute__((aligned(1024))) float4 k1slot0[S::K14];
  __shared__ __attribute__((aligned(1024))) float4 k1slot1[S::K14];
  float* epl = nullptr;                                 // conv2 accumulator init | per-row constants of P (forward modes)
  if constexpr (!G::BWD) { __shared__ float epl_arr[S::EPN]; epl = epl_arr; }
  unsigned short* mkl = nullptr;                        // backward: the forward pass's ReLU decisions of the workgroup's four 32-pixel blocks
  if constexpr (G::BWD) { __shared__ __attribute__((aligned(1024))) unsigned short mkl_arr[G::MASKN]; mkl = mkl_arr; }
  float* plds = nullptr;                                // fused coupling: the workgroup's per-tap outputs ...
  float4* vstash = nullptr;                             // ... and its pixels' four input channels
  if constexpr (G::FUSE) {
    __shared__ float plds_arr[36 * CO_PSTR];
    __shared__ float4 vstash_arr[CO_PX];
    plds = plds_arr; vstash = vstash_arr;
  }
#ifdef GLOWK_EXP_COPAD     // (diagnostic build: extra LDS per workgroup -- does a failure follow the LDS footprint / the number of resident workgroups?)
  __shared__ float copad[GLOWK_EXP_COPAD];
  if (a.Q < 0) copad[tid] = 1.0f;
#endif
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15;
  const int kq = lane >> 4;
  const int qbase = ((int)blockIdx.x * 4 + wave) * 32;
  // (training form: adjacent pixels per lane -- its hidden stores are 8-byte pairs; masks and P follow q, so the pair of launches agrees)
  const int q[2] = {G::STORE ? qbase + 2 * n16 : qbase + n16, G::STORE ? qbase + 2 * n16 + 1 : qbase + 16 + n16};
  const bool qok[2] = {q[0] < a.Q, q[1] < a.Q};
  const int pass0 = SPLIT ? (int)blockIdx.y : 0;       // the (first) pass this workgroup runs

  H3Ctx hc;                    // what h3s_X reads: conv1 operand slots and image, masks
  hc.sA = hc.sB = hc.sD = nullptr; hc.k1s0 = k1slot0; hc.k1s1 = k1slot1;
  hc.k1img = a.RSp; hc.img = a.RSp; hc.mkl = mkl; hc.mk2off = 4 * NF * 64; hc.pl = plds;
  hc.wblk = (size_t)blockIdx.x * 4 + wave;             // this wave's 32-pixel block: the unit of the ReLU-mask arrays
  hc.wok = (long)hc.wblk * 32 < a.Q;
  hc.w4 = wave; hc.voff = (unsigned)lane * 16u; hc.ub[0] = hc.ub[1] = 1.0f;
  CoCtx c;
  c.m0 = slotM0; c.m1 = slotM1; c.d = slotD; c.img = a.RSp; c.w4 = wave; c.voff = (unsigned)lane * 16u;

  stage4<G::UNITP, 60>(G::main_unit(c.img, pass0, 0, 0), slotM0, c.w4, c.voff);
  stage4<G::UNITP, 61>(G::main_unit(c.img, pass0, 0, 1), slotM1, c.w4, c.voff);
  stage4<G::UNITP, 62>(G::out_unit(c.img, pass0, 0), slotD, c.w4, c.voff);
  stage4<S::K1P, 63>(hc.k1img, k1slot0, c.w4, c.voff);
  stage4<S::K1P, 64>(hc.k1img + S::K14, k1slot1, c.w4, c.voff);
  if constexpr (G::BWD) {      // [mask1 | mask2][wave][hidden block][lane]: 4 NF 64 entries = NF / 2 pieces each
    stage4<NF / 2, 65>(reinterpret_cast<const float4*>(a.mask1 + (size_t)blockIdx.x * 4 * NF * 64), reinterpret_cast<float4*>(mkl), c.w4, c.voff);
    stage4<NF / 2, 66>(reinterpret_cast<const float4*>(a.mask2 + (size_t)blockIdx.x * 4 * NF * 64), reinterpret_cast<float4*>(mkl + 4 * NF * 64), c.w4, c.voff);
  }

  // im2col fragments of this lane's two pixels: k-step s holds k = 32 s + 8 kq + j (natural order), scaled and split
  h8 xh[KS][2], xl[KS][2];
  float xmax = 0.0f;                 // range guard: largest |network input| (scaled) this lane gathers
  {
    const int hw = a.h * a.w;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int qq = qok[hf] ? q[hf] : 0;
      const int rem = qq % hw;
      const int i = rem / a.w, j0 = rem % a.w;
      const float* base = a.vin + (long)qq * a.in_stride + a.in_off;
      if constexpr (G::BWD) {
        // linear network: the pixel's gradient vector (held by its four lanes kq = 0..3) is normalised by a power of two
        float v[KS][8];
        float pm = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          gather8<KIN, false, SGN>(base, i, j0, a.h, a.w, a.in_stride, qok[hf], 32 * s + 8 * kq, v[s]);
          pm = range8(pm, v[s]);
        }
        pm = nan_max(pm, __shfl_xor(pm, 16, 64));
        pm = nan_max(pm, __shfl_xor(pm, 32, 64));
        xmax = nan_max(xmax, pm);
        const float fac = G::STORE ? 1.0f : pixel_norm(pm, a.bnorm, hc.ub[hf]);     // (training: one scale per launch, BwdArgs::go_scale)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[s][j] *= fac;
          split8(v[s], xh[s][hf], xl[s][hf]);
        }
      } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          float v[8];
          gather8<KIN, true, SGN>(base, i, j0, a.h, a.w, a.in_stride, qok[hf], 32 * s + 8 * kq, v);
          xmax = range8(xmax, v);
          split8(v, xh[s][hf], xl[s][hf]);
        }
      }
    }
  }
  if constexpr (G::FUSE) {
    if (tid < CO_PX) {
      const int qv = (int)blockIdx.x * CO_PX + tid;
      vstash[tid] = qv < a.Q ? *reinterpret_cast<const float4*>(a.vin + (size_t)qv * 4) : float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if constexpr (!G::BWD)
    for (int i = tid; i < S::EPN; i += 256) epl[i] = a.eph[i];
  // forward: the static bound; backward (normalised per pixel): only a non-finite gradient can leave the range
  if (((G::BWD && !G::STORE) ? !(xmax <= 3.0e38f) : !(xmax <= a.xlim)) && a.flag) *a.flag = 1;
  if (a.xmax_out) range_probe(a.xmax_out, xmax);
#ifdef GLOWK_EXP_COPAD
  if (a.Q < -1 && a.flag) *a.flag = (int)copad[tid ^ 1];
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  GLOWK_STAMP(a, 1);

  f32x4 keep[G::NMT][2];
  if constexpr (SPLIT) {
    if (pass0 == 0) co_pass<KIN, MOUT, NF, MODE, 0, true>(a, hc, c, epl, plds, xh, xl, q, qok, lane, kq, keep);
    else co_pass<KIN, MOUT, NF, MODE, 1, true>(a, hc, c, epl, plds, xh, xl, q, qok, lane, kq, keep);
  } else {
    co_pass<KIN, MOUT, NF, MODE, 0, false>(a, hc, c, epl, plds, xh, xl, q, qok, lane, kq, keep);
    co_pass<KIN, MOUT, NF, MODE, 1, false>(a, hc, c, epl, plds, xh, xl, q, qok, lane, kq, keep);
  }
  GLOWK_STAMP(a, 12);
  if constexpr (G::FUSE) {
    __syncthreads();       // every wave's LDS writes of P are complete and visible (lgkmcnt(0) + barrier)
#ifdef GLOWK_EXP_COCHECK    // (diagnostic build: does the LDS copy of P still hold what this wave wrote?  A mismatch raises the range flag;
                            //  the values are written again before the tail reads them)
    {
      bool bad = false;
#pragma unroll
      for (int ml = 0; ml < 3; ++ml)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = ml * 16 + 4 * kq + r;
            if (m < MOUT) {
              float* p = &plds[m * CO_PSTR + wave * 32 + 16 * hf + (lane & 15)];
              if (__float_as_uint(*p) != __float_as_uint(keep[ml][hf][r])) bad = true;
              *p = keep[ml][hf][r];
            }
          }
      if (bad && a.flag) *a.flag = 1;
      __syncthreads();
    }
#endif
    fused_couple<CO_PX, CO_PSTR>(a, plds, vstash, tid);
  }
  GLOWK_STAMP(a, 13);
}
