// k_net_h3q: the small-grid form of the split coupling network with the conv1 blocks FIRST (round-3 verdict, item 3).
//
// At the reference's batch sizes (30 mixture tiles, training batch 32) the deeper levels launch a few dozen workgroups, each alone on its
// CU, and a launch of the half-wave form (k_net_h3s<..., MODE | 32, 4, true>: 128 pixels, one pass per workgroup) is a chain of 16
// dependent X_i -> Y_i phase pairs, 27-31 us per launch whatever the batch.  X_i does not depend on Y_{i-1} -- only Y accumulates -- so
// this form takes X out of the chain, the hidden width in two halves (so that NF/2 x 8 registers hold split B fragments at a time):
//     X half  conv1 + activation + split of NF/2 hidden blocks in batches of XB (2 XB interleaved MFMA chains; conv1 operands through a
//             double-buffered LDS slot pair per batch); the B fragments stay in REGISTERS -- possible only at one 16-pixel half per wave
//     Y half  their NF/2 conv2 k-steps back to back, two per barrier: MFMAs (dependent ones four apart) and A-fragment reads pipelined
//             across the chunk boundary; the next pair's two chunks land in the other half of a ring of four slots
//     ... the second half likewise (the operand ring and the chunk ring keep running across the seams), then
//     Z       conv3 as in k_net_h3s (h3s_tail is reused as is: slots A / B / D keep its conventions).
// All eight waves run the same op; the wave groups only split the DMA duties (group 0: weight chunks, group 1: conv1 operands).  Every op
// ends with counted vmcnt + lgkmcnt(0) + raw barrier (glowk_co.h: the rule for DMA that spans barriers).  Same weight images, same
// per-accumulator MFMA order as the half-wave form: results are bit for bit equal (tests/test_gpu_small_grid_form.py).
//
// What it measured (in-kernel stamps, glowk_debug_stamps / scripts/q_stamps.py, 30 tiles, level 1 forward): prologue 2.4 us, X halves
// 2.6 + 2.2, Y halves 5.7 + 5.6, Z 2.8 = 21.7 us in the kernel (half-wave form: ~26).  The Y halves do not move whatever the barrier
// count or the MFMA order: 0.65 us per k-step is the LDS -- at one pixel half per wave every wave reads the whole 16-KiB chunk for its
// 24 MFMAs, 128 KiB per k-step and CU against 768 MFMA cycles, i.e. 130 % of the LDS bandwidth (the half-wave form's known price).
// Fewer LDS bytes per MFMA need two pixel halves per wave (then the B fragments no longer fit the registers) or fewer waves per
// workgroup (then level 1 at 30 tiles needs 480 workgroups): the form gains 2-3 % on the 30-tile calls, not the 15 % its model promised.
#pragma once
#include "glowk_kernels.h"

template <int KIN, int MOUT, int NF, int MODE>
struct RingQ {
  using S = RingS<KIN, MOUT, NF, MODE | 32, 4>;        // geometry, weight image and epilogue constants of the half-wave four-pass form
  static constexpr int MODE7 = MODE & 7;
  static constexpr bool BWD = MODE7 == NET_BWD;
  static constexpr int KS = S::KS, NFH = S::NFH, NRB = S::NRB;
  static constexpr int EPN = BWD ? 4 : S::EPN;          // (the backward network has no epilogue constants)
  static constexpr int MASKB = BWD ? NF * 512 * 2 : 0;  // backward: ONE buffer of ReLU masks for the workgroup's 8 pixel blocks: mask2 (the ReLU after
                                                        // conv2) while X-all runs, then mask1 (after conv1) for Z -- [wave][hidden block][lane]
  static constexpr size_t FIXED = (size_t)5 * S::MAIN4 * 16 + (size_t)EPN * 4 + MASKB + 64;      // slots A, B, R2, R3, D + constants + masks
  static constexpr int XB = ((size_t)4 * S::K14 * 16 + FIXED <= 160 * 1024) ? 2 : 1;              // hidden blocks per X batch (2: four MFMA chains)
  static constexpr size_t LDS_BYTES = FIXED + (size_t)2 * XB * S::K14 * 16;
  static constexpr int NB = NF / XB;                    // X batches
  static constexpr int PPW = S::MAINP / 4;              // DMA instructions per group-0 wave and chunk
  static constexpr int GS = NRB % 4 == 0 ? 4 : 2;       // row blocks per group of MFMAs in Y (3 GS MFMAs, dependent ones GS apart)
  static constexpr int XST = ((MODE7 == NET_FWD_SAVE) ? 1 : 0) + ((MODE & 8) ? 8 : 0);     // stores a pass-0 wave issues per X (ReLU mask, hidden values)
  static constexpr bool FITS = S::FITS && LDS_BYTES <= 160 * 1024 && NF % (2 * XB) == 0 && NF % 4 == 0 && KS <= 5 && S::MAINP % 4 == 0 && NF >= 8 &&
                               (MODE7 == NET_FWD || MODE7 == NET_FWD_SAVE || MODE7 == NET_BWD) && !(MODE & 16);
};

// op end: all but this wave's N youngest vector-memory operations are done, its LDS reads have retired, workgroup barrier
template <int N>
__device__ __forceinline__ void q_end() {
  __builtin_amdgcn_s_waitcnt((N & 15) | 0x0070 | ((N >> 4) << 14));   // vmcnt(N) lgkmcnt(0)
  h3_barrier();
}

struct QSlots {
  float4 *r[4];            // chunk ring: r[0] = slot A, r[1] = slot B (h3s_tail's names), r[2], r[3]
  float4 *k[2][2];         // conv1 operand slots [batch parity][block within the batch]
  unsigned short* mk;      // backward: the mask buffer
};

// X batch B: conv1 + activation + split of hidden blocks B XB .. B XB + XB - 1 (operands in k[B & 1][*]).  The 2 XB accumulator chains
// (block, row block) are interleaved term by term, so dependent MFMAs sit 2 XB apart; per chain the order (k-step, then lo.hi, hi.lo,
// hi.hi) is that of h3s_X: bit for bit the same hidden activations.
template <int KIN, int MOUT, int NF, int MODE, int PASS, int B>
__device__ __forceinline__ void q_X_batch(const NetArgs& a, const H3Ctx& c, const QSlots& sl, const h8 (&xh)[(RingQ<KIN, MOUT, NF, MODE>::KS)][2],
                                          const h8 (&xl)[(RingQ<KIN, MOUT, NF, MODE>::KS)][2], int g, int lane, h8 (&bfh)[NF / 2], h8 (&bfl)[NF / 2]) {
  using G = RingQ<KIN, MOUT, NF, MODE>;
  using S = typename G::S;
  constexpr int XB = G::XB, KS = G::KS, MODE7 = G::MODE7;
  // group 1: the conv1 operands of batch B + 1 into the slots batch B - 1 has left (batches 0 and 1: the prologue's)
  if constexpr (B >= 1 && B + 1 < G::NB) {
    if (g) {
      stage4<S::K1P, 100 + 2 * B>(c.k1img + (size_t)((B + 1) * XB) * S::K14, sl.k[(B + 1) & 1][0], c.w4, c.voff);
      if constexpr (XB == 2) stage4<S::K1P, 101 + 2 * B>(c.k1img + (size_t)((B + 1) * XB + 1) * S::K14, sl.k[(B + 1) & 1][1], c.w4, c.voff);
    }
  }
  // group 0, first batch: chunks 2 and 3 of the ring (0 and 1 came with the prologue)
  if constexpr (B == 0) {
    if (!g) {
      stage4<S::MAINP, 84>(S::main_chunk(c.img, PASS, 2), sl.r[2], c.w4, c.voff);
      stage4<S::MAINP, 85>(S::main_chunk(c.img, PASS, 3), sl.r[3], c.w4, c.voff);
    }
  }
  const h8* k1[2] = {reinterpret_cast<const h8*>(sl.k[B & 1][0]) + lane, reinterpret_cast<const h8*>(sl.k[B & 1][XB - 1]) + lane};   // [s][row block][hi|lo][64]
  unsigned mask[2] = {0u, 0u};
  if constexpr (G::BWD) {
#pragma unroll
    for (int x = 0; x < XB; ++x) mask[x] = sl.mk[((size_t)(threadIdx.x >> 6) * NF + B * XB + x) * 64 + lane];    // mask2: the ReLU after conv2
  }
  f32x4 h1[2][2];   // [block][row block]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) h1[i >> 1][i & 1][r] = 0.0f;
  h8 A[2][2][2][2];   // [buffer][block][row block][hi|lo]
  auto load = [&](int buf, int s) {
#pragma unroll
    for (int x = 0; x < XB; ++x)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        A[buf][x][rb][0] = k1[x][((s * 2 + rb) * 2 + 0) * 64];
        A[buf][x][rb][1] = k1[x][((s * 2 + rb) * 2 + 1) * 64];
      }
  };
  load(0, 0);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + 1 < KS) load((s + 1) & 1, s + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int x = 0; x < XB; ++x)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const h8& av = A[s & 1][x][rb][t == 0 ? 1 : 0];            // lo . hi, then hi . lo, hi . hi
          const h8& bv = t == 1 ? xl[s][0] : xh[s][0];
          h1[x][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, h1[x][rb], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int x = 0; x < XB; ++x) {
    const int fi = B * XB + x;
    const int stq = (int)c.wblk * 16 + (lane & 15);
    h8 bh, bl;
    const unsigned bits = h3s_act<MODE7, (MODE & 8) != 0>(h1[x][0], h1[x][1], a.sc1, mask[x], bh, bl, (MODE & 8) && PASS == 0 && stq < a.Q,
                                                         (MODE & 8) ? uniform_fptr(a.st1 + (size_t)fi * 32 * a.Q) : nullptr,
                                                         ((unsigned)(4 * (lane >> 4)) * (unsigned)a.Q + (unsigned)stq) * 4u, (unsigned)a.Q * 4u);
    if (MODE7 == NET_FWD_SAVE && PASS == 0 && c.wok) a.mask1[(c.wblk * NF + fi) * 64 + lane] = (unsigned short)bits;
    bfh[fi % (NF / 2)] = bh;
    bfl[fi % (NF / 2)] = bl;
  }
  // the next batch's operands (issued at the top of this batch by group 1) have landed; younger: this batch's stores (pass 0 of a
  // saving / training launch), which stay in flight
  constexpr int NST = PASS == 0 ? XB * G::XST : 0;
  if (NST != 0 && !c.wok) q_end<0>();            // (a wave without a valid pixel issued no store: its youngest operations are the DMA)
  else q_end<NST>();
}

template <int KIN, int MOUT, int NF, int MODE, int PASS, int OFF, int... B>
__device__ __forceinline__ void q_X_all(const NetArgs& a, const H3Ctx& c, const QSlots& sl, const h8 (&xh)[(RingQ<KIN, MOUT, NF, MODE>::KS)][2],
                                        const h8 (&xl)[(RingQ<KIN, MOUT, NF, MODE>::KS)][2], int g, int lane, h8 (&bfh)[NF / 2], h8 (&bfl)[NF / 2],
                                        std::integer_sequence<int, B...>) {
  (q_X_batch<KIN, MOUT, NF, MODE, PASS, OFF + B>(a, c, sl, xh, xl, g, lane, bfh, bfl), ...);
}

// Y pair P: conv2 k-steps of hidden blocks 2 P and 2 P + 1 (chunks in ring slots (2 P) & 3, (2 P + 1) & 3) with ONE barrier at the end; the A
// fragments of the next group of row blocks -- across the chunk boundary too -- are read while the current group's MFMAs run; group 0
// requests the next pair's two chunks (beyond the last main chunk: conv3 chunk 1, which h3s_tail expects in slot A = r[0]) into the
// other two slots, early in the pair, so that they have most of it to land.
template <int KIN, int MOUT, int NF, int MODE, int PASS, int P>
__device__ __forceinline__ void q_Y_pair(const H3Ctx& c, const QSlots& sl, const h8 (&bfh)[NF / 2], const h8 (&bfl)[NF / 2],
                                         f32x4 (&acc2)[(RingQ<KIN, MOUT, NF, MODE>::NRB)][2], int g, int lane) {
  using G = RingQ<KIN, MOUT, NF, MODE>;
  using S = typename G::S;
  constexpr int I = 2 * P, NRB = G::NRB, GS = G::GS, NGC = NRB / GS, NGRP = 2 * NGC;      // groups per chunk, per pair
  constexpr bool REQ_MAIN = P >= 1 && I + 3 < NF;                  // chunks I + 2, I + 3 (pair 0: already requested by the prologue / X-all)
  constexpr bool REQ_OUT1 = I + 2 == NF && S::NCH >= 2;            // conv3 chunk 1 -> slot (I + 2) & 3 = 0 = A  (NF % 4 == 0)
  static_assert(!REQ_OUT1 || ((I + 2) & 3) == 0, "conv3 chunk 1 must land in slot A");
  const h8* buf[2] = {reinterpret_cast<const h8*>(sl.r[I & 3]) + lane, reinterpret_cast<const h8*>(sl.r[(I + 1) & 3]) + lane};
  h8 A[2][GS][2];
  auto load = [&](int b, int gi) {          // group gi of the pair: chunk gi / NGC, row blocks (gi % NGC) GS ...
    const h8* p = buf[gi / NGC];
#pragma unroll
    for (int j = 0; j < GS; ++j) {
      const int rb = (gi % NGC) * GS + j;
      A[b][j][0] = p[(rb * 2 + 0) * 64];
      A[b][j][1] = p[(rb * 2 + 1) * 64];
    }
  };
  load(0, 0);
#pragma unroll
  for (int gi = 0; gi < NGRP; ++gi) {
    if (gi + 1 < NGRP) load((gi + 1) & 1, gi + 1);
    if constexpr (REQ_MAIN || REQ_OUT1) {
      // this wave's pieces of the requested chunk(s): chunk 0 of the request in group 0, chunk 1 in group 1
      if (!g && gi < (REQ_MAIN ? 2 : 1)) {
        const float4* src = REQ_MAIN ? S::main_chunk(c.img, PASS, I + 2 + gi) : S::out_chunk(c.img, PASS, 1);
        const char* mb = uniform_ptr(src);
        float4* dst = sl.r[(I + 2 + gi) & 3];
#pragma unroll
        for (int e = 0; e < G::PPW; ++e) {
          const int piece = e * 4 + c.w4;
          glds16(reinterpret_cast<const float4*>(mb + (size_t)piece * 1024 + c.voff), dst + piece * 64);
        }
      }
    }
    asm volatile("; dma site %0" ::"n"(400 + P * 8 + gi));
    __builtin_amdgcn_sched_barrier(0);
    const h8& bh = bfh[(I + gi / NGC) % (NF / 2)];
    const h8& bl = bfl[(I + gi / NGC) % (NF / 2)];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < GS; ++j) {
        const int o = (gi % NGC) * GS + j;
        acc2[o][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[gi & 1][j][t == 0 ? 1 : 0], t == 1 ? bl : bh, acc2[o][0], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  q_end<0>();           // the next pair's chunks have landed, this pair's reads have retired
}

template <int KIN, int MOUT, int NF, int MODE, int PASS, int OFF, int... P>
__device__ __forceinline__ void q_Y_all(const H3Ctx& c, const QSlots& sl, const h8 (&bfh)[NF / 2], const h8 (&bfl)[NF / 2],
                                        f32x4 (&acc2)[(RingQ<KIN, MOUT, NF, MODE>::NRB)][2], int g, int lane, std::integer_sequence<int, P...>) {
  (q_Y_pair<KIN, MOUT, NF, MODE, PASS, OFF + P>(c, sl, bfh, bfl, acc2, g, lane), ...);
}

template <int KIN, int MOUT, int NF, int MODE, int PASS>
__device__ __forceinline__ void q_pass(const NetArgs& a, const H3Ctx& c, const QSlots& sl, const float* epl,
                                       const h8 (&xh)[(RingQ<KIN, MOUT, NF, MODE>::KS)][2], const h8 (&xl)[(RingQ<KIN, MOUT, NF, MODE>::KS)][2], int g,
                                       const int (&q)[2], const bool (&qok)[2], int lane, int kq) {
  using G = RingQ<KIN, MOUT, NF, MODE>;
  using S = typename G::S;
  // The hidden width goes through in two halves -- X of blocks [0, NF/2), their NF/2 conv2 k-steps, X of the second half, its k-steps --
  // so that only NF/2 x 8 registers hold B fragments at a time (all NF at once spilled in the backward shapes).  Nothing in one half
  // waits for the other: the conv1 operand ring and the chunk ring simply keep running across the seams.
  h8 bfh[NF / 2], bfl[NF / 2];
  constexpr int f2base = PASS * G::NFH * 32;
  f32x4 acc2[G::NRB][2];
#pragma unroll
  for (int ob = 0; ob < G::NRB; ++ob)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float b = G::BWD ? 0.0f : epl[f2base + ob * 16 + 4 * kq + r];   // conv2 bias (scaled)
      acc2[ob][0][r] = b;
      acc2[ob][1][r] = b;
    }
  GLOWK_STAMP(a, 2);
  q_X_all<KIN, MOUT, NF, MODE, PASS, 0>(a, c, sl, xh, xl, g, lane, bfh, bfl, std::make_integer_sequence<int, G::NB / 2>());
  GLOWK_STAMP(a, 3);
  q_Y_all<KIN, MOUT, NF, MODE, PASS, 0>(c, sl, bfh, bfl, acc2, g, lane, std::make_integer_sequence<int, NF / 4>());
  GLOWK_STAMP(a, 6);
  q_X_all<KIN, MOUT, NF, MODE, PASS, G::NB / 2>(a, c, sl, xh, xl, g, lane, bfh, bfl, std::make_integer_sequence<int, G::NB / 2>());
  if constexpr (G::BWD) {                    // mask2 is done with: the same buffer takes mask1 (read by the Z ops, many barriers from here)
    if (!g) stage4<NF, 86>(reinterpret_cast<const float4*>(a.mask1 + (size_t)blockIdx.x * 8 * NF * 64), reinterpret_cast<float4*>(sl.mk), c.w4, c.voff);
  }
  GLOWK_STAMP(a, 7);
  q_Y_all<KIN, MOUT, NF, MODE, PASS, NF / 4>(c, sl, bfh, bfl, acc2, g, lane, std::make_integer_sequence<int, NF / 4>());
  GLOWK_STAMP(a, 4);
  f32x4 acc3[S::G0N][2];
  f32x4 keep[S::G0N][2];                     // (never touched: a solo pass writes its own partial P)
  h8 bh[2], bl[2];
  h3s_tail<KIN, MOUT, NF, MODE | 32, 4, 0, PASS, true>(a, c, epl, acc2, acc3, bh, bl, g, q, qok, lane, kq, keep, std::make_integer_sequence<int, 2 * S::NCH>());
  GLOWK_STAMP(a, 5);
}

// grid (workgroups of 128 pixels, 4 passes); each workgroup runs ONE pass
template <int KIN, int MOUT, int NF, int MODE>
__global__ __launch_bounds__(512, 2) void k_net_h3q(NetArgs a) {
  using G = RingQ<KIN, MOUT, NF, MODE>;
  using S = typename G::S;
  constexpr int KS = G::KS, XB = G::XB;
  constexpr int SGN = G::BWD ? -1 : 1;
  static_assert(G::FITS, "shape");

  __shared__ __attribute__((aligned(1024))) float4 slotA[S::MAIN4];
  __shared__ __attribute__((aligned(1024))) float4 slotB[S::MAIN4];
  __shared__ __attribute__((aligned(1024))) float4 slotR2[S::MAIN4];
  __shared__ __attribute__((aligned(1024))) float4 slotR3[S::MAIN4];
  __shared__ __attribute__((aligned(1024))) float4 slotD[S::MAIN4];
  __shared__ __attribute__((aligned(1024))) float4 k1e0[S::K14];
  __shared__ __attribute__((aligned(1024))) float4 k1e1[XB == 2 ? S::K14 : 4];
  __shared__ __attribute__((aligned(1024))) float4 k1o0[S::K14];
  __shared__ __attribute__((aligned(1024))) float4 k1o1[XB == 2 ? S::K14 : 4];
  __shared__ float epl[G::EPN];
  __shared__ __attribute__((aligned(1024))) unsigned short mkl[G::MASKB / 2 + 8];

  const int tid = threadIdx.x;
  GLOWK_STAMP(a, 0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2;
  const int n16 = lane & 15;
  const int kq = lane >> 4;
  const int qbase = ((int)blockIdx.x * 8 + (tid >> 6)) * 16;
  const int q[2] = {qbase + n16, qbase + n16};
  const bool qok[2] = {q[0] < a.Q, false};
  const int pass = (int)blockIdx.y;

  H3Ctx c;
  c.sA = slotA; c.sB = slotB; c.sD = slotD; c.k1s0 = k1e0; c.k1s1 = k1e1;
  c.k1img = a.RSp; c.img = a.RSp; c.mkl = mkl; c.mk2off = 0; c.pl = nullptr;
  c.wblk = (size_t)blockIdx.x * 8 + wave;
  c.wok = (long)c.wblk * 16 < a.Q;
  c.w4 = wave & 3;
  c.voff = (unsigned)lane * 16u;
  c.ub[0] = c.ub[1] = 1.0f;
  QSlots sl;
  sl.r[0] = slotA; sl.r[1] = slotB; sl.r[2] = slotR2; sl.r[3] = slotR3;
  sl.k[0][0] = k1e0; sl.k[0][1] = XB == 2 ? k1e1 : k1e0; sl.k[1][0] = k1o0; sl.k[1][1] = XB == 2 ? k1o1 : k1o0;
  sl.mk = mkl;

  if (!g) {
    stage4<S::MAINP, 60>(S::main_chunk(c.img, pass, 0), slotA, c.w4, c.voff);
    stage4<S::MAINP, 61>(S::main_chunk(c.img, pass, 1), slotB, c.w4, c.voff);
    stage4<S::MAINP, 62>(S::out_chunk(c.img, pass, 0), slotD, c.w4, c.voff);
  } else {
    // conv1 operands of batches 0 and 1
    stage4<S::K1P, 63>(c.k1img, k1e0, c.w4, c.voff);
    if constexpr (XB == 2) stage4<S::K1P, 66>(c.k1img + S::K14, k1e1, c.w4, c.voff);
    stage4<S::K1P, 67>(c.k1img + (size_t)XB * S::K14, k1o0, c.w4, c.voff);
    if constexpr (XB == 2) stage4<S::K1P, 68>(c.k1img + (size_t)(XB + 1) * S::K14, k1o1, c.w4, c.voff);
    if (G::BWD)     // the ReLU decisions after conv2 of this workgroup's 8 pixel blocks (X-all reads them; Z's come later: q_pass)
      stage4<NF, 65>(reinterpret_cast<const float4*>(a.mask2 + (size_t)blockIdx.x * 8 * NF * 64), reinterpret_cast<float4*>(mkl), c.w4, c.voff);
  }
  // im2col fragments of this lane's pixel: k-step s holds k = 32 s + 8 kq + j, scaled and split
  h8 xh[KS][2], xl[KS][2];
  float xmax = 0.0f;
  {
    const int hw = a.h * a.w;
    const int qq = qok[0] ? q[0] : 0;
    const int rem = qq % hw;
    const int i = rem / a.w, j0 = rem % a.w;
    const float* base = a.vin + (long)qq * a.in_stride + a.in_off;
    if constexpr (G::BWD) {
      float v[KS][8];
      float pm = 0.0f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        gather8<KIN, false, SGN>(base, i, j0, a.h, a.w, a.in_stride, qok[0], 32 * s + 8 * kq, v[s]);
        pm = range8(pm, v[s]);
      }
      pm = nan_max(pm, __shfl_xor(pm, 16, 64));
      pm = nan_max(pm, __shfl_xor(pm, 32, 64));
      xmax = pm;
      const float fac = (MODE & 8) ? 1.0f : pixel_norm(pm, a.bnorm, c.ub[0]);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[s][j] *= fac;
        split8(v[s], xh[s][0], xl[s][0]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        float v[8];
        gather8<KIN, true, SGN>(base, i, j0, a.h, a.w, a.in_stride, qok[0], 32 * s + 8 * kq, v);
        xmax = range8(xmax, v);
        split8(v, xh[s][0], xl[s][0]);
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) { xh[s][1] = xh[s][0]; xl[s][1] = xl[s][0]; }      // (second half: unused at one half per wave)
  }
  if (!G::BWD)
    for (int i = tid; i < G::EPN; i += 512) epl[i] = a.eph[i];
  if (((G::BWD && !(MODE & 8)) ? !(xmax <= 3.0e38f) : !(xmax <= a.xlim)) && a.flag) *a.flag = 1;
  if (a.xmax_out) range_probe(a.xmax_out, xmax);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  GLOWK_STAMP(a, 1);

  if (pass == 0) q_pass<KIN, MOUT, NF, MODE, 0>(a, c, sl, epl, xh, xl, g, q, qok, lane, kq);
  else if (pass == 1) q_pass<KIN, MOUT, NF, MODE, 1>(a, c, sl, epl, xh, xl, g, q, qok, lane, kq);
  else if (pass == 2) q_pass<KIN, MOUT, NF, MODE, 2>(a, c, sl, epl, xh, xl, g, q, qok, lane, kq);
  else q_pass<KIN, MOUT, NF, MODE, 3>(a, c, sl, epl, xh, xl, g, q, qok, lane, kq);
}
