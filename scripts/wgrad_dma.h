// Experiment (not part of the product build): the square weight-gradient GEMM of a split sweep on PRE-SPLIT operands, staged by
// LDS-DMA.  Built and measured in round 3 (scripts/wgrad_bench.hip, DMA=1): bitwise the results of k_wgrad_h3, and
// no faster -- 306 against 300 TFLOP/s on the level-0 shape: what the LDS store path cost the register-staged kernel, the issue of
// the 1 KB DMA pieces (~200 cycles each inside the MFMA groups, six per wave and round) costs this one; with the pieces not issued
// at all (H3D_EXP_NOISSUE) the same loop runs at 471.  The producers' side (16-bit stores of the split halves from the packed
// registers of k_net_h3 / k_net_h3s) was written too and is not kept.
#pragma once
#include "glowk_train.h"

// ---- the square GEMM of a split sweep on PRE-SPLIT operands (NetArgs::st1_h16): C2 = [R1; 1] . M2^T -----------------------------------
// Both operands were stored by the split kernels as two fp16 planes [F][K] (hi, lo: the halves those kernels multiplied with), so a
// round's tiles go global -> LDS by LDS-DMA: no conversion, no register round trip, no LDS store instruction (measured on k_wgrad_h3,
// level-0 shape: 304 TFLOP/s as built, 359 with the conversion removed, 600 with the LDS stores removed as well -- the store path
// VGPR -> LDS, ~80 B/clk per CU, was the bound).  One 1 KB piece = 16 rows x 32 k of one plane; the 16-byte chunk c of row r lands at
// position c ^ ((r >> 2) & 3) of the row's 64 bytes (the DMA writes lanes in order, so the swizzle is applied to the SOURCE address a
// lane fetches): the fragment reads -- ds_read_b128, lane groups {0-3, 12-15, 20-27}, ... -- then hit 64 distinct banks.  Three
// buffers (144 KB for 256 x 128): round r computes from buffer r % 3 while the pieces of rounds r + 1 and r + 2 are in flight; one
// barrier per round (after it buffer (r - 1) % 3 is free and round r's pieces of every wave have landed: each wave waits for its own
// with a counted vmcnt first).  Rounds past the slice's end re-fetch the last round (branch-free: the counts stay constant), drained
// before the workgroup ends.  Same products in the same order as k_wgrad_h3 on the fp32 arrays: bitwise the same C.
// Needs M % TM == 0, N % TN == 0, K % 32 == 0 (launch_wgrad checks; otherwise the producers store fp32 and k_wgrad_h3 runs).
struct WgradDmaArgs {
  const unsigned short* A;   // hi plane of A [M][K]; lo plane loA halves further
  const unsigned short* B;
  size_t loA, loB;
  int M, N, K;
  int kslice, S, tm, tn;
  ptrdiff_t bsA, bsB;        // halves between the batch entries
  float* Cpart;
  size_t csz;
  int b_sums;                // 1: row M of every output = the row sums of B over the slice
};

template <int TMv, int TNv>
struct WgradDmaLds {         // one buffer: four plane tiles, rows of 32 halves (64 bytes, chunk-swizzled)
  _Float16 ah[TMv * 32], al[TMv * 32], bh[TNv * 32], bl[TNv * 32];
};

template <int WTM, int WTN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 1) void k_wgrad_h3d(WgradDmaArgs a) {
  constexpr int NW = WM * WN, TM = 32 * WTM * WM, TN = 32 * WTN * WN;
  constexpr int PA = TM / 16, PB = TN / 16, NPIECE = 2 * (PA + PB), PW = NPIECE / NW;   // 1 KB pieces per round: per plane tile, in all, per wave
  static_assert(NPIECE % NW == 0, "pieces divide over the waves");
  static_assert(WTM * WTN == 4, "six groups of four MFMAs");
  // three DISTINCT static objects with static roles (as the ring slots of k_net_h3: otherwise hipcc orders every fragment read behind
  // every DMA in flight with vmcnt(0))
  __shared__ __attribute__((aligned(1024))) WgradDmaLds<TM, TN> buf0;
  __shared__ __attribute__((aligned(1024))) WgradDmaLds<TM, TN> buf1;
  __shared__ __attribute__((aligned(1024))) WgradDmaLds<TM, TN> buf2;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int wg = xcd_contiguous_id();
  const int tiles = a.tm * a.tn, tile = wg % tiles, zz = wg / tiles;
  const int m0 = (tile % a.tm) * TM, n0 = (tile / a.tm) * TN;
  const int b = zz / a.S, s = zz % a.S;
  const long k_begin = (long)s * a.kslice;
  const long k_end = k_begin + a.kslice < a.K ? k_begin + a.kslice : a.K;
  const int R = (int)((k_end - k_begin) >> 5);                 // rounds of this slice
  // this wave's pieces: piece p = wave + NW * i; per lane the source of round 0 (row = 16 * block + lane / 4, swizzled chunk)
  const unsigned short* src[PW];
  unsigned dst[PW];                                            // byte offset of the piece inside a buffer
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int p = wave + NW * i;
    const bool isA = p < 2 * PA;
    const int pp = isA ? p : p - 2 * PA, per = isA ? PA : PB;
    const int plane = pp / per, blk = pp % per;
    const int row = blk * 16 + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
    const unsigned short* base = isA ? a.A + (ptrdiff_t)b * a.bsA + (plane ? a.loA : 0) + (size_t)(m0 + row) * a.K
                                     : a.B + (ptrdiff_t)b * a.bsB + (plane ? a.loB : 0) + (size_t)(n0 + row) * a.K;
    src[i] = base + k_begin + chunk * 8;
    dst[i] = (unsigned)((isA ? (plane ? TM * 32 : 0) : 2 * TM * 32 + (plane ? TN * 32 : 0)) + blk * 512) * 2u;
  }
  auto issue = [&](auto& bufo, int i, int r) {                 // piece i of round r (clamped to the last round) into bufo
    const int rc = r < R ? r : R - 1;
    glds16(reinterpret_cast<const float4*>(src[i] + (size_t)rc * 32),
           reinterpret_cast<float4*>(reinterpret_cast<char*>(&bufo) + __builtin_amdgcn_readfirstlane(dst[i])));
  };
  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int i32 = lane & 31, kh = lane >> 5, swz = (i32 >> 2) & 3;
  const bool sums = a.b_sums && m0 == 0 && wm == 0;            // (the waves of the first row tile that own distinct B rows)
  double bsum[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) bsum[j] = 0.0;
  typedef _Float16 h2t __attribute__((ext_vector_type(2)));
  auto round = [&](auto& cur, auto& nxt2, int r) {             // compute round r from cur; its groups issue round r + 2 into nxt2
    __builtin_amdgcn_s_waitcnt((PW & 15) | 0x0F70 | ((PW >> 4) << 14));   // vmcnt(PW): this wave's pieces of round r have landed
    __builtin_amdgcn_s_barrier();
#ifdef H3D_EXP_ISSUE_FIRST   // (diagnostic: all pieces of round r + 2 right after the barrier)
#pragma unroll
    for (int i = 0; i < PW; ++i) issue(nxt2, i, r + 2);
#endif
    h8 fah[2][WTM], fal[2][WTM], fbh[2][WTN], fbl[2][WTN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int co = ((2 * ks + kh) ^ swz) * 8;
#pragma unroll
      for (int i = 0; i < WTM; ++i) {
        const int off = ((wm * WTM + i) * 32 + i32) * 32 + co;
        fah[ks][i] = *reinterpret_cast<const h8*>(cur.ah + off);
        fal[ks][i] = *reinterpret_cast<const h8*>(cur.al + off);
      }
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        const int off = ((wn * WTN + j) * 32 + i32) * 32 + co;
        fbh[ks][j] = *reinterpret_cast<const h8*>(cur.bh + off);
        fbl[ks][j] = *reinterpret_cast<const h8*>(cur.bl + off);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      const int ks = g / 3, prod = g % 3;
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(prod == 0 ? fal[ks][i] : fah[ks][i], prod == 1 ? fbl[ks][j] : fbh[ks][j], acc[i][j], 0, 0, 0);
#if !defined(H3D_EXP_NOISSUE) && !defined(H3D_EXP_ISSUE_FIRST)
#pragma unroll
      for (int i = 0; i < PW; ++i)
        if ((i * 6) / PW == g) issue(nxt2, i, r + 2);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    if (sums) {   // row sums of B: 16 k of hi + lo per lane, k-step and row block, in fp32 (v_dot2 with ones), then fp64 across the rounds
      const h2t one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        float t = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            t = __builtin_amdgcn_fdot2(h2t{fbh[ks][j][e], fbh[ks][j][e + 1]}, one, t, false);
            t = __builtin_amdgcn_fdot2(h2t{fbl[ks][j][e], fbl[ks][j][e + 1]}, one, t, false);
          }
        bsum[j] += (double)t;
      }
    }
  };
  if (R > 0) {
#pragma unroll
    for (int i = 0; i < PW; ++i) issue(buf0, i, 0);
#pragma unroll
    for (int i = 0; i < PW; ++i) issue(buf1, i, 1);
    int r = 0;
    for (; r + 3 <= R; r += 3) {
      round(buf0, buf2, r);
      round(buf1, buf0, r + 1);
      round(buf2, buf1, r + 2);
    }
    if (r < R) round(buf0, buf2, r);
    if (r + 1 < R) round(buf1, buf0, r + 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the re-fetched rounds past the end have landed before the LDS is given back
  }
  float* C = a.Cpart + (size_t)zz * a.csz;
  if (sums) {
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      double t = bsum[j];
      t += __shfl_xor(t, 32, 64);
      const int n = n0 + (wn * WTN + j) * 32 + i32;
      if (kh == 0 && n < a.N) C[(size_t)a.M * a.N + n] = (float)t;
    }
  }
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * WTM + i) * 32 + mfma_row(r, kh), n = n0 + (wn * WTN + j) * 32 + i32;
        C[(size_t)m * a.N + n] = acc[i][j][r];
      }
}

