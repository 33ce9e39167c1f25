// glowk device code: hand-written gfx950 (CDNA4) kernels of the Glow forward / inverse path.
//
// Layout: every level tensor is NHWC fp32 [Q = N*h*w pixels][c channels] in HBM.  One flow step is
//   k_net     (coupling network, >99 % of the FLOPs, MFMA)   v[.., c/2:]  -> P  [9c][Q] (per-tap partial conv3)
//   k_couple  (gather the 9 taps, tanh/exp, affine coupling, per-sample log-det, next step's ActNorm+1x1)
// plus a handful of index kernels at block boundaries (squeeze / split / latent scatter).
//
// k_net is a chain of three transposed GEMMs per 32-pixel column block held by ONE wavefront:
//   A1^T[F x 32px]  = K1r^T [F x 9ci]  . im2col(vb)^T [9ci x 32px]      (conv1 3x3, flow_tfk_layers.py:56-60)
//   A2^T[F x 32px]  = K2^T  [F x F]    . H1^T [F x 32px]                (conv2 1x1, :63-65)
//   P^T [9c x 32px] = K3r^T [9c x F]   . H2^T [F x 32px]                (conv3 3x3 as 9 per-tap 1x1s, :68-70)
// with bias+ReLU+BN (inference affine, :61,66) applied on the accumulators in registers.  In the
// transposed orientation the 32x32 accumulator tile of one GEMM (column = pixel on the lane, rows in the
// 16 registers) IS the B operand of the next GEMM's v_mfma_f32_32x32x2_f32 (k pair = rows rho, rho+4), so
// the 512-channel hiddens never leave the register file: no LDS round trip, no HBM traffic.
// Weights stream global -> LDS by LDS-DMA (global_load_lds_dwordx4) through a two-slot ring of 64 KiB chunks that also
// carries conv1's operands, pre-packed on the host into the exact lane order the MFMA A operand wants
// (ds_read_b128, conflict free).  Three rules learned on the hardware shape this kernel:
//  1. the ring slots are distinct static __shared__ objects with static roles, otherwise hipcc puts s_waitcnt vmcnt(0)
//     between every global_load_lds and the next ds_read and the DMA never overlaps the MFMAs;
//  2. no VGPR-destination memory instruction (global load, scratch reload) may be issued while a DMA is in flight:
//     vmcnt is an in-order counter, each such load is serialised behind the whole DMA;
//  3. fp32-input MFMA and the FP32 VALU share one datapath: VALU FMAs do not hide under v_mfma_f32_32x32x2_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GLOWK_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define GLOWK_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// row of a 32x32 MFMA accumulator tile held in register r by a lane of half hh (= lane >> 5)
__device__ __forceinline__ int mfma_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

struct NetArgs {
  const float* vin;      // [Q][in_stride]; network input = channels [in_off, in_off + CI)
  int in_stride, in_off;
  int Q, h, w;
  const float* K1p;      // [NF fi][KS1][64 lanes]
  const float* ep;       // [6][F]: b1, g1, d1, b2, g2, d2   (BN folded: g = gamma/sqrt(var+eps), d = beta - mean*g)
  const float4* R0p;     // ring image (Ring1): [NF + NMT] slots
  unsigned short* mask1; // [blocks of 32 px][NF][64 lanes] ReLU mask bits of conv1 (NET_FWD_SAVE writes, NET_BWD reads)
  unsigned short* mask2; // same for conv2
  float* P;              // [9*CO][Q]
  size_t pstride;        // f16x3: the pass-p partial sums of P go to P + p * pstride (the consumer adds the partials)
  int max_np;            // host side: how many partial buffers P has room for (caps the number of passes)
  const float4* RHp;     // f16x3 image (RingH) of the network this launch runs (forward or backward), or null
  const float* eph;      // forward: [conv2 accumulator init (F) | per-row constants of P (32 NMT)], see pack_step
  const float4* RSp;     // image for the 16x16x32 kernel (RingS; same constants and scales) of the network this launch runs, or null
  int fam16;             // host side: this level's saving forward and backward launches both have a 16x16x32 image
  float sc1, sc2, sc3;   // f16x3: 2^-S of the three layers' weight scales (sc3 also undoes the activation scale)
  int* flag;             // sticky range flag of the handle, or null
  float xlim;            // split kernels: raise the flag when a gathered input (times GLOWK_ACT_SCALE) exceeds this magnitude
  float bnorm;           // split kernels, backward network (linear in its input): every pixel's gathered gradient vector is scaled by the
                         // power of two that brings its largest magnitude into [bnorm, 2 bnorm) before the split, and the pixel's outputs
                         // are scaled back (exact): no gradient magnitude can leave the fp16 range, small pixels keep all their bits
  // ---- coupling fused into k_net_h3s<..., MODE | 16> (fused_couple): what k_couple would have been given
  int fuse;              // host side: ask the launch policy for the fused instance (it answers 100 instead of a number of partials;
                         // 101: the co-resident form of it, 128-pixel workgroups -- k_couple_edge is told, EdgeArgs::pxw)
  int co;                // host side: the co-resident form (glowk_co.h) may be taken where it has an instance
  const float* fz_b3;    // [C] conv3 bias
  const float* fz_A;     // post affine [C][C] or null (forward: the NEXT step's ActNorm + 1x1; inverse: this step's inverse 1x1 + ActNorm)
  const float* fz_b;     // [C]
  float* fz_out;         // element (q, co) at fz_out[q * fz_out_stride + fz_out_off + co] (stride and offset multiples of 4)
  int fz_out_stride, fz_out_off, fz_inverse;
  float* fz_osave;       // saving pass: [Q][2] the pre-tanh log_s inputs (CoupleArgs::o_save), or null
  float* fz_edge;        // [workgroup][4][64][4]: partial sums of its first / last pixel row, contributions to the rows above / below
  double* fz_ldpart;     // per workgroup (h w >= 256) or per sample: sum of log_s over the pixels completed in the kernel; or null
  unsigned* xmax_out;    // diagnostic (glowk_range_probe_begin), normally null: the largest gathered |input| (times GLOWK_ACT_SCALE) of
                         // this launch, as float bits (non-negative floats order like unsigned ints), one atomic per wave
  // training (k_net_f32<..., STORE = true>): the two hidden tensors of this launch, PLANAR [F][Q] (the layout whose rows are the
  // K-contiguous operands of the weight-gradient GEMMs, glowk_train.h).  Forward: st1 = relu(conv1 + b1), st2 = relu(conv2 + b2)
  // (before BatchNorm); backward: st1 = mask2 * conv3^T(g_o) (gradient wrt relu2's output), st2 = mask1 * (K2 g_a2) (wrt relu1's)
  float* st1;
  float* st2;
  unsigned long long* dbg;   // diagnostic (glowk_debug_stamps), normally null: in-kernel time stamps of workgroup (0, 0)'s first lane
};

// time stamp i of the launch (constant 100 MHz counter), in builds with -DGLOWK_STAMPS (python -c "import __graft_entry__ as g;
// g.build(tag='stamps', extra_flags=['-DGLOWK_STAMPS'])", then GLOWK_LIB=.../libglowk_stamps.so): the product kernels carry none
#ifdef GLOWK_STAMPS
#define GLOWK_STAMP(a, i) do { if ((a).dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) (a).dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GLOWK_STAMP(a, i) do { } while (0)
#endif

// cache modifier of the planar hidden stores: they are streamed (gigabytes per launch, read back by the weight-gradient GEMMs after the
// level's sweep), so non-temporal -- measured -1.7 % on a 256-tile parameter-gradient sweep against "" on one box, neutral at 32 tiles
#ifndef GLOWK_ST_MOD
#define GLOWK_ST_MOD " nt"
#endif
// planar store of one 32 x 32 accumulator tile of a hidden block (16 registers per lane: rows 0-3, 8-11, 16-19, 24-27, + 4 for the
// upper lane half) into a [F][Q] array.  The block base is wave-uniform (SGPR pair), the lane's part a running 32-bit byte
// offset: global_store_dword voffset, data, saddr.  Written as asm because hipcc turns the C form into sixteen hoisted
// 64-bit per-lane addresses (1.9 KB of spills, each reload queued behind the weight DMA: the storing kernels ran 2x slower).
__device__ __forceinline__ void st_tile_planar(float* blk_base, unsigned lane_byte, unsigned row_bytes, const f32x16& v) {
  const unsigned long long base = reinterpret_cast<unsigned long long>(blk_base);
  unsigned off = lane_byte;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    asm volatile("global_store_dword %0, %1, %2" GLOWK_ST_MOD ::"v"(off), "v"(v[r]), "s"(base) : "memory");
    off += ((r & 3) == 3 ? 5u : 1u) * row_bytes;
  }
}

// per-pixel affine y = x A + b (the fused ActNorm + 1x1 of a step, or its inverse): the fma chain every kernel that applies it uses, so that
// they agree bit for bit (light kernels, fused_couple's variant, the training kernels)
template <int C>
__device__ __forceinline__ void affine_cc(const float* __restrict__ A, const float* __restrict__ b, const float (&x)[C], float (&y)[C]) {
#pragma unroll
  for (int co = 0; co < C; ++co) y[co] = b[co];
#pragma unroll
  for (int ci = 0; ci < C; ++ci)
#pragma unroll
    for (int co = 0; co < C; ++co) y[co] = fmaf(x[ci], A[ci * C + co], y[co]);
}

// one 64-lane LDS-DMA piece: LDS destination = wave-uniform base + lane*16, global source per lane
__device__ __forceinline__ void glds16(const float4* src_lane, float4* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(GLOWK_GPTR(src_lane), GLOWK_LPTR(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ float* uniform_fptr(float* p) {            // the same for a pointer stores go through
  const unsigned long long b = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ const char* uniform_ptr(const void* p) {   // force a wave-uniform pointer into SGPRs
  const unsigned long long b = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

// The two ring slots are DISTINCT static __shared__ arrays and every step names its slot statically (template
// parity): LLVM's waitcnt pass then knows that a ds_read of one slot cannot alias the LDS-DMA in flight into the
// other one.  With a single array (or a runtime-selected slot) it inserts s_waitcnt vmcnt(0) in front of the first
// ds_read after every global_load_lds, i.e. the weight DMA never overlaps the MFMAs (measured: 158 vs 127 us/tile).
template <int P>
__device__ __forceinline__ float4* ring_slot(float4* s0, float4* s1) { return P ? s1 : s0; }

// geometry of k_net_f32's ring: slot = main chunk (NF * 4 KiB) + the small convolution's MFMA A operands of the NEXT
// hidden block ([KS][64 lanes] floats, padded to whole 1-KiB wave pieces).  KIN = input channels of the small conv.
template <int KIN, int NF>
struct Ring1 {
  static constexpr int KS1 = (9 * KIN) / 2;
  static constexpr int MAIN4 = NF * 256;
  static constexpr int K1PIECES_FULL = (KS1 * 256 + 1023) / 1024;
  // the small conv's operands ride in the ring when the LDS allows it; otherwise (9*KIN = 144 at n_filters 512 only) they
  // get ONE buffer of their own, refilled during the main contraction of the previous block, at the price of a second
  // barrier per step (the buffer is requested and consumed between the same two ring barriers)
  static constexpr bool K1_IN_RING = (size_t)2 * (MAIN4 + K1PIECES_FULL * 64) * 16 + (size_t)6 * NF * 32 * 4 <= 160 * 1024;
  static constexpr int K1PIECES = K1_IN_RING ? K1PIECES_FULL : 0;
  static constexpr bool K1_IN_BUF = !K1_IN_RING && (size_t)2 * MAIN4 * 16 + (size_t)6 * NF * 32 * 4 + (size_t)K1PIECES_FULL * 1024 <= 160 * 1024;
  // K = 288 at c = 32 (the 4-level graphs' last level, backward) fits neither: its operands pass through a buffer of HALF a
  // block -- first half fetched during the previous block's main contraction, second half between the two halves of the
  // small conv (an exposed 18-KiB DMA per step; loading them from global memory instead cost 4x: every load queues behind
  // the chunk DMA and 144 live im2col registers leave no room to batch them)
  static constexpr int K1HALF_KS = KS1 / 2;
  static constexpr int K1HALF_PIECES = (K1HALF_KS * 256 + 1023) / 1024;
  static constexpr bool K1_IN_HALF = !K1_IN_RING && !K1_IN_BUF && KS1 % 2 == 0 && (K1HALF_KS * 256) % 1024 == 0 &&
                                     (size_t)2 * MAIN4 * 16 + (size_t)6 * NF * 32 * 4 + (size_t)K1HALF_PIECES * 1024 <= 160 * 1024;
  static constexpr int K1BUF4 = K1_IN_BUF ? K1PIECES_FULL * 64 : K1_IN_HALF ? K1HALF_PIECES * 64 : 1;   // float4 of the separate buffer
  static constexpr int SLOT4 = MAIN4 + K1PIECES * 64;
  static constexpr int PIECES = NF * 4 + K1PIECES;
  static constexpr size_t LDS_BYTES = (size_t)2 * SLOT4 * 16 + (size_t)6 * NF * 32 * 4 + ((K1_IN_BUF || K1_IN_HALF) ? (size_t)K1BUF4 * 16 : 0);
  static constexpr bool FITS = LDS_BYTES <= 160 * 1024;
};

// pieces [P0, P0 + N) of a ring slot image, dealt round-robin to the 4 waves (wave is scalar; surplus re-fetches the last)
template <int P0, int N>
__device__ __forceinline__ void stage_range(const float4* __restrict__ src, float4* dst, int wave, unsigned voff) {
  const char* gb = uniform_ptr(src);
#pragma unroll
  for (int i = 0; i < (N + 3) / 4; ++i) {
    int p = i * 4 + wave;
    p = P0 + (p < N ? p : N - 1);
    glds16(reinterpret_cast<const float4*>(gb + (size_t)p * 1024 + voff), dst + p * 64);
  }
}

// k_net_f32 modes.  The backward pass of the coupling network has the forward pass's structure with transposed weights:
//   NET_FWD / NET_FWD_SAVE   conv1 (K = 9ci chain) -> bias+ReLU+BN1 -> conv2 (16 accumulator tiles) -> bias+ReLU+BN2 -> conv3 per-tap
//   NET_BWD                  conv3^T (K = 9c chain) -> x g2 x mask2  -> conv2^T (16 accumulator tiles) -> x g1 x mask1 -> conv1^T per-tap
// NET_FWD_SAVE also stores the two ReLU masks (16 bits per lane per hidden block) for NET_BWD.
enum { NET_FWD = 0, NET_FWD_SAVE = 1, NET_BWD = 2,
       NET_FWD2 = 3 };   // k_net_h3s only: plain forward with TWO split terms per product (activations rounded to fp16 once,
                         // weights still hi + lo): ~1e-5-class log_prob instead of fp32-class, 2/3 of the MFMAs

// select word j of a small register array with a wave-uniform index (v_cndmask chain; no dynamic register indexing)
template <int N>
__device__ __forceinline__ unsigned pick_word(const unsigned (&w)[N], int j) {
  unsigned r = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) r = (j == i) ? w[i] : r;
  return r;
}

// contribution of hidden block fi to the 16 accumulator tiles: small conv + epilogue first, then 256 MFMAs reading slot P,
// with the DMA of the next chunk into slot P^1.  The small conv's A operands were published with the PREVIOUS chunk
// (slot P^1, read before the barrier that frees it).
template <int KIN, int NF, int P, int MODE, bool STORE>
__device__ __forceinline__ void net_step(const NetArgs& a, int fi, bool first, const float4* nsrc, float4* s0, float4* s1, float4* k1buf,
                                         const float* epl, const float (&xcol)[(9 * KIN) / 2], f32x16 (&acc2)[NF],
                                         const unsigned (&mk)[NF / 2], size_t wblk, int wave, unsigned voff, int lane, int hh, int q, bool qok, bool wok) {
  using G = Ring1<KIN, NF>;
  constexpr int KS1 = G::KS1;
  constexpr int F = NF * 32;
  f32x16 h1;
#pragma unroll
  for (int r = 0; r < 16; ++r) h1[r] = 0.0f;
  if constexpr (G::K1_IN_HALF) {
    constexpr int KH = G::K1HALF_KS;
    if (!first) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // chunk fi and the first half of this block's small-conv operands landed in every wave's view
    }
    const float* k1 = reinterpret_cast<const float*>(k1buf) + lane;
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[ks * 64], xcol[ks], h1, 0, 0, 0);
    __syncthreads();     // every wave has read the first half
    stage_range<0, G::K1HALF_PIECES>(reinterpret_cast<const float4*>(a.K1p + ((size_t)fi * KS1 + KH) * 64), k1buf, wave, voff);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();     // second half landed (nothing else is in flight here)
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[ks * 64], xcol[KH + ks], h1, 0, 0, 0);
  } else {
    if (G::K1_IN_BUF && !first) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // chunk fi and this block's small-conv operands landed in every wave's view
    }
    const float* k1 = G::K1_IN_RING ? reinterpret_cast<const float*>(ring_slot<P ^ 1>(s0, s1) + G::MAIN4) + lane
                      : G::K1_IN_BUF ? reinterpret_cast<const float*>(k1buf) + lane
                                     : a.K1p + (size_t)fi * KS1 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[ks * 64], xcol[ks], h1, 0, 0, 0);
  }
  float* st_blk = STORE ? uniform_fptr(a.st1 + (size_t)fi * 32 * a.Q) : nullptr;                   // rows 32 fi .. of the planar array
  const unsigned st_lane = STORE ? ((unsigned)(4 * hh) * (unsigned)a.Q + (unsigned)q) * 4u : 0u;   // this lane's pixel, row 4 hh
  const unsigned st_row = (unsigned)a.Q * 4u;
  if (MODE == NET_BWD) {
    // g_a2 = g_h2 * g2 * [a2 + b2 > 0]
    const unsigned w = pick_word<NF / 2>(mk, fi >> 1) >> ((fi & 1) * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] = ((w >> r) & 1u) ? h1[r] : 0.0f;
    if (STORE && qok) st_tile_planar(st_blk, st_lane, st_row, h1);
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] *= epl[4 * F + fi * 32 + mfma_row(r, hh)];
  } else {
    unsigned bits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = fi * 32 + mfma_row(r, hh);
      const float pre = h1[r] + epl[f];
      if (MODE == NET_FWD_SAVE) bits |= (pre > 0.0f ? 1u : 0u) << r;
      h1[r] = fmaxf(pre, 0.0f);
    }
    if (STORE && qok) st_tile_planar(st_blk, st_lane, st_row, h1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = fi * 32 + mfma_row(r, hh);
      h1[r] = fmaf(epl[F + f], h1[r], epl[2 * F + f]);
    }
    if (MODE == NET_FWD_SAVE) a.mask1[(wblk * NF + fi) * 64 + lane] = (unsigned short)bits;
  }
  {
    // ALSO at the first step: block 0's small-conv operands were staged behind slot 1's main part, and the DMA below rewrites
    // that slot -- without this barrier a wave still in conv1(0) could read block 2's operands (seen as run-to-run
    // differences of the K = 72 backward kernel in 1 % of the calls: its 36 MFMAs leave the widest window)
    // The stores of this step (STORE: 16 planar rows; NET_FWD_SAVE: the mask word) were issued AFTER the DMA that must have
    // landed, and vmcnt counts in order: waiting until only that many operations are outstanding completes every DMA piece
    // without waiting for the stores' round trip to HBM (with vmcnt(0) the storing kernels ran 2x slower at small grids).
    // (a wave without a single pixel issues none of them -- the compiler branches around exec = 0 -- and waits for everything)
    constexpr int NST = (STORE ? 16 : 0) + (MODE == NET_FWD_SAVE ? 1 : 0);
    if (!first || G::K1_IN_BUF) {   // (nothing is in flight at the first step)
      if (NST && wok) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();   // chunk fi landed in every wave's view; slot P^1 (and the separate operand buffer) is no longer read
  }
  stage_range<0, G::PIECES>(nsrc, ring_slot<P ^ 1>(s0, s1), wave, voff);
  if (G::K1_IN_BUF && fi + 1 < NF)
    stage_range<0, G::K1PIECES_FULL>(reinterpret_cast<const float4*>(a.K1p + (size_t)(fi + 1) * KS1 * 64), k1buf, wave, voff);
  if constexpr (G::K1_IN_HALF) {
    if (fi + 1 < NF) stage_range<0, G::K1HALF_PIECES>(reinterpret_cast<const float4*>(a.K1p + (size_t)(fi + 1) * KS1 * 64), k1buf, wave, voff);
  }
  const float4* buf = ring_slot<P>(s0, s1);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int g = 0; g < NF / 4; ++g) {
      const float4 wv = buf[(r * (NF / 4) + g) * 64 + lane];
      acc2[4 * g + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, h1[r], acc2[4 * g + 0], 0, 0, 0);
      acc2[4 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, h1[r], acc2[4 * g + 1], 0, 0, 0);
      acc2[4 * g + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, h1[r], acc2[4 * g + 2], 0, 0, 0);
      acc2[4 * g + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, h1[r], acc2[4 * g + 3], 0, 0, 0);
    }
  }
}

// One workgroup = 128 pixels (4 waves x 32-pixel column blocks), all weights through the LDS ring (image R0p: slot c < NF =
// main chunk c | small-conv operands of hidden block c+1; slot NF + mt = output chunk mt).  No VGPR-destination memory op is
// issued while a DMA is in flight (hipcc would serialise each one behind the DMA with s_waitcnt vmcnt(0)).
//   KIN   input channels of the small (3x3, K = 9*KIN) convolution: c/2 forward, c backward
//   MOUT  rows of the per-tap output: 9*c forward (conv3), 9*c/2 backward (conv1^T)
template <int KIN, int MOUT, int NF, int MODE, bool STORE = false>
__global__ __launch_bounds__(256, 1) void k_net_f32(NetArgs a) {
  using G = Ring1<KIN, NF>;
  constexpr int KS1 = G::KS1;         // k-steps (k = 2) of the small conv's contraction over (tap, channel)
  constexpr int NMT = (MOUT + 31) / 32;
  constexpr int F = NF * 32;
  constexpr int SLOT4 = G::SLOT4;
  constexpr int SGN = (MODE == NET_BWD) ? -1 : 1;   // backward gathers at q - d(tap)
  static_assert(KIN % 2 == 0 && NF % 4 == 0 && G::FITS, "shape");

  __shared__ float4 slot0[SLOT4];     // weight ring: two distinct objects (see ring_slot)
  __shared__ float4 slot1[SLOT4];
  __shared__ float epl[6 * F];        // b1, g1, d1, b2, g2, d2
  __shared__ float4 k1buf[G::K1BUF4];  // small-conv operands of one block when they do not fit the ring slots

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned voff = (unsigned)lane * 16u;
  const int pix = lane & 31;
  const int hh = lane >> 5;
  const size_t wblk = (size_t)blockIdx.x * 4 + (tid >> 6);   // 32-pixel column block of this wave
  const int q = (int)wblk * 32 + pix;
  const bool qok = q < a.Q;
  const bool wok = (long)(blockIdx.x * 4 + wave) * 32 < a.Q;   // this wave holds at least one pixel (scalar)
  const float4* ring = a.R0p;

  // im2col column of this lane's pixel: xcol[ks] = in[pixel +- d(tap)][ch], k = 2*ks + hh = tap*KIN + ch
  float xcol[KS1];
  {
    const int hw = a.h * a.w;
    const int qq = qok ? q : 0;
    const int rem = qq % hw;
    const int i = rem / a.w, j = rem % a.w;
    const float* base = a.vin + (long)qq * a.in_stride + a.in_off;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const int k = 2 * ks + hh;
      const int tap = k / KIN, cin = k % KIN;
      const int dy = SGN * (tap / 3 - 1), dx = SGN * (tap % 3 - 1);
      const int ii = i + dy, jj = j + dx;
      const bool ok = qok && ii >= 0 && ii < a.h && jj >= 0 && jj < a.w;
      const int off = ok ? ((dy * a.w + dx) * a.in_stride + cin) : 0;     // clamped: always in bounds
      const float v = base[off];
      xcol[ks] = ok ? v : 0.0f;
    }
  }
  // backward: ReLU masks of this column block (mk1 = conv1 output / epilogue 2, mk2 = conv2 output / epilogue 1),
  // one 32-bit word per pair of hidden blocks, kept in registers for the whole kernel
  unsigned mk1[NF / 2], mk2[NF / 2];
#pragma unroll
  for (int j = 0; j < NF / 2; ++j) { mk1[j] = 0; mk2[j] = 0; }
  if (MODE == NET_BWD) {
#pragma unroll
    for (int j = 0; j < NF / 2; ++j) {
      mk1[j] = (unsigned)a.mask1[(wblk * NF + 2 * j) * 64 + lane] | ((unsigned)a.mask1[(wblk * NF + 2 * j + 1) * 64 + lane] << 16);
      mk2[j] = (unsigned)a.mask2[(wblk * NF + 2 * j) * 64 + lane] | ((unsigned)a.mask2[(wblk * NF + 2 * j + 1) * 64 + lane] << 16);
    }
  }
  for (int i = tid; i < 6 * F; i += 256) epl[i] = a.ep[i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the gathers above are done before any DMA is issued
  stage_range<0, G::PIECES>(ring, slot0, wave, voff);                                           // chunk 0 -> slot 0
  if (G::K1_IN_RING) stage_range<NF * 4, (G::K1PIECES > 0 ? G::K1PIECES : 1)>(ring + (size_t)(NF - 1) * SLOT4, slot1, wave, voff);   // small-conv operands of block 0
  else if (G::K1_IN_BUF) stage_range<0, G::K1PIECES_FULL>(reinterpret_cast<const float4*>(a.K1p), k1buf, wave, voff);
  else if (G::K1_IN_HALF) stage_range<0, G::K1HALF_PIECES>(reinterpret_cast<const float4*>(a.K1p), k1buf, wave, voff);

  f32x16 acc2[NF];
#pragma unroll
  for (int fo = 0; fo < NF; ++fo)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[fo][r] = 0.0f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // ep visible; chunk 0 and block-0 operands landed

#pragma nounroll
  for (int fi = 0; fi < NF; fi += 2) {   // chunk fi lives in slot fi & 1 (NF is even)
    net_step<KIN, NF, 0, MODE, STORE>(a, fi, fi == 0, ring + (size_t)(fi + 1) * SLOT4, slot0, slot1, k1buf, epl, xcol, acc2, mk2, wblk, wave, voff, lane, hh, q, qok, wok);
    net_step<KIN, NF, 1, MODE, STORE>(a, fi + 1, false, ring + (size_t)(fi + 2) * SLOT4, slot0, slot1, k1buf, epl, xcol, acc2, mk2, wblk, wave, voff, lane, hh, q, qok, wok);
  }

  // ---- epilogue on the 16 accumulator tiles, in place ----
#pragma unroll
  for (int fo = 0; fo < NF; ++fo) {
    if (MODE == NET_BWD) {
      const unsigned w = mk1[fo >> 1] >> ((fo & 1) * 16);   // g_a1 = g_h1 * g1 * [a1 + b1 > 0]
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[fo][r] = ((w >> r) & 1u) ? acc2[fo][r] : 0.0f;
      if (STORE && qok) st_tile_planar(uniform_fptr(a.st2 + (size_t)fo * 32 * a.Q), ((unsigned)(4 * hh) * (unsigned)a.Q + (unsigned)q) * 4u, (unsigned)a.Q * 4u, acc2[fo]);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[fo][r] *= epl[F + fo * 32 + mfma_row(r, hh)];
    } else {
      unsigned bits = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = fo * 32 + mfma_row(r, hh);
        const float pre = acc2[fo][r] + epl[3 * F + f];
        if (MODE == NET_FWD_SAVE) bits |= (pre > 0.0f ? 1u : 0u) << r;
        acc2[fo][r] = fmaxf(pre, 0.0f);
      }
      if (STORE && qok) st_tile_planar(uniform_fptr(a.st2 + (size_t)fo * 32 * a.Q), ((unsigned)(4 * hh) * (unsigned)a.Q + (unsigned)q) * 4u, (unsigned)a.Q * 4u, acc2[fo]);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = fo * 32 + mfma_row(r, hh);
        acc2[fo][r] = fmaf(epl[4 * F + f], acc2[fo][r], epl[5 * F + f]);
      }
      if (MODE == NET_FWD_SAVE) a.mask2[(wblk * NF + fo) * 64 + lane] = (unsigned short)bits;
    }
  }

  // ---- the 3x3 output convolution as nine per-tap 1x1s: P^T[(tap,ch) x 32px] = W^T . H^T; chunk NF+mt in slot mt & 1 ----
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float4* cur = (mt & 1) ? slot1 : slot0;
    float4* nxt = (mt & 1) ? slot0 : slot1;
    if (mt + 1 < NMT) stage_range<0, NF * 4>(ring + (size_t)(NF + mt + 1) * SLOT4, nxt, wave, voff);
    const float4* buf = cur;
    f32x16 acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[r] = 0.0f;
#pragma unroll
    for (int fo = 0; fo < NF; ++fo) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float4 wv = buf[(fo * 4 + r4) * 64 + lane];
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, acc2[fo][4 * r4 + 0], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, acc2[fo][4 * r4 + 1], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, acc2[fo][4 * r4 + 2], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, acc2[fo][4 * r4 + 3], acc3, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mt * 32 + mfma_row(r, hh);
      if (m < MOUT && qok) a.P[(size_t)m * a.Q + q] = acc3[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// k_net_h3: the coupling network with every contraction as THREE fp16 MFMAs on error-compensated splits
//   x = hi + lo,  hi = fp16(x), lo = fp16(x - hi);   a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi   (fp32 accumulate)
// (fp16 products are exact in fp32; the dropped lo.lo term is 2^-22 relative) -- fp32-class accuracy at 3/16 of the
// fp32-MFMA time.  Weights are split on the host after scaling by a power of two that keeps lo out of the fp16
// subnormal range; activations are scaled by GLOWK_ACT_SCALE before splitting.  Every per-channel constant (BatchNorm, biases) is
// folded into the weights by the host (glowk.hip:pack_step), so the epilogues are  B = split(max(acc * 2^-S, 0)).
// B operands taken from accumulators use registers 8s..8s+7 as k-step s, so element j of lane half h is k-row
// 16s + 8(j>>2) + 4h + (j&3) of the tile: the host packs the A operands in that order.
//
// Geometry: 8 waves per workgroup, one 32-pixel column block each (256 pixels per workgroup); the hidden width is covered
// in NP passes of NFH = NF/NP accumulator tiles (NP = 2: 128 registers per wave; NP = 4: 64, for shapes whose small-conv
// fragments need the room and for small grids, where a pass becomes a workgroup of its own).  The same kernel runs the
// forward network, the forward network with saves (ReLU masks) and the backward network (MODE, as k_net_f32).
// A pass is a sequence of OPS, one per phase:
//     X_i  small conv of hidden block i + activation + split       (3 KS MFMAs, then ~80 VALU: MFMA pipe mostly idle)
//     Y_i  main contraction's contribution of block i to the NFH accumulators   (6 NFH MFMAs, no VALU)
//     Z_z  half a chunk of per-tap output A tiles                  (3 NFH MFMAs + activation/split of the accumulators)
// in the order X_0 Y_0 X_1 Y_1 ... Z_0 Z_1 ...; phases are separated by one workgroup barrier.  The two waves of a SIMD
// (wave w and w+4, "groups" 0 and 1) run the SAME op sequence ONE PHASE APART: while one is in the VALU-heavy X the
// other is in the MFMA-only Y, so the matrix pipe sees one Y per phase instead of idling while both waves do epilogues
// (measured before this: 5450 cycles per X+Y step against 3456 cycles of MFMA).
//
// LDS: main slots A/B (main chunks, and the later output chunks), slot D (first output chunk, loaded early), two small
// slots for the small-conv operands (+ the masks of the 8 column blocks in backward mode).  DMA is requested only during Y
// ops, spread between the MFMA groups: group 0 asks for the next main chunk (into the slot its partner group finished
// reading one phase ago), group 1 for the small-conv operands of block i+2 (into the slot its own X_i has just left); the
// op after a Y (X or Z) ends with vmcnt(0) + barrier, so a piece has two phases to land, every wait is static, and every
// slot is rewritten only after the barrier that follows its last reader.  Output chunks beyond the first are requested
// during the Z ops as soon as their slot is free; the last Z op of a pass requests the next pass's first chunks.
// ------------------------------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
// Activation scale before the fp16 split.  Overflow (a hidden activation above 65504 / scale) turns into inf/NaN, underflow only
// costs the low bits of activations below ~6e-5 / scale * 2^11: 4 leaves |activation| < 16 376 with activations down to 0.03
// fully split; log_prob accuracy measured identical for 1, 4 and 32 (scripts/act_scale_probe.py).
#ifndef GLOWK_ACT_SCALE
#define GLOWK_ACT_SCALE 4.0f
#endif

// KIN = input channels of the small 3x3 convolution (c/2 forward: conv1; c backward: conv3^T), MOUT = rows of the per-tap
// output (18 ci forward: conv3; 9 ci backward: conv1^T), MODE as for k_net_f32.
// NP = passes over the hidden width (2, or 4: a quarter of the accumulators per pass -- for shapes whose small-conv fragments
// need the registers, and for small grids, where each pass becomes a workgroup of its own).
template <int KIN, int MOUT, int NF, int MODE, int NP>
struct RingH {
  static constexpr bool BWD = (MODE & 7) == NET_BWD;
  static constexpr int NFH = NF / NP;                             // accumulator tiles (hidden 32-blocks) per pass
  static constexpr int K1 = 9 * KIN;
  static constexpr int KROWS = K1 + (BWD ? 0 : 1);                // forward: a spare k row carries conv1's bias
  static constexpr int KS = (KROWS + 15) / 16;                    // small-conv k-steps of 16
  static constexpr int MAINP = NFH * 4;                           // 1-KiB pieces per main chunk: NFH tiles x 2 k-steps x (hi, lo)
  static constexpr int MAIN4 = MAINP * 64;                        // float4 per main chunk
  static constexpr int K1P = KS * 2;                              // pieces of one block's small-conv operands
  static constexpr int K14 = K1P * 64;
  static constexpr int M3 = MOUT;
  static constexpr int NMT = (M3 + 31) / 32;
  static constexpr int G0N = NMT < 3 ? NMT : 3;                   // output row tiles are processed in fused groups of <= 3:
  static constexpr int G1N = NMT - G0N;                           // one ReLU/split of a hidden block feeds all tiles of a group
  static constexpr int G1D = G1N > 0 ? G1N : 1;
  static constexpr int EPN = (NF * 32 + 32 * NMT + 3) & ~3;       // conv2 accumulator init (F) | per-row constants of P (32 NMT)
  static constexpr int MASK2B = BWD ? 2 * NF * 512 * 2 : 0;       // backward: both ReLU masks of the workgroup's 8 column blocks
  static constexpr size_t LDS_BYTES = (size_t)3 * MAIN4 * 16 + (size_t)2 * K14 * 16 + (size_t)EPN * 4 + MASK2B;
  static constexpr bool FITS = LDS_BYTES <= 160 * 1024 && NF % 4 == 0 && NFH >= 2 && NFH % 2 == 0 && KS <= (NP == 2 ? 5 : 9) && G1N <= 3;   // (KS = 5 with NP = 2 spills a few registers outside the main loop: still 3 % faster than four passes at large batches)
  // image (one for every NP): small-conv operands of all blocks, then per HALF of the hidden width main chunks 0..NF-1 and
  // output chunks, tiles of NF/2 hidden blocks each; a pass of NP = 4 reads its half of every chunk
  static constexpr int K1TOT4 = NF * K14;
  static constexpr int IMGH = NF / 2;                             // tiles per image chunk
  static constexpr int IMG_MAIN4 = IMGH * 256;
  static constexpr int IMG_PASS4 = (NF + NMT) * IMG_MAIN4;
  static constexpr int SUBS = NP / 2;                             // passes per image half
  __device__ static const float4* main_chunk(const float4* img, int pass, int ch) {
    return img + K1TOT4 + (size_t)(pass / SUBS) * IMG_PASS4 + (size_t)ch * IMG_MAIN4 + (size_t)(pass % SUBS) * MAIN4;
  }
  __device__ static const float4* out_chunk(const float4* img, int pass, int s) {   // output chunk s of a pass (NFH tiles)
    const int sub = pass % SUBS;
    const int t2 = s < G0N ? sub * NFH * G0N + s * NFH : IMGH * G0N + sub * NFH * G1N + (s - G0N) * NFH;
    return img + K1TOT4 + (size_t)(pass / SUBS) * IMG_PASS4 + (size_t)NF * IMG_MAIN4 + (size_t)t2 * 256;
  }
  // output A tile t of a pass -> (hidden block, row tile); tiles of a group are ordered (hidden block, row tile)
  static constexpr int tile_fo(int t) { return t < NFH * G0N ? t / G0N : (t - NFH * G0N) / G1D; }
  static constexpr int tile_mt(int t) { return t < NFH * G0N ? t % G0N : G0N + (t - NFH * G0N) % G1D; }
};

// pieces [0, N) of an image of 1-KiB pieces dealt round-robin to the 4 waves of a group (w4 = wave & 3, scalar; surplus
// lanes re-fetch the last piece).  TAG is unique per call site: the marker keeps LLVM from sinking the DMA of two branches
// into one block with a phi'd LDS pointer (the slot would no longer be static and every later ds_read would get an
// s_waitcnt vmcnt(0): see ring_slot).
template <int N, int TAG>
__device__ __forceinline__ void stage4(const float4* __restrict__ src, float4* dst, int w4, unsigned voff) {
  const char* gb = uniform_ptr(src);
#pragma unroll
  for (int i = 0; i < (N + 3) / 4; ++i) {
    int p = i * 4 + w4;
    p = p < N ? p : N - 1;
    glds16(reinterpret_cast<const float4*>(gb + (size_t)p * 1024 + voff), dst + p * 64);
  }
  asm volatile("; dma site %0" ::"n"(TAG));
}

// Range guard of the split arithmetic.  A value above 65504 on its way into a split becomes hi = inf, lo = -inf, the next
// contraction turns into NaN -- and the ReLU after it (v_max_f32 returns the non-NaN operand) would silently turn that NaN
// into 0: a wrong, FINITE result.  Tracking the maximum of every split value in the epilogues was measured at 5.5 % of the
// kernel (half a VALU instruction per value: the epilogues are on the critical path), so the guard is static instead: the
// host bounds every hidden value of a step by the L1 norms of its weights as a linear function of the largest network INPUT
// (glowk.hip: step_range_limits) and hands the kernel the input magnitude `xlim` up to which no split can leave the fp16
// range; the kernel only checks the inputs it gathers (prologue, once per pixel).  Conservative (a worst-case bound: it may
// send a legitimate call to the fp32 kernels), never silent.
// (nan_max: v_max_f32 / fmaxf return the non-NaN operand, which would make a NaN input invisible to the guard -- round-3 advisor;
//  this maximum keeps a NaN from either side, so `!(xmax <= xlim)` fires on it.  Prologue only: once per gathered value.)
__device__ __forceinline__ float nan_max(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ float range8(float m, const float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) m = nan_max(m, __builtin_fabsf(v[j]));
  return m;
}

// margin of the static range guard: wave maximum of the gathered input magnitudes -> one atomic per wave (diagnostic runs only)
__device__ __forceinline__ void range_probe(unsigned* out, float xmax) {
  float m = xmax == xmax ? xmax : 3.0e38f;                 // (a NaN input counts as "beyond any limit")
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// backward network: factor (a power of two) that brings a pixel's largest gathered magnitude m into [target, 2 target), and its inverse
// `undo` for the pixel's outputs.  m = 0 (no gradient: padding, an idle lane), a subnormal or non-finite m: 1, 1.
__device__ __forceinline__ float pixel_norm(float m, float target, float& undo) {
  const unsigned E = (__float_as_uint(m) >> 23) & 0xFFu;                 // m >= 0: 2^(E - 127) <= m < 2^(E - 126)
  const unsigned T = (__float_as_uint(target) >> 23) & 0xFFu;            // target = 2^(T - 127)
  const int ef = (int)T - (int)E + 127, eu = (int)E - (int)T + 127;      // biased exponents of target / 2^(E-127) and of its inverse
  const bool ok = E >= 1u && E <= 254u && ef >= 1 && ef <= 254 && eu >= 1 && eu <= 254;
  undo = ok ? __uint_as_float((unsigned)eu << 23) : 1.0f;
  return ok ? __uint_as_float((unsigned)ef << 23) : 1.0f;
}

// x = hi + lo with hi = fp16(x), lo = fp16(x - hi), two values at a time: one packed conversion for the hi pair, two
// conversions back, one packed subtraction, one packed conversion for the lo pair (3 instructions per value; hipcc's scalar
// form of the same arithmetic converts hi twice: 4)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float (&v)[8], h8& hi, h8& lo) {
#ifdef GLOWK_SPLIT_SCALAR   // (A/B builds only, scripts/ab.py: round 1's element-wise form)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hi[j] = (_Float16)v[j];
    lo[j] = (_Float16)(v[j] - (float)hi[j]);
  }
  return;
#endif
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const f32x2 p = {v[j], v[j + 1]};
    const h2v h = __builtin_convertvector(p, h2v);
    const f32x2 d = p - __builtin_convertvector(h, f32x2);
    const h2v l = __builtin_convertvector(d, h2v);
    hi[j] = h[0]; hi[j + 1] = h[1];
    lo[j] = l[0]; lo[j + 1] = l[1];
  }
}

__device__ __forceinline__ f32x16 mfma3(const h8& ahi, const h8& alo, const h8& bhi, const h8& blo, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
  return acc;
}

// 8 consecutive im2col entries k0 .. k0+7 (k = tap * KIN + channel, k0 a multiple of 8) of one pixel, scaled for the split:
// they are G = min(KIN, 8) consecutive channels of 8 / G taps, fetched as one or two vector loads per tap (the scalar form
// -- 8 loads with their own address arithmetic -- made the prologue 15-30 us of a 45-65 us launch at small batches).
// base = the pixel's row + channel offset, aligned to G floats.  BIAS: entry k == K1 is the constant that carries conv1's bias.
template <int KIN, bool BIAS, int SGN>
__device__ __forceinline__ void gather8(const float* base, int i, int j0, int h, int w, int in_stride, bool qok, int k0, float (&v)[8]) {
  constexpr int K1 = 9 * KIN;
  constexpr int G = KIN < 8 ? KIN : 8;
  static_assert(G == 2 || G == 4 || G == 8, "channel group");
#pragma unroll
  for (int t = 0; t < 8 / G; ++t) {
    const int k = k0 + t * G;
    const int tap = k / KIN, cin = k % KIN;
    const int dy = SGN * (tap / 3 - 1), dx = SGN * (tap % 3 - 1);
    const int ii = i + dy, jj = j0 + dx;
    const bool ok = qok && k < K1 && ii >= 0 && ii < h && jj >= 0 && jj < w;
    const float* p = base + (ok ? (dy * w + dx) * in_stride + cin : 0);     // clamped: always in bounds
    float x[G];
    if constexpr (G == 8) {
      const float4 a = *reinterpret_cast<const float4*>(p), b = *(reinterpret_cast<const float4*>(p) + 1);
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else if constexpr (G == 4) {
      const float4 a = *reinterpret_cast<const float4*>(p);
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
    } else {
      const float2 a = *reinterpret_cast<const float2*>(p);
      x[0] = a.x; x[1] = a.y;
    }
#pragma unroll
    for (int e = 0; e < G; ++e) v[t * G + e] = ok ? x[e] * GLOWK_ACT_SCALE : ((BIAS && k + e == K1) ? GLOWK_ACT_SCALE : 0.0f);
  }
}

// end of a phase.  DMA is only issued by ops that end with the bare barrier (Y); the op after it (X, Z) ends with
// "everything of this wave has landed" + barrier, so a piece has two phases to land and is published by the second barrier.
// The waits are builtins so that the compiler's own wait-count bookkeeping sees them.
__device__ __forceinline__ void h3_barrier() {
  asm volatile("" ::: "memory");
#ifndef GLOWK_EXP_NOBARRIER   // (diagnostic build, racy on purpose: what do the phase barriers cost?)
  __builtin_amdgcn_s_barrier();
#endif
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void h3_wait_barrier() {
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  h3_barrier();
}

struct H3Ctx;
template <int MODE, int PASS, int NSTV = 16> __device__ __forceinline__ void h3_x_end(const H3Ctx& c);

struct H3Ctx {                      // wave-uniform pointers of the kernel (LDS arrays are distinct statics: see ring_slot)
  float4 *sA, *sB, *sD, *k1s0, *k1s1;
  const float4 *k1img, *img;
  const unsigned short* mkl;        // backward: LDS copy of the masks, [mask1 | mask2][wave][hidden block][lane] (the global order)
  int mk2off;                       // ... and where mask2 starts in it, in entries (8 NF 64 for the eight-wave workgroups, 4 NF 64 for k_net_h3c)
  size_t wblk;                      // 32-pixel column block of this wave
  bool wok;                         // it holds at least one pixel
  int w4;
  unsigned voff;
  float* pl;                        // fused coupling: LDS copy of the workgroup's per-tap outputs, [36][FUSE_PSTR]
  float ub[2];                      // backward: per-lane (per pixel) power of two the pixel's outputs are multiplied by (NetArgs::bnorm); [1] = the 16x16 family's second pixel
};

// activation epilogue of 16 accumulator values: forward max(acc * sc, 0) (+ the ReLU decisions as bits), backward
// acc * sc where the forward pass's ReLU was open
// (MODE & 8: training -- the 16 activation values, in the scaled units they are split in, also go to a planar [F][Q] array:
//  st_blk = wave-uniform base of the hidden block's 32 rows, st_lane = this lane's byte offset, st_row = bytes per row)
template <int MODE>
__device__ __forceinline__ unsigned h3_act(const f32x16& acc, float sc, unsigned mask, h8 (&bh)[2], h8 (&bl)[2], bool do_st = false,
                                           float* st_blk = nullptr, unsigned st_lane = 0, unsigned st_row = 0, float stu = 1.0f) {
  unsigned bits = 0;
  const unsigned long long st_base = reinterpret_cast<unsigned long long>(st_blk);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const int r = 8 * s + j;
      const f32x2 t2 = f32x2{acc[r], acc[r + 1]} * f32x2{sc, sc};     // v_pk_mul_f32
      const float t[2] = {t2.x, t2.y};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if ((MODE & 7) == NET_BWD) v[j + e] = __int_as_float(__float_as_int(t[e]) & __builtin_amdgcn_sbfe((int)mask, r + e, 1));   // bit ? t : 0
        else {
          v[j + e] = fmaxf(t[e], 0.0f);
          if ((MODE & 7) == NET_FWD_SAVE) bits |= (acc[r + e] > 0.0f ? 1u : 0u) << (r + e);
        }
      }
    }
#ifdef GLOWK_EXP_NOHST
    do_st = false;
#endif
    if ((MODE & 8) && do_st) {
      unsigned off = st_lane + (unsigned)(16 * s) * st_row;      // rows 8 (r >> 2) + (r & 3): registers 8 s .. 8 s + 7 = rows 16 s + {0..3, 8..11}
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sv = (MODE & 7) == NET_BWD ? v[j] * stu : v[j];      // backward: back in the pixel's own units (NetArgs::bnorm)
        asm volatile("global_store_dword %0, %1, %2" GLOWK_ST_MOD ::"v"(off), "v"(sv), "s"(st_base) : "memory");
        off += (j == 3 ? 5u : 1u) * st_row;
      }
    }
    split8(v, bh[s], bl[s]);
  }
  return bits;
}

// end of an X phase.  In pass 0 a saving launch stores the ReLU mask of the hidden block from here (1 store per wave) and a training
// launch (MODE & 8) the block itself (16 planar stores): the wave's YOUNGEST memory operations -- everything the phase has to wait
// for (the DMA requested during the previous Y) is older, and vmcnt counts in issue order, so the stores stay in flight instead of
// exposing their write latency at every one of the 16 X phases
template <int MODE, int PASS, int NSTV>
__device__ __forceinline__ void h3_x_end(const H3Ctx& c) {
  constexpr int NST = ((MODE & 8) ? NSTV : 0) + (((MODE & 7) == NET_FWD_SAVE) ? 1 : 0);   // (NSTV: value stores of a training launch per X phase)
  if constexpr (NST != 0 && PASS == 0) {
    if (c.wok) {      // (a wave without a valid pixel issues no store: its youngest operations are the DMA)
      __builtin_amdgcn_s_waitcnt((NST & 15) | 0x0F70 | ((NST >> 4) << 14));   // vmcnt(NST)
      h3_barrier();
      return;
    }
  }
  h3_wait_barrier();
}

// X: small convolution of one hidden block (operands in slot KP) + activation, as the split B fragments of the main
// contraction's two k-steps
template <int KIN, int MOUT, int NF, int MODE, int NP, int KP, int PASS>
__device__ __forceinline__ void h3_X(const NetArgs& a, const H3Ctx& c, int fi, const h8 (&xh)[(RingH<KIN, MOUT, NF, MODE, NP>::KS)],
                                     const h8 (&xl)[(RingH<KIN, MOUT, NF, MODE, NP>::KS)], int lane, h8 (&bh)[2], h8 (&bl)[2]) {
  using G = RingH<KIN, MOUT, NF, MODE, NP>;
#ifdef GLOWK_EXP_NOX   // (diagnostic build, wrong results: X does nothing -- what do conv1 and its epilogue cost a phase?)
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) { bh[s][j] = (_Float16)0.0f; bl[s][j] = (_Float16)0.0f; }
  return;
#endif
  f32x16 h1;
#pragma unroll
  for (int r = 0; r < 16; ++r) h1[r] = 0.0f;
  const h8* k1 = reinterpret_cast<const h8*>(KP ? c.k1s1 : c.k1s0) + lane;   // [s][hi|lo][64]
  unsigned mask = 0;
  if ((MODE & 7) == NET_BWD) mask = c.mkl[((size_t)(8 + (threadIdx.x >> 6)) * NF + fi) * 64 + lane];  // mask2: the ReLU after conv2
  if constexpr (G::KS <= 3) {       // all operand reads in flight before the first MFMA
    h8 kf[2 * G::KS];
#pragma unroll
    for (int i = 0; i < 2 * G::KS; ++i) kf[i] = k1[i * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < G::KS; ++s) h1 = mfma3(kf[2 * s], kf[2 * s + 1], xh[s], xl[s], h1);
  } else {
#pragma unroll
    for (int s = 0; s < G::KS; ++s) h1 = mfma3(k1[(2 * s + 0) * 64], k1[(2 * s + 1) * 64], xh[s], xl[s], h1);
  }
  const int stq = (int)c.wblk * 32 + (lane & 31);
  const unsigned bits = h3_act<MODE>(h1, a.sc1, mask, bh, bl, (MODE & 8) && PASS == 0 && stq < a.Q,
                                     (MODE & 8) ? uniform_fptr(a.st1 + (size_t)fi * 32 * a.Q) : nullptr,
                                     ((unsigned)(4 * (lane >> 5)) * (unsigned)a.Q + (unsigned)stq) * 4u, (unsigned)a.Q * 4u, c.ub[0]);
  if ((MODE & 7) == NET_FWD_SAVE && PASS == 0 && c.wok) a.mask1[(c.wblk * NF + fi) * 64 + lane] = (unsigned short)bits;
}

// Y: main contraction's contribution of one hidden block to this pass's NFH accumulator tiles (chunk in `slot`).  The wave
// is alone on the matrix pipe during Y (its SIMD partner is in X), so two accumulator tiles are interleaved to keep
// dependent MFMAs apart, the A fragments of the next 6-MFMA group are read while the current one computes (hipcc alone
// reads each fragment right before its MFMA and waits: ~1000 exposed cycles per Y), and the DMA this wave owes (group 0:
// its NFH pieces of the next main chunk, group 1: its piece of the next-but-one block's small-conv operands) is spread over
// the groups instead of delaying the first MFMA.
template <int KIN, int MOUT, int NF, int MODE, int NP, int TAG>
__device__ __forceinline__ void h3_Y(const float4* slot, const h8 (&bh)[2], const h8 (&bl)[2], f32x16 (&acc2)[NF / NP], int lane, int g,
                                     bool main_ok, const float4* main_src, float4* main_dst, const float4* k1_src, float4* k1_dst,
                                     int w4, unsigned voff) {
  using G = RingH<KIN, MOUT, NF, MODE, NP>;
  const h8* buf = reinterpret_cast<const h8*>(slot) + lane;
  const char* mb = uniform_ptr(main_src);
  constexpr int NG = G::NFH;        // groups of 6 MFMAs = (tile pair p, k-step s)
  h8 A[2][4];
  auto load = [&](h8 (&d)[4], int gi) {
    const int p = gi >> 1, s = gi & 1;
    d[0] = buf[((2 * p) * 4 + 2 * s + 0) * 64];      // tile 2p hi, lo; tile 2p+1 hi, lo
    d[1] = buf[((2 * p) * 4 + 2 * s + 1) * 64];
    d[2] = buf[((2 * p + 1) * 4 + 2 * s + 0) * 64];
    d[3] = buf[((2 * p + 1) * 4 + 2 * s + 1) * 64];
  };
  load(A[0], 0);
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int p = gi >> 1, s = gi & 1, f0 = 2 * p, f1 = 2 * p + 1;
    if (gi + 1 < NG) load(A[(gi + 1) & 1], gi + 1);
    if (!g) {
      if (main_ok) {
        const int piece = gi * 4 + w4;                // NFH pieces per wave, one per group
        glds16(reinterpret_cast<const float4*>(mb + (size_t)piece * 1024 + voff), main_dst + piece * 64);
      }
      asm volatile("; dma site %0" ::"n"(TAG * 16 + gi));
    } else if (gi == 0) {
      stage4<G::K1P, TAG * 16 + 15>(k1_src, k1_dst, w4, voff);
    }
    __builtin_amdgcn_sched_barrier(0);
    const h8 (&af)[4] = A[gi & 1];
    acc2[f0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bh[s], acc2[f0], 0, 0, 0);
    acc2[f1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[3], bh[s], acc2[f1], 0, 0, 0);
    acc2[f0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bl[s], acc2[f0], 0, 0, 0);
    acc2[f1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2], bl[s], acc2[f1], 0, 0, 0);
    acc2[f0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bh[s], acc2[f0], 0, 0, 0);
    acc2[f1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[2], bh[s], acc2[f1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Z: output op z of a pass = half a chunk of per-tap A tiles.  The activation + split of hidden block fo happens once per
// fused group of row tiles; a row tile is stored when its last hidden block has been added.
template <int KIN, int MOUT, int NF, int MODE, int NP, int P0, int PASS, bool SOLO, int Z>
__device__ __forceinline__ void h3_Z(const NetArgs& a, const H3Ctx& c, const float* epl, f32x16 (&acc2)[NF / NP],
                                     f32x16 (&acc3)[(RingH<KIN, MOUT, NF, MODE, NP>::G0N)], h8 (&bh)[2], h8 (&bl)[2], int g, int q, bool qok,
                                     int lane, int hh) {
  using G = RingH<KIN, MOUT, NF, MODE, NP>;
  constexpr int NFH = G::NFH, M3 = G::M3;
  constexpr int S = Z >> 1;
  constexpr int P0N = (G::NMT + 1 + P0) & 1;                       // main slot of the next pass's chunk 0
  if (!g) {   // group 0 requests the later output chunks (chunk 2 as soon as the last main chunk is dead, chunk s >= 3 when chunk s-2 is) ...
    constexpr int SN = Z == 1 ? 2 : (Z >= 5 && (Z & 1)) ? (Z + 1) / 2 : 0;
    if constexpr (SN >= 2 && SN < G::NMT)
      stage4<G::MAINP, 16 + Z>(G::out_chunk(c.img, PASS, SN), ((NF + SN - 1 + P0) & 1) ? c.sB : c.sA, c.w4, c.voff);
    if constexpr (PASS + 1 < NP && !SOLO && Z == 2 * G::NMT - 1) {  // ... and, in its last op, the next pass's first chunks
      stage4<G::MAINP, 48>(G::main_chunk(c.img, PASS + 1, 0), P0N ? c.sB : c.sA, c.w4, c.voff);
      if constexpr (G::NMT >= 2) stage4<G::MAINP, 49>(G::out_chunk(c.img, PASS + 1, 0), c.sD, c.w4, c.voff);   // (NMT = 1: slot D still read; see h3_pass)
    }
  }
  const float4* slot = S == 0 ? c.sD : (((NF + S - 1 + P0) & 1) ? c.sB : c.sA);
  const h8* buf = reinterpret_cast<const h8*>(slot) + lane;
  float* Pp = a.P + (size_t)PASS * a.pstride;
  const float* pb = epl + NF * 32;
#pragma unroll
  for (int i = 0; i < NFH / 2; ++i) {
    const int tp = (Z & 1) * (NFH / 2) + i;            // tile position in the chunk
    const int t = S * NFH + tp;
    const int fo = G::tile_fo(t), mt = G::tile_mt(t);
    const bool first_group = t < NFH * G::G0N;
    const int ml = first_group ? mt : mt - G::G0N;     // this row tile's accumulator within its group
    if (ml == 0) {
      unsigned mask = 0;
      if ((MODE & 7) == NET_BWD) mask = c.mkl[((size_t)(threadIdx.x >> 6) * NF + PASS * NFH + fo) * 64 + lane];   // mask1: the ReLU after conv1
      const unsigned bits = h3_act<MODE>(acc2[fo], a.sc2, mask, bh, bl, (MODE & 8) && qok,
                                         (MODE & 8) ? uniform_fptr(a.st2 + (size_t)(PASS * NFH + fo) * 32 * a.Q) : nullptr,
                                         ((unsigned)(4 * hh) * (unsigned)a.Q + (unsigned)q) * 4u, (unsigned)a.Q * 4u, c.ub[0]);
      if ((MODE & 7) == NET_FWD_SAVE && first_group && c.wok) a.mask2[(c.wblk * NF + PASS * NFH + fo) * 64 + lane] = (unsigned short)bits;
    }
    if (fo == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[ml][r] = 0.0f;
    }
    acc3[ml] = mfma3(buf[(tp * 4 + 0) * 64], buf[(tp * 4 + 1) * 64], bh[0], bl[0], acc3[ml]);
    acc3[ml] = mfma3(buf[(tp * 4 + 2) * 64], buf[(tp * 4 + 3) * 64], bh[1], bl[1], acc3[ml]);
    if (fo == NFH - 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * 32 + mfma_row(r, hh);
        if (m < M3 && qok) Pp[(size_t)m * a.Q + q] = (MODE & 7) == NET_BWD ? acc3[ml][r] * (a.sc3 * c.ub[0])
                                                     : PASS == 0 ? fmaf(acc3[ml][r], a.sc3, pb[m]) : acc3[ml][r] * a.sc3;
      }
    }
  }
  h3_wait_barrier();
}

template <int KIN, int MOUT, int NF, int MODE, int NP, int P0, int PASS, bool SOLO, int... Z>
__device__ __forceinline__ void h3_tail(const NetArgs& a, const H3Ctx& c, const float* epl, f32x16 (&acc2)[NF / NP],
                                        f32x16 (&acc3)[(RingH<KIN, MOUT, NF, MODE, NP>::G0N)], h8 (&bh)[2], h8 (&bl)[2], int g, int q, bool qok,
                                        int lane, int hh, std::integer_sequence<int, Z...>) {
  (h3_Z<KIN, MOUT, NF, MODE, NP, P0, PASS, SOLO, Z>(a, c, epl, acc2, acc3, bh, bl, g, q, qok, lane, hh), ...);
}

// one pass (hidden half PASS) of the workgroup's 256 pixels.  g = this wave's group: its ops run in phase (op index + g).
// SOLO: the workgroup runs this pass only (the other half belongs to another workgroup).
template <int KIN, int MOUT, int NF, int MODE, int NP, int P0, int PASS, bool SOLO>
__device__ __forceinline__ void h3_pass(const NetArgs& a, const H3Ctx& c, const float* epl,
                                        const h8 (&xh)[(RingH<KIN, MOUT, NF, MODE, NP>::KS)], const h8 (&xl)[(RingH<KIN, MOUT, NF, MODE, NP>::KS)],
                                        int g, int q, bool qok, int lane, int hh) {
  using G = RingH<KIN, MOUT, NF, MODE, NP>;
  constexpr int NFH = G::NFH, NMT = G::NMT;
  constexpr int f2base = PASS * NFH * 32;
  f32x16 acc2[NFH];
#pragma unroll
  for (int fo = 0; fo < NFH; ++fo)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[fo][r] = (MODE & 7) == NET_BWD ? 0.0f : epl[f2base + fo * 32 + mfma_row(r, hh)];   // conv2 bias (scaled)

  h8 bh[2], bl[2];
#pragma nounroll
  for (int i0 = 0; i0 < NF; i0 += 2) {
    // X_i0 | Y_i0 | X_i0+1 | Y_i0+1.  During its Y ops group 0 requests the next main chunk (after the last one the second
    // output chunk) into the other main slot, group 1 the small-conv operands of block i+2 (wrapping into the next pass)
    // into the slot its X_i has just finished with.
    h3_X<KIN, MOUT, NF, MODE, NP, 0, PASS>(a, c, i0, xh, xl, lane, bh, bl);
    h3_x_end<MODE, PASS>(c);
    if (PASS >= 1 && !SOLO && NMT == 1 && i0 == 0 && !g)   // single output chunk: slot D of the previous pass is read until the phase before this one
      stage4<G::MAINP, 50>(G::out_chunk(c.img, PASS, 0), c.sD, c.w4, c.voff);
    h3_Y<KIN, MOUT, NF, MODE, NP, 1>(P0 ? c.sB : c.sA, bh, bl, acc2, lane, g, true, G::main_chunk(c.img, PASS, i0 + 1), P0 ? c.sA : c.sB,
                                     c.k1img + (size_t)((i0 + 2) % NF) * G::K14, c.k1s0, c.w4, c.voff);
    h3_barrier();
    h3_X<KIN, MOUT, NF, MODE, NP, 1, PASS>(a, c, i0 + 1, xh, xl, lane, bh, bl);
    h3_x_end<MODE, PASS>(c);
    h3_Y<KIN, MOUT, NF, MODE, NP, 2>(P0 ? c.sA : c.sB, bh, bl, acc2, lane, g, i0 + 2 < NF || NMT >= 2,
                                     i0 + 2 < NF ? G::main_chunk(c.img, PASS, i0 + 2) : G::out_chunk(c.img, PASS, 1), P0 ? c.sB : c.sA,
                                     c.k1img + (size_t)((i0 + 3) % NF) * G::K14, c.k1s1, c.w4, c.voff);
    h3_barrier();
  }

  // the 3x3 output convolution as per-tap 1x1 partial sums over this pass's hidden half (the consumer adds the two passes)
  f32x16 acc3[G::G0N];
  h3_tail<KIN, MOUT, NF, MODE, NP, P0, PASS, SOLO>(a, c, epl, acc2, acc3, bh, bl, g, q, qok, lane, hh, std::make_integer_sequence<int, 2 * NMT>());
}

// all passes of a workgroup, one after the other (the slot parity of a pass's first chunk follows from the previous pass)
template <int KIN, int MOUT, int NF, int MODE, int NP, int PASS, int P0>
__device__ __forceinline__ void h3_passes(const NetArgs& a, const H3Ctx& c, const float* epl,
                                          const h8 (&xh)[(RingH<KIN, MOUT, NF, MODE, NP>::KS)], const h8 (&xl)[(RingH<KIN, MOUT, NF, MODE, NP>::KS)],
                                          int g, int q, bool qok, int lane, int hh) {
  h3_pass<KIN, MOUT, NF, MODE, NP, P0, PASS, false>(a, c, epl, xh, xl, g, q, qok, lane, hh);
  if constexpr (PASS + 1 < NP)
    h3_passes<KIN, MOUT, NF, MODE, NP, PASS + 1, (RingH<KIN, MOUT, NF, MODE, NP>::NMT + 1 + P0) & 1>(a, c, epl, xh, xl, g, q, qok, lane, hh);
}

// SPLIT: grid.y = NP and each workgroup runs ONE pass (small grids: 1/NP of the latency per launch on NP times the workgroups)
template <int KIN, int MOUT, int NF, int MODE, int NP, bool SPLIT>
__global__ __launch_bounds__(512, 2) void k_net_h3(NetArgs a) {
  using G = RingH<KIN, MOUT, NF, MODE, NP>;
  constexpr int K1 = G::K1;
  constexpr int KS = G::KS;
  constexpr int SGN = ((MODE & 7) == NET_BWD) ? -1 : 1;   // backward gathers at q - d(tap)
  static_assert(G::FITS, "shape");

  __shared__ float4 slotA[G::MAIN4];
  __shared__ float4 slotB[G::MAIN4];
  __shared__ float4 slotD[G::MAIN4];
  __shared__ float4 k1slot0[G::K14];
  __shared__ float4 k1slot1[G::K14];
  __shared__ float epl[G::EPN];
  __shared__ __attribute__((aligned(16))) unsigned short mkl[G::MASK2B / 2 + 8];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7; waves w and w+4 share a SIMD
  const int g = wave >> 2;
  const int pix = lane & 31;
  const int hh = lane >> 5;
  const int q = (blockIdx.x * 8 + (tid >> 6)) * 32 + pix;
  const bool qok = q < a.Q;

  H3Ctx c;
  c.sA = slotA; c.sB = slotB; c.sD = slotD; c.k1s0 = k1slot0; c.k1s1 = k1slot1;
  c.k1img = a.RHp;
  const int solo_pass = SPLIT ? (int)blockIdx.y : 0;
  c.img = a.RHp;
  c.mkl = mkl;
  c.wblk = (size_t)blockIdx.x * 8 + wave;
  c.wok = (long)c.wblk * 32 < a.Q;
  c.w4 = wave & 3;
  c.voff = (unsigned)lane * 16u;
  c.ub[0] = c.ub[1] = 1.0f;

  // first the DMA of everything the first phases need (it has the longest latency of the prologue), then the gathers
  if (!g) {
    stage4<G::MAINP, 60>(G::main_chunk(c.img, solo_pass, 0), slotA, c.w4, c.voff);       // main chunk 0
    stage4<G::MAINP, 61>(G::out_chunk(c.img, solo_pass, 0), slotD, c.w4, c.voff);        // first output chunk
    if ((MODE & 7) == NET_BWD)   // the forward pass's ReLU decisions of this workgroup's 8 column blocks: [mask1 | mask2][wave][block][lane]
      stage4<NF, 64>(reinterpret_cast<const float4*>(a.mask1 + (size_t)blockIdx.x * 8 * NF * 64), reinterpret_cast<float4*>(mkl), c.w4, c.voff);
  } else {
    stage4<G::K1P, 62>(c.k1img, k1slot0, c.w4, c.voff);                                   // small-conv operands of blocks 0, 1
    stage4<G::K1P, 63>(c.k1img + G::K14, k1slot1, c.w4, c.voff);
    if ((MODE & 7) == NET_BWD)
      stage4<NF, 65>(reinterpret_cast<const float4*>(a.mask2 + (size_t)blockIdx.x * 8 * NF * 64), reinterpret_cast<float4*>(mkl + 8 * NF * 64), c.w4, c.voff);
  }
  // im2col fragments of this lane's pixel: k-step s holds k = 16 s + 8 hh + j (natural order), scaled and split
  h8 xh[KS], xl[KS];
  float xmax = 0.0f;                 // range guard: largest |network input| (scaled) this lane gathers
  {
    const int hw = a.h * a.w;
    const int qq = qok ? q : 0;
    const int rem = qq % hw;
    const int i = rem / a.w, j0 = rem % a.w;
    const float* base = a.vin + (long)qq * a.in_stride + a.in_off;
    if constexpr ((MODE & 7) == NET_BWD) {
      // the backward network is linear: normalise the pixel's gradient vector (its two lanes hh = 0, 1 hold it) by a power of two
      float v[KS][8];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        gather8<KIN, false, SGN>(base, i, j0, a.h, a.w, a.in_stride, qok, 16 * s + 8 * hh, v[s]);
        xmax = range8(xmax, v[s]);
      }
      xmax = nan_max(xmax, __shfl_xor(xmax, 32, 64));
      // (training, MODE & 8: the hiddens this launch stores feed GEMMs over ALL pixels, so one scale has to serve the whole
      //  launch -- the producer pre-scaled g_o by a host-chosen power of two, BwdArgs::go_scale, and the static bound xlim checks it)
      const float fac = (MODE & 8) ? 1.0f : pixel_norm(xmax, a.bnorm, c.ub[0]);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[s][j] *= fac;
        split8(v[s], xh[s], xl[s]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        float v[8];
        gather8<KIN, true, SGN>(base, i, j0, a.h, a.w, a.in_stride, qok, 16 * s + 8 * hh, v);
        xmax = range8(xmax, v);
        split8(v, xh[s], xl[s]);
      }
    }
  }
  if ((MODE & 7) != NET_BWD)
    for (int i = tid; i < G::EPN; i += 512) epl[i] = a.eph[i];
  // forward: beyond this input magnitude the host cannot rule out an fp16 overflow; backward (normalised per pixel): only a
  // non-finite gradient can
  if ((((MODE & 7) == NET_BWD && !(MODE & 8)) ? !(xmax <= 3.0e38f) : !(xmax <= a.xlim)) && a.flag) *a.flag = 1;
  if (a.xmax_out) range_probe(a.xmax_out, xmax);           // (diagnostic runs only: how far below the limit do the inputs stay?)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                         // constants and the first chunks visible to every wave

  if (g) h3_barrier();                                     // group 1 runs one phase behind group 0
  if constexpr (SPLIT) {
    if (solo_pass == 0) h3_pass<KIN, MOUT, NF, MODE, NP, 0, 0, true>(a, c, epl, xh, xl, g, q, qok, lane, hh);
    else if (solo_pass == 1) h3_pass<KIN, MOUT, NF, MODE, NP, 0, 1, true>(a, c, epl, xh, xl, g, q, qok, lane, hh);
    else if constexpr (NP > 2) {
      if (solo_pass == 2) h3_pass<KIN, MOUT, NF, MODE, NP, 0, 2, true>(a, c, epl, xh, xl, g, q, qok, lane, hh);
      else h3_pass<KIN, MOUT, NF, MODE, NP, 0, 3, true>(a, c, epl, xh, xl, g, q, qok, lane, hh);
    }
  } else {
    h3_passes<KIN, MOUT, NF, MODE, NP, 0, 0>(a, c, epl, xh, xl, g, q, qok, lane, hh);
  }
  if (!g) h3_barrier();                                    // group 0 idles through the last phase
}

// ------------------------------------------------------------------------------------------------------------------
// k_net_h3s: the forward coupling network of k_net_h3 on v_mfma_f32_16x16x32_f16.  Under this chip's power management the
// 16x16x32 shape sustains ~10 % more FLOP/s than 32x32x16 (scripts/mfma_shape.hip: 1 697 vs 1 535 TFLOP/s, LDS-fed, random
// data), and k_net_h3 is bound by exactly that.  Same workgroup geometry, op sequence, LDS slots and DMA
// schedule as k_net_h3 (chunks have the same size); what changes is the tiling inside an op:
//   * a wave's 32 pixels are two 16-pixel halves; lane l = (n = l & 15, kq = l >> 4) holds pixel n of each half;
//   * accumulators are 16 x 16 tiles (4 registers): rows 4 kq + r of a 16-row block;
//   * a hidden block of 32 channels = two row blocks = 8 values per lane and half = exactly one B fragment of the next
//     contraction (one k-step of 32): k slot (kq, j) <-> channel 16 (j >> 2) + 4 kq + (j & 3); the host packs A that way.
// Forward only (NET_FWD): the gradient path keeps the 32x32 kernels, whose ReLU-mask layout the fp32 fallback shares.
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FUSE_PSTR = 260;     // fused coupling: floats per LDS row of P: 4 * 260 = 16 (mod 64) banks apart, so the four row
                                   // quarters (kq) of a 16 x 16 accumulator tile store to disjoint banks
constexpr int FUSE_EW = 64;        // pixels per row slot of the edge buffer

// KIN / MOUT / MODE / NP as for RingH; MODE | 16: the coupling fused into the kernel (fused_couple)
template <int KIN, int MOUT, int NF, int MODE, int NP>
struct RingS {
  static constexpr bool BWD = (MODE & 7) == NET_BWD;
  static constexpr bool FUSE = (MODE & 16) != 0;                   // coupling fused into the kernel (see fused_couple); plain forward modes only
  static constexpr int PXH = (MODE & 32) ? 1 : 2;                  // 16-pixel halves per wave.  MODE | 32: ONE half -- a workgroup of 128 pixels with
                                                                   // half the MFMAs and half the epilogue work per phase: for grids that leave CUs
                                                                   // idle (latency per launch is what counts there) and for shapes whose small-conv
                                                                   // fragments (KS k-steps x PXH halves x 2 registers x 4) need the room
  static constexpr int NFH = NF / NP;                             // hidden 32-channel blocks per pass
  static constexpr int NRB = 2 * NFH;                             // 16-row blocks per pass
  static constexpr int K1 = 9 * KIN;
  static constexpr int KS = (K1 + (BWD ? 0 : 1) + 31) / 32;       // small-conv k-steps of 32 (forward: a spare row carries the bias)
  static constexpr int MAINP = NRB * 2;                           // pieces per main chunk: NRB row blocks x (hi, lo)
  static constexpr int MAIN4 = MAINP * 64;
  static constexpr int K1P = KS * 4;                              // one block's conv1 operands: KS x 2 row blocks x (hi, lo)
  static constexpr int K14 = K1P * 64;
  static constexpr int M3 = MOUT;
  static constexpr int NMT = (M3 + 15) / 16;                      // 16-row blocks of P
  static constexpr int TPC = MAINP / 2;                           // conv3 tiles (16 rows x 32 k, hi + lo) per chunk
  static constexpr int NT = NFH * NMT;                            // conv3 tiles per pass
  static constexpr int NCH = (NT + TPC - 1) / TPC;                // conv3 chunks per pass (the last one may be part empty)
  static constexpr int G0N = NMT < 6 ? NMT : 6;                   // row blocks are processed in fused groups of <= 6 (the last may be short)
  static constexpr int NGRP = (NMT + 5) / 6;
  static constexpr int GT = NFH * 6;                              // tiles of a full group
  static constexpr int grp_n(int gi) { return NMT - 6 * gi < 6 ? NMT - 6 * gi : 6; }
  static constexpr int EPN = (NF * 32 + 16 * NMT + 3) & ~3;       // conv2 accumulator init (F) | per-row constants of P
  static constexpr int MASK2B = BWD ? 2 * NF * 512 * 2 : 0;       // backward: both ReLU masks of the workgroup's 8 column blocks
  static constexpr size_t LDS_BYTES = (size_t)3 * MAIN4 * 16 + (size_t)2 * K14 * 16 + (size_t)EPN * 4 + MASK2B;
  static constexpr bool FITS = LDS_BYTES <= 160 * 1024 && NF % 4 == 0 && NFH >= 2 && KS <= (PXH == 1 ? (NP == 2 ? 5 : 9) : NP == 2 ? 3 : 5) && NGRP <= 3 &&
                               (NGRP == 1 || GT % TPC == 0) &&   // (chunks do not straddle groups)
                               (!FUSE || (MOUT == 36 && NP == 2 && PXH == 2 && LDS_BYTES + (size_t)36 * FUSE_PSTR * 4 + 4096 + 128 <= 160 * 1024));
  // image (one for every NP, laid out for NP = 2): conv1 operands of all blocks, then per half of the hidden width the main
  // chunks (NF row blocks) and the conv3 tiles of NF/2 hidden blocks, 16 tiles per chunk; a pass of NP = 4 reads its half
  static constexpr int K1TOT4 = NF * K14;
  static constexpr int IMGH = NF / 2;
  static constexpr int IMG_MAIN4 = IMGH * 256;
  static constexpr int IMG_NCH = (IMGH * NMT + 2 * IMGH - 1) / (2 * IMGH);
  static constexpr int IMG_PASS4 = (NF + IMG_NCH) * IMG_MAIN4;
  static constexpr int SUBS = NP / 2;
  __device__ static const float4* main_chunk(const float4* img, int pass, int ch) {
#ifdef GLOWK_EXP_SAMECHUNK   // (diagnostic build, wrong results: every weight DMA reads the same 32 KiB -- is the weight stream's L2 / HBM latency on the critical path?)
    ch = 0;
#endif
    return img + K1TOT4 + (size_t)(pass / SUBS) * IMG_PASS4 + (size_t)ch * IMG_MAIN4 + (size_t)(pass % SUBS) * MAIN4;
  }
  __device__ static const float4* out_chunk(const float4* img, int pass, int s) {   // conv3 chunk s of a pass (TPC tiles)
    const int sub = pass % SUBS, t4 = s * TPC;
    const int gi = t4 / GT;                                        // image order: group by group, within a group sub-pass by sub-pass
    const int t2 = IMGH * 6 * gi + sub * NFH * grp_n(gi) + (t4 - GT * gi);
    return img + K1TOT4 + (size_t)(pass / SUBS) * IMG_PASS4 + (size_t)NF * IMG_MAIN4 + (size_t)t2 * 128;
  }
  static constexpr int tile_fo(int t) { return (t - GT * (t / GT)) / grp_n(t / GT); }
  static constexpr int tile_mt(int t) { return 6 * (t / GT) + (t - GT * (t / GT)) % grp_n(t / GT); }
  // P written ONCE: with two passes in one workgroup and at most three row blocks of P (c = 4: the level that is 3/4 of the
  // time) pass 0 keeps its partial sums in 24 registers and pass 1 adds them before the only store.  (Diagnostic builds: the P
  // stores are ~10 % of the kernel's time -- skipping two thirds of them made a pass 7 % faster -- so half of them is worth having;
  // the consumer reads one partial buffer instead of two.)  Plain forward modes only: the saving pass keeps its two buffers.
#ifdef GLOWK_NO_MERGE   // (A/B builds only, scripts/ab.py)
  static constexpr bool MERGE = false;
#else
  static constexpr bool MERGE = PXH == 2 && NP == 2 && NGRP == 1 && NMT <= 3 &&
                                ((MODE & 7) == NET_FWD || (MODE & 7) == NET_FWD2 || ((MODE & 7) == NET_BWD && NMT <= 2) || ((MODE & 7) == NET_FWD_SAVE && FUSE));   // (the saving pass keeps no P
                                // any more -- the coupling's pre-tanh inputs are what the backward pass reads -- so it can take the fused form too;
                                // the 4-channel level's backward network (18 output rows: 16 registers) merges its two passes as well, the 8-channel
                                // one (36 rows: 24 registers) spills 18 registers if it does)
#endif
};

template <bool TWO = false>
__device__ __forceinline__ f32x4 mfma3s(const h8& ahi, const h8& alo, const h8& bhi, const h8& blo, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bhi, acc, 0, 0, 0);
  if (!TWO) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, blo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bhi, acc, 0, 0, 0);
  return acc;
}

// activation + split of one hidden block (two row blocks) of one pixel half -> the B fragment of the next contraction.
// Forward: ReLU (returns the 8 decisions as bits j = 4 * row block + r); backward: pass where the forward ReLU was open.
// (ST: training -- the 8 values, in the scaled units they are split in, also go to a planar [F][Q] array: rows 4 kq + r of the two
//  16-row blocks of the hidden block whose base is st_blk; st_lane = byte offset of (row 4 kq, this lane's pixel), st_row = bytes per row)
// (the values: v = what the split B fragments are made of; returns the ReLU decisions of a saving launch)
template <int MODE>
__device__ __forceinline__ unsigned h3s_act_vals(const f32x4& r0, const f32x4& r1, float sc, unsigned mask8, float (&v)[8]) {
  unsigned bits = 0;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const f32x2 a2 = j < 4 ? f32x2{r0[j], r0[j + 1]} : f32x2{r1[j - 4], r1[j - 3]};
    const f32x2 t2 = a2 * f32x2{sc, sc};                               // v_pk_mul_f32
    const float t[2] = {t2.x, t2.y}, av[2] = {a2.x, a2.y};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (MODE == NET_BWD) v[j + e] = __int_as_float(__float_as_int(t[e]) & __builtin_amdgcn_sbfe((int)mask8, j + e, 1));   // bit ? t : 0
      else {
        v[j + e] = fmaxf(t[e], 0.0f);
        if (MODE == NET_FWD_SAVE) bits |= (av[e] > 0.0f ? 1u : 0u) << (j + e);
      }
    }
  }
  return bits;
}
template <int MODE, bool ST = false>
__device__ __forceinline__ unsigned h3s_act(const f32x4& r0, const f32x4& r1, float sc, unsigned mask8, h8& bh, h8& bl, bool do_st = false,
                                            float* st_blk = nullptr, unsigned st_lane = 0, unsigned st_row = 0) {
  float v[8];
  const unsigned bits = h3s_act_vals<MODE>(r0, r1, sc, mask8, v);
  if constexpr (ST) {
#ifdef GLOWK_EXP_NOHST      // (diagnostic build, wrong weight gradients: the hidden tensors are not stored -- what do the stores cost a training launch?)
    do_st = false;
#endif
    if (do_st) {
      const unsigned long long st_base = reinterpret_cast<unsigned long long>(st_blk);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned off = st_lane + (unsigned)((j >> 2) * 16 + (j & 3)) * st_row;
        asm volatile("global_store_dword %0, %1, %2" GLOWK_ST_MOD ::"v"(off), "v"(v[j]), "s"(st_base) : "memory");
      }
    }
  }
  if (MODE == NET_FWD2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { bh[j] = (_Float16)v[j]; bl[j] = (_Float16)0.0f; }
  } else split8(v, bh, bl);
  return bits;
}
// ... of BOTH pixel halves of a wave whose lane n holds the ADJACENT pixels 2 n, 2 n + 1 (k_net_h3c's training form): the stored values
// leave as 8-byte pairs, 16 lanes x 8 B = one full 128-byte line per hidden row and instruction (st_lane: byte offset of (row 4 kq, pixel
// 2 n)); mask16 / result: bits 8 hf + ..., as the callers of h3s_act assemble them
template <int MODE, bool ST>
__device__ __forceinline__ unsigned h3s_act_pair(const f32x4& r00, const f32x4& r10, const f32x4& r01, const f32x4& r11, float sc, unsigned mask16,
                                                 h8 (&bh)[2], h8 (&bl)[2], float* st_blk, unsigned st_lane, unsigned st_row) {
  float v0[8], v1[8];
  unsigned bits = h3s_act_vals<MODE>(r00, r10, sc, mask16, v0);
  bits |= h3s_act_vals<MODE>(r01, r11, sc, mask16 >> 8, v1) << 8;
  // (The CU's vector-store path takes ~24 B/clk into L2 and a single wave ~4.5 B/clk, whatever the width per lane: 21 clk per CU / ~110 clk
  //  per wave and 8-byte wave store -- scripts/store_rate_probe.hip.  Issuing the stores two at a time between the conversions below
  //  changed nothing, docs/EXPERIMENTS.md round 4.)
  if constexpr (ST) {
#ifndef GLOWK_EXP_NOHST
    const unsigned long long st_base = reinterpret_cast<unsigned long long>(st_blk);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned off = st_lane + (unsigned)((j >> 2) * 16 + (j & 3)) * st_row;
      const f32x2 pr = {v0[j], v1[j]};
      asm volatile("global_store_dwordx2 %0, %1, %2" GLOWK_ST_MOD ::"v"(off), "v"(pr), "s"(st_base) : "memory");
    }
#endif
  }
  split8(v0, bh[0], bl[0]);
  split8(v1, bh[1], bl[1]);
  return bits;
}

// (the ReLU masks of this kernel family: per lane and hidden block 16 bits, bit 8 * pixel half + 4 * row block + r)
// (PAIR: lane n of a wave holds the adjacent pixels 2 n, 2 n + 1 instead of n, 16 + n, and a training launch stores pairs)
template <int KIN, int MOUT, int NF, int MODE, int NP, int KP, int PASS, bool PAIR = false>
__device__ __forceinline__ void h3s_X(const NetArgs& a, const H3Ctx& c, int fi, const h8 (&xh)[(RingS<KIN, MOUT, NF, MODE, NP>::KS)][2],
                                      const h8 (&xl)[(RingS<KIN, MOUT, NF, MODE, NP>::KS)][2], int lane, h8 (&bh)[2], h8 (&bl)[2]) {
  using G = RingS<KIN, MOUT, NF, MODE, NP>;
  unsigned mask = 0;
  if ((MODE & 7) == NET_BWD) mask = c.mkl[c.mk2off + ((size_t)(threadIdx.x >> 6) * NF + fi) * 64 + lane];   // mask2: the ReLU after conv2
  f32x4 h1[2][2];   // [row block][pixel half]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) h1[i >> 1][i & 1][r] = 0.0f;
  const h8* k1 = reinterpret_cast<const h8*>(KP ? c.k1s1 : c.k1s0) + lane;   // [s][row block][hi|lo][64]
  if constexpr (G::KS <= 3) {       // all operand reads in flight before the first MFMA
    h8 kf[G::K1P];
#pragma unroll
    for (int i = 0; i < G::K1P; ++i) kf[i] = k1[i * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < G::KS; ++s)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int hf = 0; hf < G::PXH; ++hf)
          h1[rb][hf] = mfma3s<(MODE & 7) == NET_FWD2>(kf[(s * 2 + rb) * 2 + 0], kf[(s * 2 + rb) * 2 + 1], xh[s][hf], xl[s][hf], h1[rb][hf]);
  } else {
#pragma unroll
    for (int s = 0; s < G::KS; ++s)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const h8 ah = k1[((s * 2 + rb) * 2 + 0) * 64], al = k1[((s * 2 + rb) * 2 + 1) * 64];
#pragma unroll
        for (int hf = 0; hf < G::PXH; ++hf) h1[rb][hf] = mfma3s<(MODE & 7) == NET_FWD2>(ah, al, xh[s][hf], xl[s][hf], h1[rb][hf]);
      }
  }
  unsigned bits = 0;
  if constexpr (PAIR) {      // (every wave full: the host's condition for the form)
    static_assert(G::PXH == 2, "pairs: two pixel halves per wave");
    const int stq = (int)c.wblk * 32 + 2 * (lane & 15);
    if constexpr ((MODE & 8) && PASS == 0)
#ifdef GLOWK_EXP_TILEDST   // (diagnostic builds, wrong weight gradients: the stores of a workgroup land in ONE contiguous [F][128] block -- do the 1-MB row
                           //  strides cost?  GLOWK_EXP_TILEDST = 0: every hidden block of a workgroup lands on the SAME 16 KB -- does the path behind L2?)
      bits = h3s_act_pair<(MODE & 7), true>(h1[0][0], h1[1][0], h1[0][1], h1[1][1], a.sc1, mask, bh, bl,
                                            uniform_fptr(a.st1 + ((size_t)blockIdx.x * NF * 32 + (size_t)fi * 32 * GLOWK_EXP_TILEDST) * 128),
                                            ((unsigned)(4 * (lane >> 4)) * 128u + (unsigned)(stq & 127)) * 4u, 128u * 4u);
#else
      bits = h3s_act_pair<(MODE & 7), true>(h1[0][0], h1[1][0], h1[0][1], h1[1][1], a.sc1, mask, bh, bl, uniform_fptr(a.st1 + (size_t)fi * 32 * a.Q),
                                            ((unsigned)(4 * (lane >> 4)) * (unsigned)a.Q + (unsigned)stq) * 4u, (unsigned)a.Q * 4u);
#endif
    else bits = h3s_act_pair<(MODE & 7), false>(h1[0][0], h1[1][0], h1[0][1], h1[1][1], a.sc1, mask, bh, bl, nullptr, 0u, 0u);
  } else {
#pragma unroll
    for (int hf = 0; hf < G::PXH; ++hf) {
      const int stq = (int)c.wblk * (16 * G::PXH) + 16 * hf + (lane & 15);
      bits |= h3s_act<(MODE & 7), (MODE & 8) != 0>(h1[0][hf], h1[1][hf], a.sc1, mask >> (8 * hf), bh[hf], bl[hf], (MODE & 8) && PASS == 0 && stq < a.Q,
                                                   (MODE & 8) ? uniform_fptr(a.st1 + (size_t)fi * 32 * a.Q) : nullptr,
                                                   ((unsigned)(4 * (lane >> 4)) * (unsigned)a.Q + (unsigned)stq) * 4u, (unsigned)a.Q * 4u) << (8 * hf);
    }
  }
  if ((MODE & 7) == NET_FWD_SAVE && PASS == 0 && c.wok) a.mask1[(c.wblk * NF + fi) * 64 + lane] = (unsigned short)bits;
}

// Y: conv2 contribution of one hidden block (one k-step of 32) to the pass's NRB x 2 accumulator tiles; same pipelining and
// DMA duties as h3_Y (groups of 12 MFMAs = two row blocks x two pixel halves x three split terms)
template <int KIN, int MOUT, int NF, int MODE, int NP, int TAG>
__device__ __forceinline__ void h3s_Y(const float4* slot, const h8 (&bh)[2], const h8 (&bl)[2], f32x4 (&acc2)[(RingS<KIN, MOUT, NF, MODE, NP>::NRB)][2],
                                      int lane, int g, bool main_ok, const float4* main_src, float4* main_dst, const float4* k1_src,
                                      float4* k1_dst, int w4, unsigned voff) {
  using G = RingS<KIN, MOUT, NF, MODE, NP>;
  const h8* buf = reinterpret_cast<const h8*>(slot) + lane;
  const char* mb = uniform_ptr(main_src);
  constexpr int NG = G::NRB / 2;
  constexpr int PPG = G::MAINP / 4 / NG;            // DMA pieces per wave and group
#ifdef GLOWK_EXP_YPRIO   // (A/B build: the wave in its matrix phase outranks its SIMD partner, which is in the VALU-heavy X phase)
  __builtin_amdgcn_s_setprio(1);
#endif
  h8 A[2][4];
  auto load = [&](h8 (&d)[4], int gi) {
    d[0] = buf[((2 * gi) * 2 + 0) * 64];            // row block 2gi hi, lo; row block 2gi+1 hi, lo
    d[1] = buf[((2 * gi) * 2 + 1) * 64];
    d[2] = buf[((2 * gi + 1) * 2 + 0) * 64];
    d[3] = buf[((2 * gi + 1) * 2 + 1) * 64];
  };
  load(A[0], 0);
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int o0 = 2 * gi, o1 = 2 * gi + 1;
#ifdef GLOWK_EXP_SAMEFRAG   // (diagnostic build, wrong results: every group reuses the first group's A fragments -- what do Y's LDS reads cost?)
    if (gi == 0) load(A[1], 1);
#else
    if (gi + 1 < NG) load(A[(gi + 1) & 1], gi + 1);
#endif
    if (!g) {
      if (main_ok) {
#pragma unroll
        for (int e = 0; e < PPG; ++e) {
          const int piece = (gi * PPG + e) * 4 + w4;
          glds16(reinterpret_cast<const float4*>(mb + (size_t)piece * 1024 + voff), main_dst + piece * 64);
        }
      }
      asm volatile("; dma site %0" ::"n"(TAG * 16 + gi));
    } else if (gi == 0) {
      stage4<G::K1P, TAG * 16 + 15>(k1_src, k1_dst, w4, voff);
    }
    __builtin_amdgcn_sched_barrier(0);
    const h8 (&af)[4] = A[gi & 1];
    acc2[o0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], bh[0], acc2[o0][0], 0, 0, 0);
    if constexpr (G::PXH == 2) acc2[o0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], bh[1], acc2[o0][1], 0, 0, 0);
    acc2[o1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[3], bh[0], acc2[o1][0], 0, 0, 0);
    if constexpr (G::PXH == 2) acc2[o1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[3], bh[1], acc2[o1][1], 0, 0, 0);
    if ((MODE & 7) != NET_FWD2) {
      acc2[o0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bl[0], acc2[o0][0], 0, 0, 0);
      if constexpr (G::PXH == 2) acc2[o0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bl[1], acc2[o0][1], 0, 0, 0);
      acc2[o1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2], bl[0], acc2[o1][0], 0, 0, 0);
      if constexpr (G::PXH == 2) acc2[o1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2], bl[1], acc2[o1][1], 0, 0, 0);
    }
    acc2[o0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bh[0], acc2[o0][0], 0, 0, 0);
    if constexpr (G::PXH == 2) acc2[o0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bh[1], acc2[o0][1], 0, 0, 0);
    acc2[o1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2], bh[0], acc2[o1][0], 0, 0, 0);
    if constexpr (G::PXH == 2) acc2[o1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[2], bh[1], acc2[o1][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#ifdef GLOWK_EXP_YPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// Z: conv3 op z of a pass = half a chunk of A tiles (16 rows x one hidden block)
template <int KIN, int MOUT, int NF, int MODE, int NP, int P0, int PASS, bool SOLO, int Z>
__device__ __forceinline__ void h3s_Z(const NetArgs& a, const H3Ctx& c, const float* epl, f32x4 (&acc2)[(RingS<KIN, MOUT, NF, MODE, NP>::NRB)][2],
                                      f32x4 (&acc3)[(RingS<KIN, MOUT, NF, MODE, NP>::G0N)][2], h8 (&bh)[2], h8 (&bl)[2], int g, const int (&q)[2],
                                      const bool (&qok)[2], int lane, int kq, f32x4 (&keep)[(RingS<KIN, MOUT, NF, MODE, NP>::G0N)][2]) {
  using G = RingS<KIN, MOUT, NF, MODE, NP>;
  constexpr int NFH = G::NFH, M3 = G::M3, TPC = G::TPC;
  constexpr bool MERGE = G::MERGE && !SOLO;
  constexpr int S = Z >> 1;
  constexpr int P0N = (G::NCH + 1 + P0) & 1;
  if (!g) {
    constexpr int SN = Z == 1 ? 2 : (Z >= 5 && (Z & 1)) ? (Z + 1) / 2 : 0;
    if constexpr (SN >= 2 && SN < G::NCH)
      stage4<G::MAINP, 16 + Z>(G::out_chunk(c.img, PASS, SN), ((NF + SN - 1 + P0) & 1) ? c.sB : c.sA, c.w4, c.voff);
    if constexpr (PASS + 1 < NP && !SOLO && Z == 2 * G::NCH - 1) {
      stage4<G::MAINP, 48>(G::main_chunk(c.img, PASS + 1, 0), P0N ? c.sB : c.sA, c.w4, c.voff);
      if constexpr (G::NCH >= 2) stage4<G::MAINP, 49>(G::out_chunk(c.img, PASS + 1, 0), c.sD, c.w4, c.voff);   // (NCH = 1: slot D still read; see h3s_pass)
    }
  }
  const float4* slot = S == 0 ? c.sD : (((NF + S - 1 + P0) & 1) ? c.sB : c.sA);
  const h8* buf = reinterpret_cast<const h8*>(slot) + lane;
  float* Pp = a.P + (MERGE ? (size_t)0 : (size_t)PASS * a.pstride);
  const float* pb = epl + NF * 32;
#pragma unroll
  for (int i = 0; i < TPC / 2; ++i) {
    const int tp = (Z & 1) * (TPC / 2) + i;            // tile position in the chunk
    const int t = S * TPC + tp;
    if (t < G::NT) {
      const int fo = G::tile_fo(t), mt = G::tile_mt(t);
      const int ml = mt % 6;                             // this row block's accumulator within its group
      if (ml == 0) {
        unsigned mask = 0, bits = 0;
        if ((MODE & 7) == NET_BWD) mask = c.mkl[((size_t)(threadIdx.x >> 6) * NF + PASS * NFH + fo) * 64 + lane];   // mask1: the ReLU after conv1
#pragma unroll
        for (int hf = 0; hf < G::PXH; ++hf)
          bits |= h3s_act<(MODE & 7), (MODE & 8) != 0>(acc2[2 * fo][hf], acc2[2 * fo + 1][hf], a.sc2, mask >> (8 * hf), bh[hf], bl[hf], (MODE & 8) && qok[hf],
                                                       (MODE & 8) ? uniform_fptr(a.st2 + (size_t)(PASS * NFH + fo) * 32 * a.Q) : nullptr,
                                                       ((unsigned)(4 * kq) * (unsigned)a.Q + (unsigned)q[hf]) * 4u, (unsigned)a.Q * 4u) << (8 * hf);
        if ((MODE & 7) == NET_FWD_SAVE && t < NFH * G::G0N && c.wok) a.mask2[(c.wblk * NF + PASS * NFH + fo) * 64 + lane] = (unsigned short)bits;
      }
      if (fo == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc3[ml][0][r] = 0.0f; acc3[ml][1][r] = 0.0f; }
      }
      const h8 ah = buf[(tp * 2 + 0) * 64], al = buf[(tp * 2 + 1) * 64];
      acc3[ml][0] = mfma3s<(MODE & 7) == NET_FWD2>(ah, al, bh[0], bl[0], acc3[ml][0]);
      if constexpr (G::PXH == 2) acc3[ml][1] = mfma3s<(MODE & 7) == NET_FWD2>(ah, al, bh[1], bl[1], acc3[ml][1]);
      if (fo == NFH - 1) {
#pragma unroll
        for (int hf = 0; hf < G::PXH; ++hf)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mt * 16 + 4 * kq + r;
            const float val = (MODE & 7) == NET_BWD ? acc3[ml][hf][r] * (a.sc3 * c.ub[hf])
                              : PASS == 0 ? fmaf(acc3[ml][hf][r], a.sc3, pb[m]) : acc3[ml][hf][r] * a.sc3;
            if (MERGE && PASS == 0) { keep[ml][hf][r] = val; continue; }     // pass 1 adds it and stores once
            if constexpr (G::FUSE) {      // the per-tap outputs stay in the workgroup: LDS row m, pixel = wave * 32 + 16 hf + lane % 16
              if (m < M3) c.pl[m * FUSE_PSTR + (int)(threadIdx.x >> 6) * 32 + 16 * hf + (lane & 15)] = val + keep[ml][hf][r];
              continue;
            }
#ifdef GLOWK_EXP_NOSTORE   // (diagnostic build, wrong results: only one row tile of P is written -- what do the P stores cost?)
            if (mt == 0)
#endif
            if (m < M3 && qok[hf]) Pp[(size_t)m * a.Q + q[hf]] = MERGE ? val + keep[ml][hf][r] : val;
          }
      }
    }
  }
  h3_wait_barrier();
}

template <int KIN, int MOUT, int NF, int MODE, int NP, int P0, int PASS, bool SOLO, int... Z>
__device__ __forceinline__ void h3s_tail(const NetArgs& a, const H3Ctx& c, const float* epl, f32x4 (&acc2)[(RingS<KIN, MOUT, NF, MODE, NP>::NRB)][2],
                                         f32x4 (&acc3)[(RingS<KIN, MOUT, NF, MODE, NP>::G0N)][2], h8 (&bh)[2], h8 (&bl)[2], int g, const int (&q)[2],
                                         const bool (&qok)[2], int lane, int kq, f32x4 (&keep)[(RingS<KIN, MOUT, NF, MODE, NP>::G0N)][2],
                                         std::integer_sequence<int, Z...>) {
  (h3s_Z<KIN, MOUT, NF, MODE, NP, P0, PASS, SOLO, Z>(a, c, epl, acc2, acc3, bh, bl, g, q, qok, lane, kq, keep), ...);
}

template <int KIN, int MOUT, int NF, int MODE, int NP, int P0, int PASS, bool SOLO>
__device__ __forceinline__ void h3s_pass(const NetArgs& a, const H3Ctx& c, const float* epl, const h8 (&xh)[(RingS<KIN, MOUT, NF, MODE, NP>::KS)][2],
                                         const h8 (&xl)[(RingS<KIN, MOUT, NF, MODE, NP>::KS)][2], int g, const int (&q)[2], const bool (&qok)[2], int lane, int kq,
                                         f32x4 (&keep)[(RingS<KIN, MOUT, NF, MODE, NP>::G0N)][2]) {
  using G = RingS<KIN, MOUT, NF, MODE, NP>;
  constexpr int NRB = G::NRB;
  constexpr int f2base = PASS * G::NFH * 32;
  f32x4 acc2[NRB][2];
#pragma unroll
  for (int ob = 0; ob < NRB; ++ob)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float b = (MODE & 7) == NET_BWD ? 0.0f : epl[f2base + ob * 16 + 4 * kq + r];   // conv2 bias (scaled)
      acc2[ob][0][r] = b;
      acc2[ob][1][r] = b;
    }
  h8 bh[2], bl[2];
#pragma nounroll
  for (int i0 = 0; i0 < NF; i0 += 2) {
    h3s_X<KIN, MOUT, NF, MODE, NP, 0, PASS>(a, c, i0, xh, xl, lane, bh, bl);
    h3_x_end<MODE, PASS, 8 * G::PXH>(c);
    if (PASS >= 1 && !SOLO && G::NCH == 1 && i0 == 0 && !g)   // single output chunk: slot D of the previous pass is read until the phase before this one
      stage4<G::MAINP, 50>(G::out_chunk(c.img, PASS, 0), c.sD, c.w4, c.voff);
    h3s_Y<KIN, MOUT, NF, MODE, NP, 1>(P0 ? c.sB : c.sA, bh, bl, acc2, lane, g, true, G::main_chunk(c.img, PASS, i0 + 1), P0 ? c.sA : c.sB,
                         c.k1img + (size_t)((i0 + 2) % NF) * G::K14, c.k1s0, c.w4, c.voff);
    h3_barrier();
    h3s_X<KIN, MOUT, NF, MODE, NP, 1, PASS>(a, c, i0 + 1, xh, xl, lane, bh, bl);
    h3_x_end<MODE, PASS, 8 * G::PXH>(c);
    h3s_Y<KIN, MOUT, NF, MODE, NP, 2>(P0 ? c.sA : c.sB, bh, bl, acc2, lane, g, i0 + 2 < NF || G::NCH >= 2,
                         i0 + 2 < NF ? G::main_chunk(c.img, PASS, i0 + 2) : G::out_chunk(c.img, PASS, 1),
                         P0 ? c.sB : c.sA, c.k1img + (size_t)((i0 + 3) % NF) * G::K14, c.k1s1, c.w4, c.voff);
    h3_barrier();
  }
  f32x4 acc3[G::G0N][2];
  h3s_tail<KIN, MOUT, NF, MODE, NP, P0, PASS, SOLO>(a, c, epl, acc2, acc3, bh, bl, g, q, qok, lane, kq, keep, std::make_integer_sequence<int, 2 * G::NCH>());
}

template <int KIN, int MOUT, int NF, int MODE, int NP, int PASS, int P0>
__device__ __forceinline__ void h3s_passes(const NetArgs& a, const H3Ctx& c, const float* epl, const h8 (&xh)[(RingS<KIN, MOUT, NF, MODE, NP>::KS)][2],
                                           const h8 (&xl)[(RingS<KIN, MOUT, NF, MODE, NP>::KS)][2], int g, const int (&q)[2], const bool (&qok)[2], int lane, int kq,
                                           f32x4 (&keep)[(RingS<KIN, MOUT, NF, MODE, NP>::G0N)][2]) {
  h3s_pass<KIN, MOUT, NF, MODE, NP, P0, PASS, false>(a, c, epl, xh, xl, g, q, qok, lane, kq, keep);
  if constexpr (PASS + 1 < NP) h3s_passes<KIN, MOUT, NF, MODE, NP, PASS + 1, (RingS<KIN, MOUT, NF, MODE, NP>::NCH + 1 + P0) & 1>(a, c, epl, xh, xl, g, q, qok, lane, kq, keep);
}

// ------------------------------------------------------------------------------------------------------------------
// Coupling fused into the network kernel (4-channel level, both passes in one workgroup: RingS::MERGE).  The workgroup's per-tap
// conv3 outputs P [36][256 px] never go to HBM: the closing Z ops write them to LDS, and after the last phase every pixel whose
// 3 x 3 neighbourhood lies inside the workgroup (or outside the image: zero padding) is finished here -- gather of the nine taps,
// conv3 bias, tanh / exp, the affine coupling (flow_tfp_bijectors.py:134-148), the next step's ActNorm + 1x1, the store of the
// NHWC output, and its share of sum log_s (:150-153).  A workgroup holds 256 / w whole image rows; the pixels of its first / last
// row that have a neighbour row in ANOTHER workgroup cannot be finished: for them the kernel leaves (a) their partial sums over the
// taps it holds and (b) what its own first / last row contributes to the rows above / below (8 floats per edge pixel in all, instead
// of 36 floats of P per pixel), and k_couple_edge -- one small workgroup per sample -- completes those rows and adds the log-det.
// Two threads per pixel: lane half k = lane >> 5 owns channel pair (k, 2 + k) = (pre-tanh log_s_k, t_k).
// Requires: w a power of two <= 64, h w a multiple of 256 or a power of two in [32, 256) (the host checks: glowk.hip).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool fz_not_finite(float v) { return !(fabsf(v) <= 3.0e38f); }

// PXW = pixels per workgroup (256: k_net_h3s; 128: the co-resident form k_net_h3c, glowk_co.h), PSTR = floats per LDS row of P
template <int PXW = 256, int PSTR = FUSE_PSTR>
__device__ __forceinline__ void fused_couple(const NetArgs& a, const float* pl, const float4* vst, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  const int px = wave * 32 + (lane & 31), k = lane >> 5;
  const int q = (int)blockIdx.x * PXW + px;
  const bool qok = q < a.Q;
  const int w = a.w, hw = a.h * a.w, nrows = PXW / w;
  const int pp = (qok ? q : 0) % hw;
  const int i = pp / w, j = pp % w, r = px / w;
  float ols = 0.0f, ot = 0.0f;
  bool missing = false;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const int ii = i + dy, jj = j + dx;
    if (ii < 0 || ii >= a.h || jj < 0 || jj >= w) continue;      // zero padding
    const int rr = r + dy;
    if (rr < 0 || rr >= nrows) { missing = true; continue; }      // that row belongs to another workgroup
    const int spx = px + dy * w + dx;
    ols += pl[(tap * 4 + k) * PSTR + spx];
    ot += pl[(tap * 4 + 2 + k) * PSTR + spx];
  }
  ols += a.fz_b3[k];
  ot += a.fz_b3[2 + k];
  float lsum = 0.0f;
  // (the partner lane l ^ 32 holds the other channel pair of the SAME pixel: the two take identical branches)
#ifdef GLOWK_EXP_VGLOBAL   // (diagnostic build: the tail reads the coupling input from global memory, not from the LDS stash)
  const float4 v4 = qok ? *reinterpret_cast<const float4*>(a.vin + (size_t)q * 4) : float4{0.f, 0.f, 0.f, 0.f};
#else
  const float4 v4 = vst[px];          // the pixel's four input channels, parked in LDS by the prologue (a global load here would be
                                      // a full memory latency on the workgroup's serial tail: nothing else runs on this CU)
#endif
  const float vk = k ? v4.y : v4.x;
  const float log_s = tanhf(ols);
  const float sc = expf(log_s);
  const float yk = a.fz_inverse ? (vk - ot) / sc : sc * vk + ot;
  const float yo = __shfl_xor(yk, 32, 64);                        // the other transformed channel
  const float y[4] = {k ? yo : yk, k ? yk : yo, v4.z, v4.w};
  if (qok && a.fz_osave) a.fz_osave[(size_t)q * 2 + k] = ols;     // (edge pixels: k_couple_edge overwrites with the complete sums)
  if (qok && !missing) {
    lsum = log_s;
    bool bad = fz_not_finite(ols) | fz_not_finite(ot) | fz_not_finite(yk);
    if (bad && a.flag) *a.flag = 1;
    // this lane stores output channels 2k, 2k + 1: the same fma chain over the input channels as k_couple's affine_cc
    float z0, z1;
    if (a.fz_A) {
      z0 = a.fz_b[2 * k]; z1 = a.fz_b[2 * k + 1];
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        z0 = fmaf(y[ci], a.fz_A[ci * 4 + 2 * k], z0);
        z1 = fmaf(y[ci], a.fz_A[ci * 4 + 2 * k + 1], z1);
      }
    } else {
      z0 = y[2 * k]; z1 = y[2 * k + 1];
    }
    *reinterpret_cast<float2*>(a.fz_out + (size_t)q * a.fz_out_stride + a.fz_out_off + 2 * k) = float2{z0, z1};
  } else if (qok) {
    // edge pixel: partial sums (bias included) of the taps this workgroup holds; slot 0 = its first row, 1 = its last row
    float* e = a.fz_edge + (((size_t)blockIdx.x * 4 + (r == 0 ? 0 : 1)) * FUSE_EW + j) * 4;
    e[k] = ols;
    e[2 + k] = ot;
  }
  // what the first row contributes to the row above (slot 2: taps dy = +1 seen from there) and the last row to the row below
  // (slot 3: taps dy = -1): 2 w targets x 4 channels, one thread each
  for (int te = tid; te < 2 * w * 4; te += 2 * PXW) {      // (2 PXW threads per workgroup; w <= 64: at most two rounds)
    const int which = te / (4 * w), jt = (te / 4) % w, ch = te & 3;
    float hsum = 0.0f;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int js = jt + dx;
      if (js < 0 || js >= w) continue;
      const int tap = (which ? 0 : 6) + dx + 1;
      hsum += pl[(tap * 4 + ch) * PSTR + (which ? (nrows - 1) * w : 0) + js];
    }
    a.fz_edge[(((size_t)blockIdx.x * 4 + 2 + which) * FUSE_EW + jt) * 4 + ch] = hsum;
  }
  // log-det share of the pixels finished here: one fp64 partial per wave (32 pixels: h w >= 32, so a wave never straddles two
  // samples), lanes added in a fixed order; k_couple_edge adds a sample's partials in order (no barrier on this serial tail)
  if (a.fz_ldpart) {
    double d = (double)lsum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
    if (lane == 0) a.fz_ldpart[(size_t)blockIdx.x * (PXW / 32) + wave] = d;
  }
}

template <int KIN, int MOUT, int NF, int MODE, int NP, bool SPLIT>
__global__ __launch_bounds__(512, 2) void k_net_h3s(NetArgs a) {
  using G = RingS<KIN, MOUT, NF, MODE, NP>;
  constexpr int KS = G::KS;
  constexpr int SGN = ((MODE & 7) == NET_BWD) ? -1 : 1;   // backward gathers at q - d(tap)
  static_assert(G::FITS, "shape");

  __shared__ float4 slotA[G::MAIN4];
  __shared__ float4 slotB[G::MAIN4];
  __shared__ float4 slotD[G::MAIN4];
  __shared__ float4 k1slot0[G::K14];
  __shared__ float4 k1slot1[G::K14];
  __shared__ float epl[G::EPN];
  __shared__ __attribute__((aligned(16))) unsigned short mkl[G::MASK2B / 2 + 8];
  __shared__ float plds[G::FUSE ? 36 * FUSE_PSTR : 1];   // fused coupling: the workgroup's per-tap outputs
  __shared__ float4 vstash[G::FUSE ? 256 : 1];           // ... and its pixels' four input channels (coupling input)
  static_assert(!G::FUSE || (G::MERGE && !SPLIT && MOUT == 36), "the fused coupling needs both passes in one workgroup and a 4-channel level");
  static_assert(!(MODE & 8) || G::PXH == 1, "hidden stores (training): half-wave form only -- with two pixel halves a wave whose second half lies "
                                            "beyond Q would issue fewer stores than h3_x_end's counted wait assumes");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2;
  const int n16 = lane & 15;
  const int kq = lane >> 4;
  const int qbase = (blockIdx.x * 8 + (tid >> 6)) * (16 * G::PXH);
  const int q[2] = {qbase + n16, G::PXH == 2 ? qbase + 16 + n16 : qbase + n16};
  const bool qok[2] = {q[0] < a.Q, G::PXH == 2 && q[1] < a.Q};

  H3Ctx c;
  c.sA = slotA; c.sB = slotB; c.sD = slotD; c.k1s0 = k1slot0; c.k1s1 = k1slot1;
  const int solo_pass = SPLIT ? (int)blockIdx.y : 0;
  c.k1img = a.RSp;
  c.img = a.RSp;
  c.mkl = mkl;
  c.mk2off = 8 * NF * 64;
  c.pl = plds;
  c.wblk = (size_t)blockIdx.x * 8 + wave;      // this wave's pixel block (16 PXH pixels): the unit of the ReLU-mask arrays
  c.wok = (long)c.wblk * (16 * G::PXH) < a.Q;
  c.w4 = wave & 3;
  c.voff = (unsigned)lane * 16u;
  c.ub[0] = c.ub[1] = 1.0f;

  if (!g) {
    stage4<G::MAINP, 60>(G::main_chunk(c.img, solo_pass, 0), slotA, c.w4, c.voff);
    stage4<G::MAINP, 61>(G::out_chunk(c.img, solo_pass, 0), slotD, c.w4, c.voff);
    if ((MODE & 7) == NET_BWD)   // the forward pass's ReLU decisions of this workgroup's 8 column blocks: [mask1 | mask2][wave][block][lane]
      stage4<NF, 64>(reinterpret_cast<const float4*>(a.mask1 + (size_t)blockIdx.x * 8 * NF * 64), reinterpret_cast<float4*>(mkl), c.w4, c.voff);
  } else {
    stage4<G::K1P, 62>(c.k1img, k1slot0, c.w4, c.voff);
    stage4<G::K1P, 63>(c.k1img + G::K14, k1slot1, c.w4, c.voff);
    if ((MODE & 7) == NET_BWD)
      stage4<NF, 65>(reinterpret_cast<const float4*>(a.mask2 + (size_t)blockIdx.x * 8 * NF * 64), reinterpret_cast<float4*>(mkl + 8 * NF * 64), c.w4, c.voff);
  }
  // im2col fragments of this lane's two pixels: k-step s holds k = 32 s + 8 kq + j (natural order), scaled and split
  h8 xh[KS][2], xl[KS][2];
  float xmax = 0.0f;                 // range guard: largest |network input| (scaled) this lane gathers
  {
    const int hw = a.h * a.w;
#pragma unroll
    for (int hf = 0; hf < G::PXH; ++hf) {
      const int qq = qok[hf] ? q[hf] : 0;
      const int rem = qq % hw;
      const int i = rem / a.w, j0 = rem % a.w;
      const float* base = a.vin + (long)qq * a.in_stride + a.in_off;
      if constexpr ((MODE & 7) == NET_BWD) {
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
        const float fac = (MODE & 8) ? 1.0f : pixel_norm(pm, a.bnorm, c.ub[hf]);     // (training: one scale per launch, BwdArgs::go_scale)
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
    if (tid < 256) {
      const int qv = (int)blockIdx.x * 256 + tid;
      vstash[tid] = qv < a.Q ? *reinterpret_cast<const float4*>(a.vin + (size_t)qv * 4) : float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if ((MODE & 7) != NET_BWD)
    for (int i = tid; i < G::EPN; i += 512) epl[i] = a.eph[i];   // RingS::EPN <= RingH::EPN, same content
  // forward: the static bound; backward (normalised per pixel): only a non-finite gradient can leave the range
  if ((((MODE & 7) == NET_BWD && !(MODE & 8)) ? !(xmax <= 3.0e38f) : !(xmax <= a.xlim)) && a.flag) *a.flag = 1;
  if (a.xmax_out) range_probe(a.xmax_out, xmax);           // (diagnostic runs only: how far below the limit do the inputs stay?)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 keep[G::G0N][2];     // pass 0's partial sums of P (RingS::MERGE; otherwise never touched and compiled away)
#ifdef GLOWK_EXP_STATICPRIO   // (A/B build: the younger half of the workgroup -- group 1 -- outranks its SIMD partners for good)
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
#endif
  if (g) h3_barrier();
  if constexpr (SPLIT) {
    if (solo_pass == 0) h3s_pass<KIN, MOUT, NF, MODE, NP, 0, 0, true>(a, c, epl, xh, xl, g, q, qok, lane, kq, keep);
    else if (solo_pass == 1) h3s_pass<KIN, MOUT, NF, MODE, NP, 0, 1, true>(a, c, epl, xh, xl, g, q, qok, lane, kq, keep);
    else if constexpr (NP > 2) {
      if (solo_pass == 2) h3s_pass<KIN, MOUT, NF, MODE, NP, 0, 2, true>(a, c, epl, xh, xl, g, q, qok, lane, kq, keep);
      else h3s_pass<KIN, MOUT, NF, MODE, NP, 0, 3, true>(a, c, epl, xh, xl, g, q, qok, lane, kq, keep);
    }
  } else {
    h3s_passes<KIN, MOUT, NF, MODE, NP, 0, 0>(a, c, epl, xh, xl, g, q, qok, lane, kq, keep);
  }
  if (!g) h3_barrier();
  if constexpr (G::FUSE) {
    __syncthreads();       // every wave's LDS writes of P are complete and visible (lgkmcnt(0) + barrier)
    fused_couple(a, plds, vstash, tid);
  }
}

