// glowk device code, part 4: the training step's own kernels (train_glow.py:29-44: loss = sum(-log_prob) / global batch,
// gradients wrt flow.trainable_variables, optimizer.apply_gradients).
//
// The data-gradient sweep of glowk_log_prob_grad already walks the steps in reverse; the training sweep adds:
//   1. the saving forward launch also stores  R1 = relu(conv1 + b1), R2 = relu(conv2 + b2)      planar [F][Q]
//      (k_net_f32<.., STORE> / k_net_h3<.., MODE | 8>; kept for all steps, or recomputed per step when memory is short)
//   2. the backward launch also stores        M2 = mask2 . conv3^T(g_o), M1 = mask1 . (K2 g_a2)   planar [F][Q]
//   and then, for a whole level at a time (every kernel below takes a batch index = step; step by step when memory is short):
//   3. k_im2col_planar                 Xcol[(tap, ci)][q] = v_b[q + d(tap)][ci] (+ a row of ones), Gcol[(tap, co)][q] = g_o[q - d(tap)][co]
//   4. three GEMMs C = A . B^T over the pixel dimension, deterministic (fixed-order partial sums, no atomics):
//                                      C3 = [R2; 1] . Gcol^T   C2 = [R1; 1] . M2^T   C1 = M1 . [Xcol; 1]^T
//      k_wgrad_nt (exact: fp32 MFMA) or k_wgrad_h3 (sweeps in the split arithmetic: three-product fp16 split, fp32 accumulate)
//   5. k_assemble_*                    BatchNorm gamma / beta, biases, and the per-channel factors that turn C1..C3 into dK1..dK3
// Planar [channel][pixel] is the layout in which a channel row is K-contiguous for those GEMMs and in which the MFMA
// accumulator tiles of k_net (row = channel in a register, column = pixel on the lane) store as 128-byte segments.
// ActNorm / 1x1 gradients: k_affine_wgrad reduces dA = sum_q u^T g_v, db = sum_q g_v per step; the chain rule through
// A = diag(e^ls) P L U, b = sh W is c x c algebra done by the host in fp64 (glowk.hip).  Optimizer: one elementwise kernel.
#pragma once
#include "glowk_kernels.h"

// ---- im2col in planar form -------------------------------------------------------------------------------------------
// out[(tap * CH + ch)][q] = in[q + sgn * d(tap)][in_off + ch] inside the image, 0 outside; row 9 * CH (if ones) = 1
// blockIdx.y = batch entry (a step of the level): in / out advance by in_bs / out_bs floats (in_bs may be negative)
__global__ __launch_bounds__(256) void k_im2col_planar(const float* __restrict__ in, int in_stride, int in_off, int CH, int Q, int h, int w,
                                                      int sgn, int ones, float* __restrict__ out, ptrdiff_t in_bs, ptrdiff_t out_bs) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= Q) return;
  in += (ptrdiff_t)blockIdx.y * in_bs;
  out += (ptrdiff_t)blockIdx.y * out_bs;
  const int hw = h * w, rem = q % hw, i = rem / w, j = rem % w;
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = sgn * (tap / 3 - 1), dx = sgn * (tap % 3 - 1);
    const int ii = i + dy, jj = j + dx;
    const bool ok = ii >= 0 && ii < h && jj >= 0 && jj < w;
    const float* src = in + (size_t)(q + dy * w + dx) * in_stride + in_off;
    for (int ch = 0; ch < CH; ++ch) out[(size_t)(tap * CH + ch) * Q + q] = ok ? src[ch] : 0.0f;
  }
  if (ones) out[(size_t)(9 * CH) * Q + q] = 1.0f;
}

// workgroup barrier of the GEMM main loops: this wave's LDS writes are complete (lgkmcnt(0)), but its global loads stay in flight --
// __syncthreads() makes hipcc drain vmcnt(0) as well, which exposes the full HBM latency of the prefetch in every round (measured:
// 4.4 us per 32-deep round of k_wgrad_h3 against 0.3 us of MFMA work)
__device__ __forceinline__ void gemm_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---- weight-gradient GEMM:  Cpart[s][m][n] = sum_{k in slice s} A[m][k] * B[n][k] ---------------------------------------
// A: [M][ldk], B: [N][ldk] row-major, K = the pixel dimension (contiguous).  a_ones: row M of A is an implicit row of ones
// (so that C[M][n] = sum_k B[n][k]: the bias / BatchNorm-offset sums come out of the same GEMM).
// One workgroup = a 64 x 64 tile of C over one K slice: 4 waves, each a 32 x 32 tile on v_mfma_f32_32x32x2_f32 (exact fp32
// products: the gradients feed an optimizer, no reason to round them).  Tiles of 64 x 32 are staged k-major in LDS so that
// an MFMA operand is one conflict-free ds_read_b32 per lane; the next tile's global loads are issued before the MFMAs of
// the current one.  Split-K partials are written, not atomically added: the reduction order is fixed (bitwise repeatable).
struct WgradArgs {
  const float* A;
  const float* B;
  int M, N;          // rows of A / B actually stored
  int a_ones;        // 1: an extra row M of ones
  int K;             // pixels
  int kslice;        // pixels per split (multiple of 32)
  float* Cpart;      // [batch][S][Mp][Np], Mp = M + a_ones, Np = N; output z = b * S + s starts at Cpart + z * csz
  int S;             // slices per batch entry
  int tm, tn;        // tiles in each direction (1-D grid of tm * tn * batch * S workgroups, XCD-aware order: see k_wgrad_h3)
  ptrdiff_t bsA, bsB;   // floats between the batch entries of A / B (may be negative)
  size_t csz;        // floats between consecutive outputs (>= Mp * Np)
};

// logical workgroup id of a 1-D grid such that each of the 8 XCDs (hardware deals consecutive ids round-robin to them) owns a contiguous
// run of logical ids: tiles that share operand panels then share an L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_contiguous_id() {
  const int nwg = (int)gridDim.x, orig = (int)blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// WT = 32 x 32 accumulator tiles per wave in each direction: workgroup tile (64 WT) x (64 WT).  WT = 2 (128 x 128, 64 accumulator
// registers per lane, every LDS operand read feeds two MFMAs) for the square conv2 gradient, WT = 1 for the skinny ones.
template <int WT, bool VEC>
__global__ __launch_bounds__(256) void k_wgrad_nt(WgradArgs a) {
  constexpr int TS = 64 * WT;          // tile side
  constexpr int LD = TS + 2;           // k-major LDS rows, padded: the transposing writes of a wave hit 64 distinct banks
  constexpr int RPT = TS / 32;         // rows per thread and operand in one 32-deep stage
  __shared__ float As[2][32][LD];
  __shared__ float Bs[2][32][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wg = xcd_contiguous_id(), tiles = a.tm * a.tn, tile = wg % tiles, zz = wg / tiles;
  const int m0 = (tile % a.tm) * TS, n0 = (tile / a.tm) * TS, s = zz % a.S;
  const int Mp = a.M + a.a_ones;
  a.A += (ptrdiff_t)(zz / a.S) * a.bsA;
  a.B += (ptrdiff_t)(zz / a.S) * a.bsB;
  const long k_begin = (long)s * a.kslice;
  const long k_end = k_begin + a.kslice < a.K ? k_begin + a.kslice : a.K;
  f32x16 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  // loader: thread t brings rows (t >> 3) + 32 e of both tiles, 4 consecutive k each
  const int lr = tid >> 3, lk = (tid & 7) * 4;
  // two register sets: the loads of stage c + 2 are issued before the MFMAs of stage c, so a load has two stages to land
  float4 ra[2][RPT], rb[2][RPT];
  auto fetch = [&](long k0, int set) {
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
      const int m = m0 + lr + 32 * e, n = n0 + lr + 32 * e;
      const long k = k0 + lk;
      float4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if constexpr (VEC) {  // K a multiple of 4: rows are 16-byte aligned and a float4 never straddles the end of a slice.
        // Unconditional loads from clamped addresses, then selects: a branch around a load makes hipcc wait for each one in turn.
        const long kc = k < (long)a.K - 4 ? k : (long)a.K - 4;
        const float4 ta = *reinterpret_cast<const float4*>(a.A + (size_t)(m < a.M ? m : a.M - 1) * a.K + kc);
        const float4 tb = *reinterpret_cast<const float4*>(a.B + (size_t)(n < a.N ? n : a.N - 1) * a.K + kc);
        const bool in = k < k_end, oka = in && m < a.M, one = in && m == a.M && a.a_ones, okb = in && n < a.N;
        va.x = oka ? ta.x : one ? 1.f : 0.f; va.y = oka ? ta.y : one ? 1.f : 0.f; va.z = oka ? ta.z : one ? 1.f : 0.f; va.w = oka ? ta.w : one ? 1.f : 0.f;
        vb.x = okb ? tb.x : 0.f; vb.y = okb ? tb.y : 0.f; vb.z = okb ? tb.z : 0.f; vb.w = okb ? tb.w : 0.f;
      } else {              // odd pixel counts (1 x 1 or 3 x 1 images at the last level): element by element
        float ta[4], tb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool in = k + j < k_end;
          ta[j] = !in ? 0.0f : m < a.M ? a.A[(size_t)m * a.K + k + j] : (m == a.M && a.a_ones) ? 1.0f : 0.0f;
          tb[j] = (in && n < a.N) ? a.B[(size_t)n * a.K + k + j] : 0.0f;
        }
        va = float4{ta[0], ta[1], ta[2], ta[3]};
        vb = float4{tb[0], tb[1], tb[2], tb[3]};
      }
      ra[set][e] = va; rb[set][e] = vb;
    }
  };
  auto stage = [&](int buf, int set) {
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
      const int r = lr + 32 * e;
      As[buf][lk + 0][r] = ra[set][e].x; As[buf][lk + 1][r] = ra[set][e].y; As[buf][lk + 2][r] = ra[set][e].z; As[buf][lk + 3][r] = ra[set][e].w;
      Bs[buf][lk + 0][r] = rb[set][e].x; Bs[buf][lk + 1][r] = rb[set][e].y; Bs[buf][lk + 2][r] = rb[set][e].z; Bs[buf][lk + 3][r] = rb[set][e].w;
    }
  };
  const int i32 = lane & 31, kh = lane >> 5;
  auto compute = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      float av[WT], bv[WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) av[i] = As[buf][kk + kh][(wm * WT + i) * 32 + i32];
#pragma unroll
      for (int j = 0; j < WT; ++j) bv[j] = Bs[buf][kk + kh][(wn * WT + j) * 32 + i32];
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  };
  if (k_begin < k_end) {
    fetch(k_begin, 0);
    stage(0, 0);
    if (k_begin + 32 < k_end) fetch(k_begin + 32, 1);
    __syncthreads();
    // stages come in pairs so that the register set of every fetch / stage is a compile-time constant
    for (long k0 = k_begin; k0 < k_end; k0 += 64) {
      if (k0 + 64 < k_end) fetch(k0 + 64, 0);
      compute(0);
      if (k0 + 32 < k_end) stage(1, 1);
      gemm_barrier();
      if (k0 + 32 >= k_end) break;
      if (k0 + 96 < k_end) fetch(k0 + 96, 1);
      compute(1);
      if (k0 + 64 < k_end) stage(0, 0);
      gemm_barrier();
    }
  }
  float* C = a.Cpart + (size_t)zz * a.csz;
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * WT + i) * 32 + mfma_row(r, kh), n = n0 + (wn * WT + j) * 32 + i32;
        if (m < Mp && n < a.N) C[(size_t)m * a.N + n] = acc[i][j][r];
      }
}

// ---- the same GEMM in the split arithmetic (sweeps that ran the fp16-split kernels) ------------------------------------------
// Cpart[b][s][m][n] = 1 / (sa sb) * sum_{k in slice s} split(sa A_b[m][k]) . split(sb B_b[n][k]),  x = hi + lo in fp16, the three products
// hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation (fp32-class, section 5 of DESIGN.md; 8/3 of the fp32 MFMA rate).
// The operands are the fp32 planar arrays the split kernels stored -- the very values those kernels were about to split, in the
// units they split them in, so the range guard of the sweep covers them (sa = 1) -- and the planar im2col arrays (raw units:
// sb = GLOWK_ACT_SCALE, what the kernels' own gathers apply).  A workgroup converts every element ONCE while staging it (global ->
// registers -> hi / lo planes in LDS, double buffered; rows of 32 k padded to 80 bytes: a wave's 16-byte fragment reads are
// conflict free); a wave owns WTM x WTN accumulator tiles of 32 x 32.  Shapes: 256 x 128 (4 x 2 waves of 64 x 64; 128 x 128 = 2 x 2 waves where M is not a multiple of 256) for the square
// conv2 gradient, 128 x 64 (4 x 1 waves of 32 x 64) for the skinny conv1 / conv3 ones, which are bound by streaming (and converting) A.
// A launch covers `batch` independent GEMMs (the steps of a level) of S slices each.
struct WgradSplitArgs {
  const float* A;
  const float* B;
  int M, N, K;
  int kslice;        // pixels per slice (multiple of 32)
  int S;             // slices per batch entry
  int tm, tn;        // tiles in each direction
  ptrdiff_t bsA, bsB;   // floats between the batch entries of A / B (may be negative)
  float sa, sb;      // powers of two; sa must be 1
  float* Cpart;      // output z = b * S + s starts at Cpart + z * csz
  size_t csz;        // floats between consecutive outputs (>= M * N, or (M + 1) * N with b_sums)
  int plain;         // 1: the plain round order for every shape (A/B timing: GLOWK_WGRAD_PLAIN)
  int b_sums;        // 1: row M of every output = the row sums of B over the slice (what a row of ones appended to A would give; fp64
                     // accumulation in the workgroups of the first row tile, which stage those rows anyway)
};

// VEC: K is a multiple of 4 (float4 loads; a template parameter because a run-time branch around every load makes hipcc wait for
// each load in turn)
template <int WTM, int WTN, int WM, int WN, bool VEC>
__global__ __launch_bounds__(64 * WM * WN, WM * WN >= 8 ? 1 : 2) void k_wgrad_h3(WgradSplitArgs a) {
  static_assert(WM * WN == 4 || WM * WN == 8 || WM * WN == 16, "four, eight or sixteen waves");
  constexpr int RPP = 8 * WM * WN;     // rows one pass of the loader covers (8 threads per row of 32 k)
  constexpr int TM = 32 * WTM * WM, TN = 32 * WTN * WN;
  constexpr int LDH = 40;              // halves per LDS row: 32 k + 8 of padding
  constexpr int EA = TM / RPP, EB = TN / RPP;
  __shared__ __attribute__((aligned(16))) _Float16 Ah[2][TM * LDH];
  __shared__ __attribute__((aligned(16))) _Float16 Al[2][TM * LDH];
  __shared__ __attribute__((aligned(16))) _Float16 Bh[2][TN * LDH];
  __shared__ __attribute__((aligned(16))) _Float16 Bl[2][TN * LDH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // 1-D grid of tm * tn * batch * S workgroups.  Hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own
  // L2); the tiles of one (batch entry, slice) share their operand panels, so they are made to share an XCD: ids are remapped so that
  // every XCD owns a contiguous run of logical ids (bijective for any grid size), and a logical id is (slice-major) tile-minor.
  // Without this the 128 x 128 form is bound by HBM at its own arithmetic intensity (fp32 operands: 32 flop / byte = 126 TFLOP/s measured).
  const int wg = xcd_contiguous_id();
  const int tiles = a.tm * a.tn, tile = wg % tiles, zz = wg / tiles;
  const int m0 = (tile % a.tm) * TM, n0 = (tile / a.tm) * TN;
  const int b = zz / a.S, s = zz % a.S;
  const float* __restrict__ Ab = a.A + (ptrdiff_t)b * a.bsA;
  const float* __restrict__ Bb = a.B + (ptrdiff_t)b * a.bsB;
  const long k_begin = (long)s * a.kslice;
  const long k_end = k_begin + a.kslice < a.K ? k_begin + a.kslice : a.K;
  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int lr = tid >> 3, lk = (tid & 7) * 4;       // loader: rows lr + RPP e, 4 consecutive k
  float4 ra[EA], rb[EB];
  // full: the tile lies inside the matrices and the slice is whole 32-deep rounds -- no clamps, no selects (wave-uniform, decided once)
  const bool full = VEC && m0 + TM <= a.M && n0 + TN <= a.N && ((k_end - k_begin) & 31) == 0;
  auto load4 = [&](auto full_tag, const float* base, int row, int rows, long k) -> float4 {
    float4 v = {0.f, 0.f, 0.f, 0.f};
#ifdef WGRAD_EXP_NOLOAD  // (diagnostic build: no global loads)
    return v;
#endif
    if constexpr (decltype(full_tag)::value) {
      v = *reinterpret_cast<const float4*>(base + (size_t)row * a.K + k);
    } else if constexpr (VEC) {   // unconditional load from a clamped address, then select (no branch around the load)
      const int rc = row < rows ? row : rows - 1;
      const long kc = k < (long)a.K - 4 ? k : (long)a.K - 4;
      const float4 t = *reinterpret_cast<const float4*>(base + (size_t)rc * a.K + kc);
      const bool ok = row < rows && k < k_end;
      v.x = ok ? t.x : 0.0f; v.y = ok ? t.y : 0.0f; v.z = ok ? t.z : 0.0f; v.w = ok ? t.w : 0.0f;
    } else if (row < rows) {   // odd pixel counts (1 x 1 or 3 x 1 images at the last level): element by element
      const float* p = base + (size_t)row * a.K + k;
      if (k + 0 < k_end) v.x = p[0];
      if (k + 1 < k_end) v.y = p[1];
      if (k + 2 < k_end) v.z = p[2];
      if (k + 3 < k_end) v.w = p[3];
    }
    return v;
  };
  auto fetch = [&](auto full_tag, long k0) {
#pragma unroll
    for (int e = 0; e < EA; ++e) ra[e] = load4(full_tag, Ab, m0 + lr + RPP * e, a.M, k0 + lk);
#pragma unroll
    for (int e = 0; e < EB; ++e) rb[e] = load4(full_tag, Bb, n0 + lr + RPP * e, a.N, k0 + lk);
  };
  typedef _Float16 h4v __attribute__((ext_vector_type(4)));
  auto put = [&](auto unit_tag, const float4& v, float sc, _Float16* hi, _Float16* lo, int row) {
    constexpr bool UNIT = decltype(unit_tag)::value;    // scale 1: no multiply
#ifdef WGRAD_EXP_NOSTAGE // (diagnostic build: no LDS writes at all -- fragment reads, MFMAs and barriers alone)
    return;
#endif
#ifdef WGRAD_EXP_NOCVT   // (diagnostic build, wrong numbers on purpose: what does the conversion cost?  same bytes through the same path)
    *reinterpret_cast<float2*>(hi + row * LDH + lk) = make_float2(v.x, v.y);
    *reinterpret_cast<float2*>(lo + row * LDH + lk) = make_float2(v.z, v.w);
    return;
#endif
    const f32x2 p0 = {UNIT ? v.x : v.x * sc, UNIT ? v.y : v.y * sc}, p1 = {UNIT ? v.z : v.z * sc, UNIT ? v.w : v.w * sc};
    const h2v h0 = __builtin_convertvector(p0, h2v), h1 = __builtin_convertvector(p1, h2v);
    const f32x2 d0 = p0 - __builtin_convertvector(h0, f32x2), d1 = p1 - __builtin_convertvector(h1, f32x2);
    const h2v l0 = __builtin_convertvector(d0, h2v), l1 = __builtin_convertvector(d1, h2v);
    *reinterpret_cast<h4v*>(hi + row * LDH + lk) = h4v{h0[0], h0[1], h1[0], h1[1]};
    *reinterpret_cast<h4v*>(lo + row * LDH + lk) = h4v{l0[0], l0[1], l1[0], l1[1]};
  };
  double bsum[EB];
#pragma unroll
  for (int e = 0; e < EB; ++e) bsum[e] = 0.0;
  auto stage = [&](auto ub_tag, auto sum_tag, int buf) {        // (sa = 1 always: the A operands are the split kernels' own stores)
#pragma unroll
    for (int e = 0; e < EA; ++e) put(std::true_type{}, ra[e], 1.0f, Ah[buf], Al[buf], lr + RPP * e);
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      put(ub_tag, rb[e], a.sb, Bh[buf], Bl[buf], lr + RPP * e);
      if constexpr (decltype(sum_tag)::value) bsum[e] += ((double)rb[e].x + (double)rb[e].y) + ((double)rb[e].z + (double)rb[e].w);
    }
  };
  const int i32 = lane & 31, kh = lane >> 5;
  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h8 fah[WTM], fal[WTM], fbh[WTN], fbl[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i) {
        const int off = ((wm * WTM + i) * 32 + i32) * LDH + ks * 16 + 8 * kh;
        fah[i] = *reinterpret_cast<const h8*>(Ah[buf] + off);
        fal[i] = *reinterpret_cast<const h8*>(Al[buf] + off);
      }
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        const int off = ((wn * WTN + j) * 32 + i32) * LDH + ks * 16 + 8 * kh;
        fbh[j] = *reinterpret_cast<const h8*>(Bh[buf] + off);
        fbl[j] = *reinterpret_cast<const h8*>(Bl[buf] + off);
      }
      // product by product over the wave's tiles: consecutive MFMAs write different accumulators
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
    }
  };
  auto round_interleaved = [&](auto full_tag, auto ub_tag, auto sum_tag, int buf, long kf) {
    constexpr int NU = EA + EB;
    h8 fah[2][WTM], fal[2][WTM], fbh[2][WTN], fbl[2][WTN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < WTM; ++i) {
        const int off = ((wm * WTM + i) * 32 + i32) * LDH + ks * 16 + 8 * kh;
        fah[ks][i] = *reinterpret_cast<const h8*>(Ah[buf] + off);
        fal[ks][i] = *reinterpret_cast<const h8*>(Al[buf] + off);
      }
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        const int off = ((wn * WTN + j) * 32 + i32) * LDH + ks * 16 + 8 * kh;
        fbh[ks][j] = *reinterpret_cast<const h8*>(Bh[buf] + off);
        fbl[ks][j] = *reinterpret_cast<const h8*>(Bl[buf] + off);
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
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if ((u * 6) / NU != g) continue;
        if (u < EA) {
          const int e = u;
          put(std::true_type{}, ra[e], 1.0f, Ah[buf ^ 1], Al[buf ^ 1], lr + RPP * e);
          ra[e] = load4(full_tag, Ab, m0 + lr + RPP * e, a.M, kf + lk);
        } else {
          const int e = u - EA;
          put(ub_tag, rb[e], a.sb, Bh[buf ^ 1], Bl[buf ^ 1], lr + RPP * e);
          if constexpr (decltype(sum_tag)::value) bsum[e] += ((double)rb[e].x + (double)rb[e].y) + ((double)rb[e].z + (double)rb[e].w);
          rb[e] = load4(full_tag, Bb, n0 + lr + RPP * e, a.N, kf + lk);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto run = [&](auto full_tag, auto ub_tag, auto sum_tag) {
    fetch(full_tag, k_begin);
    stage(ub_tag, sum_tag, 0);
    if (k_begin + 32 < k_end) fetch(full_tag, k_begin + 32);
    __syncthreads();
    int buf = 0;
    constexpr bool PLAIN_ONLY = WTM * WTN != 4 || WM * WN == 16;      // (sixteen waves: 128 registers per wave -- one k-step's fragments at a time)
    if (PLAIN_ONLY || a.plain) {
      // the skinny shapes are bound by streaming A: the plain order (whole staging, all loads, then the MFMAs) keeps their loads earliest
      // (the branch-free, fenced form below: 29 -> 21 TFLOP/s on the conv1 / conv3 shapes)
      for (long k0 = k_begin; k0 < k_end; k0 += 32, buf ^= 1) {
        const bool more = k0 + 32 < k_end;
        if (more) stage(ub_tag, sum_tag, buf ^ 1);  // (buffer buf ^ 1 was last read before the barrier that ended the previous round)
        if (k0 + 64 < k_end) fetch(full_tag, k0 + 64);
        compute(buf);
        gemm_barrier();
      }
      return;
    }
    if constexpr (!PLAIN_ONLY) {
    long k0 = k_begin;
    // steady state: a round without a branch in its body, written as six groups -- one product of one k-step (WTM x WTN MFMAs) followed
    // by a share of the NEXT round's staging (conversion of one loader unit, its two LDS writes) and the global load that refills the
    // unit's registers for the round after -- fenced so that the compiler keeps the interleave: the matrix pipe works through a group's
    // MFMAs while the wave issues the group's ~20 VALU instructions.  (With the staging behind an `if (more)` it was a basic block of
    // its own: ~130 VALU instructions per wave with the matrix pipe idle, both waves of a SIMD in step because of the barrier.)
    for (; k0 + 64 < k_end; k0 += 32, buf ^= 1) {
      round_interleaved(full_tag, ub_tag, sum_tag, buf, k0 + 64);
      gemm_barrier();
    }
    if (k0 + 32 < k_end) {                        // the last round but one: nothing left to fetch
      stage(ub_tag, sum_tag, buf ^ 1);
      compute(buf);
      gemm_barrier();
      buf ^= 1;
    }
    compute(buf);
    }
  };
  const bool sums = a.b_sums && m0 == 0;
  if (k_begin < k_end) {
    if (sums) {
      if (full && a.sb == 1.0f) run(std::true_type{}, std::true_type{}, std::true_type{});
      else run(std::false_type{}, std::false_type{}, std::true_type{});
    } else {
      if (full && a.sb == 1.0f) run(std::true_type{}, std::true_type{}, std::false_type{});
      else if (full) run(std::true_type{}, std::false_type{}, std::false_type{});
      else run(std::false_type{}, std::false_type{}, std::false_type{});
    }
  }
  const float inv = 1.0f / a.sb;
  float* C = a.Cpart + (size_t)zz * a.csz;
  if (sums) {   // the 8 loader threads of a row hold its partial sums
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      double t = bsum[e];
      t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
      const int n = n0 + lr + RPP * e;
      if ((tid & 7) == 0 && n < a.N) C[(size_t)a.M * a.N + n] = (float)t;
    }
  }
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * WTM + i) * 32 + mfma_row(r, kh), n = n0 + (wn * WTN + j) * 32 + i32;
        if (m < a.M && n < a.N) C[(size_t)m * a.N + n] = acc[i][j][r] * inv;
      }
}

// out[n] = sum_k B[n][k]  (fp64 accumulation, one workgroup per row): the sums the appended row of ones used to deliver
// blockIdx.y = batch entry: B advances by b_bs floats, out by out_bs
__global__ __launch_bounds__(256) void k_rowsum(const float* __restrict__ B, int K, float* __restrict__ out, ptrdiff_t b_bs, size_t out_bs) {
  __shared__ double red[4];
  const float* b = B + (ptrdiff_t)blockIdx.y * b_bs + (size_t)blockIdx.x * K;
  out += (size_t)blockIdx.y * out_bs;
  double t = 0.0;
  for (int k = threadIdx.x; k < K; k += 256) t += (double)b[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// sum of the split-K partials: C[e] = sum_s Cpart[s][e]  (fixed order); blockIdx.y = batch entry: S * n partials each, out_bs floats apart
__global__ __launch_bounds__(256) void k_sum_parts(const float* __restrict__ part, int S, size_t n, float* __restrict__ out, size_t out_bs) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  part += (size_t)blockIdx.y * S * n;
  out += (size_t)blockIdx.y * out_bs;
  float t = 0.0f;
  for (int s = 0; s < S; ++s) t += part[(size_t)s * n + e];
  out[e] = t;
}

// the same for fp64 partials (k_affine_wgrad)
__global__ __launch_bounds__(256) void k_sum_parts_f64(const double* __restrict__ part, int S, size_t n, double* __restrict__ out, size_t out_bs) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  part += (size_t)blockIdx.y * S * n;
  out += (size_t)blockIdx.y * out_bs;
  double t = 0.0;
  for (int s = 0; s < S; ++s) t += part[(size_t)s * n + e];
  out[e] = t;
}

// ---- gradients of one step's coupling network from the GEMM results ------------------------------------------------
// Notation (flow_tfk_layers.py:73-84): r = relu(conv + b), h = g r + d with g = gamma / sqrt(var + eps), d = beta - mean g.
//   C3 [F+1][9c]   = [R2; 1] Gcol^T        C2 [F+1][F] = [R1; 1] M2^T        C1 [F][9ci+1] = M1 [Xcol; 1]^T
//   T1[f] = sum_q M1 R1,  T2[f] = sum_q M2 R2  -- not reduced over the pixels (two more passes over four planar arrays) but taken from the
//   GEMM results: M1 = mask1 . (K2 g_a2) and R1 vanishes where mask1 does, so T1[f1] = sum_f2 K2[f1][f2] g2[f2] C2[f1][f2]; likewise
//   T2[f] = sum_{tap,co} K3[tap][f][co] C3[f][tap c + co]
// and with  SM2[f] = C2[F][f] = sum_q M2,  SM1[f] = C1[f][9ci] = sum_q M1,  SG[k] = C3[F][k] = sum_q Gcol[k]:
//   dK3[tap][f][co] = g2[f] C3[f][tap c + co] + d2[f] SG[tap c + co]        db3[co] = SG[4 c + co]   (centre tap: no border)
//   SH2[f] = sum_q g_h2 = sum_{tap,co} K3[tap][f][co] SG[tap c + co]         dbeta2 = SH2    dgamma2 = (T2 - mean2 SH2) / sqrt(var2 + eps)
//   dK2[f1][f2] = g2[f2] (g1[f1] C2[f1][f2] + d1[f1] SM2[f2])                db2[f2] = g2[f2] SM2[f2]
//   SH1[f1] = sum_f2 K2[f1][f2] g2[f2] SM2[f2]                               dbeta1 = SH1    dgamma1 = (T1 - mean1 SH1) / sqrt(var1 + eps)
//   dK1[tap][ci][f] = g1[f] C1[f][tap ci_n + ci]                             db1[f] = g1[f] SM1[f]
// Every gradient is multiplied by `scale` (= -1 / global batch: the loss is the mean negative log-likelihood).
struct StepGradArgs {
  int F, c;
  const float* K2;      // [F][F] parameters (reference layout)
  const float* K3;      // [9][F][c]
  const float* bn;      // [8][F]: gamma1, beta1, mean1, var1, gamma2, beta2, mean2, var2
  const float* ep;      // [6][F]: b1, g1, d1, b2, g2, d2 -- the step's folded BatchNorm block as the kernels read it (k_fold_bn / pack_step)
  float eps;
  // scaled = 1: the planar arrays came from the split kernels (k_net_h3, MODE | 8) in the units those kernels split in:
  //   A1 = ACT 2^e1[f] R1,  A2 = ACT 2^e2[f] R2  (g = m 2^e: the power of two of the BatchNorm gain is folded into the producer),
  //   G2 = ACT g2[f] M2 = ACT g_a2,  G1 = ACT g1[f] M1 = ACT g_a1  (the backward images carry the gains),  ACT = GLOWK_ACT_SCALE.
  // The GEMMs ran on those; the per-row factors are undone here (powers of two and the gains themselves; a gain of exactly zero
  // -- a dead channel -- gets a zero gamma gradient).
  int scaled;
  const float *C1, *C2, *C3;
  float scale;
  float *dK1, *db1, *dgamma1, *dbeta1, *dK2, *db2, *dgamma2, *dbeta2, *dK3, *db3;
  // batch entry b (blockIdx.y: a step of the level): parameters and gradients advance by ps floats, ep by es, C1 / C2 / C3 by their strides
  size_t ps, es, c1s, c2s, c3s;
};
__device__ __forceinline__ void step_grad_batch(StepGradArgs& a, size_t b) {
  a.K2 += b * a.ps; a.K3 += b * a.ps; a.bn += b * a.ps; a.ep += b * a.es;
  a.C1 += b * a.c1s; a.C2 += b * a.c2s; a.C3 += b * a.c3s;
  a.dK1 += b * a.ps; a.db1 += b * a.ps; a.dgamma1 += b * a.ps; a.dbeta1 += b * a.ps; a.dK2 += b * a.ps; a.db2 += b * a.ps;
  a.dgamma2 += b * a.ps; a.dbeta2 += b * a.ps; a.dK3 += b * a.ps; a.db3 += b * a.ps;
}

__device__ __forceinline__ void bn_fold(const float* bn, int F, int layer, int f, float eps, float& g, float& d) {
  const float* b = bn + (size_t)layer * 4 * F;
  const double gd = (double)b[f] / sqrt((double)b[3 * F + f] + (double)eps);   // fp64 like the host packer (glowk.hip: pack_step): the
  g = (float)gd;                                                                // device-refreshed images equal the host-packed ones bit for bit
  d = (float)((double)b[F + f] - (double)b[2 * F + f] * gd);
}

__device__ __forceinline__ float pow2_of_gain_inv(float g) {   // 2^-e with g = m 2^e, |m| in [0.5, 1) (e = 0 for g = 0: pack_step)
  int e = 0;
  if (g != 0.0f) frexpf(g, &e);
  return ldexpf(1.0f, -e);
}

// grid: enough blocks of 256 threads to cover max(F * F, 9 * F * c, 9 * ci * F)
__global__ __launch_bounds__(256) void k_assemble_step_grads(StepGradArgs a) {
  step_grad_batch(a, blockIdx.y);
  const int F = a.F, c = a.c, ci = c / 2, N1 = 9 * ci + 1, N3 = 9 * c;
  const float *g1 = a.ep + F, *d1 = a.ep + 2 * (size_t)F, *g2 = a.ep + 4 * (size_t)F, *d2 = a.ep + 5 * (size_t)F;
  const float iact = a.scaled ? 1.0f / GLOWK_ACT_SCALE : 1.0f;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e < (size_t)F * F) {
    const int f1 = (int)(e / F), f2 = (int)(e % F);
    const float c2 = a.C2[(size_t)f1 * F + f2], s2 = a.C2[(size_t)F * F + f2];
    // exact:  g2[f2] (g1[f1] R1.M2 + d1[f1] sum M2);  scaled:  (g1[f1] a1[f1] A1.G2 + d1[f1] sum G2) / ACT  with a1 = 2^-e1 / ACT
    a.dK2[e] = a.scaled ? a.scale * iact * (g1[f1] * pow2_of_gain_inv(g1[f1]) * iact * c2 + d1[f1] * s2)
                        : a.scale * g2[f2] * (g1[f1] * c2 + d1[f1] * s2);
  }
  if (e < (size_t)9 * F * c) {
    const int tap = (int)(e / ((size_t)F * c)), f = (int)((e / c) % F), co = (int)(e % c);
    const float r2 = a.scaled ? pow2_of_gain_inv(g2[f]) * iact : 1.0f;
    a.dK3[e] = a.scale * (g2[f] * r2 * a.C3[(size_t)f * N3 + tap * c + co] + d2[f] * a.C3[(size_t)F * N3 + tap * c + co]);
  }
  if (e < (size_t)9 * ci * F) {
    const int k = (int)(e / F), f = (int)(e % F);       // dK1 [tap][ci][F] flattened as [(tap, ci)][F]
    a.dK1[e] = a.scale * (a.scaled ? iact : g1[f]) * a.C1[(size_t)f * N1 + k];
  }
  if (e < (size_t)c) a.db3[e] = a.scale * a.C3[(size_t)F * N3 + 4 * c + e];
}

// the per-channel part (biases, BatchNorm gamma / beta): one workgroup per hidden channel f, the two matrix-vector products
// SH2[f] = sum_k K3[k][f] SG[k] and SH1[f] = sum_f2 K2[f][f2] (sum_q g_a2[f2]) reduced across its threads (fp64, fixed order)
__global__ __launch_bounds__(256) void k_assemble_channel_grads(StepGradArgs a) {
  __shared__ double red[4][4];
  step_grad_batch(a, blockIdx.y);
  const int F = a.F, c = a.c, ci = c / 2, N1 = 9 * ci + 1, N3 = 9 * c, f = blockIdx.x;
  const float *g1 = a.ep + F, *g2 = a.ep + 4 * (size_t)F;
  const float iact = a.scaled ? 1.0f / GLOWK_ACT_SCALE : 1.0f;
  double sh2 = 0.0, sh1 = 0.0, t2 = 0.0, t1 = 0.0;
  for (int k = threadIdx.x; k < N3; k += 256) {
    const double k3 = (double)a.K3[((size_t)(k / c) * F + f) * c + (k % c)];
    sh2 += k3 * (double)a.C3[(size_t)F * N3 + k];
    t2 += k3 * (double)a.C3[(size_t)f * N3 + k];
  }
  for (int f2 = threadIdx.x; f2 < F; f2 += 256) {    // sum_q g_a2[f2] = g2 sum M2 (exact) = sum G2 / ACT (scaled)
    const double k2 = (double)a.K2[(size_t)f * F + f2];
    sh1 += k2 * (double)((a.scaled ? iact : g2[f2]) * a.C2[(size_t)F * F + f2]);
    t1 += k2 * (double)(a.scaled ? 1.0f : g2[f2]) * (double)a.C2[(size_t)f * F + f2];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sh2 += __shfl_down(sh2, o, 64); sh1 += __shfl_down(sh1, o, 64); t2 += __shfl_down(t2, o, 64); t1 += __shfl_down(t1, o, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sh2; red[1][threadIdx.x >> 6] = sh1; red[2][threadIdx.x >> 6] = t2; red[3][threadIdx.x >> 6] = t1; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  sh2 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  sh1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  t2 = red[2][0] + red[2][1] + red[2][2] + red[2][3];
  t1 = red[3][0] + red[3][1] + red[3][2] + red[3][3];
  const float* b1 = a.bn;
  const float* b2 = a.bn + (size_t)4 * F;
  // scaled: the GEMMs ran on A1 = ACT 2^e1 R1, G2 = ACT g2 M2 (C2s = ACT^2 2^e1[f1] g2[f2] C2: the gain is already in) and A2 = ACT 2^e2 R2
  // against the raw im2col of g_o (C3s = ACT 2^e2[f] C3)
  if (a.scaled) {
    t1 *= (double)pow2_of_gain_inv(g1[f]) * (double)iact * (double)iact;
    t2 *= (double)pow2_of_gain_inv(g2[f]) * (double)iact;
  }
  a.db2[f] = a.scale * (a.scaled ? iact : g2[f]) * a.C2[(size_t)F * F + f];
  a.dbeta2[f] = a.scale * (float)sh2;
  a.dgamma2[f] = a.scale * (float)((t2 - (double)b2[2 * F + f] * sh2) / sqrt((double)b2[3 * F + f] + (double)a.eps));
  a.db1[f] = a.scale * (a.scaled ? iact : g1[f]) * a.C1[(size_t)f * N1 + 9 * ci];
  a.dbeta1[f] = a.scale * (float)sh1;
  a.dgamma1[f] = a.scale * (float)((t1 - (double)b1[2 * F + f] * sh1) / sqrt((double)b1[3 * F + f] + (double)a.eps));
}

// ---- ActNorm + 1x1: the fused per-pixel affine v = u A + b.  part[block][c*c + c]: sum_q u_i g_j, sum_q g_j over the block's pixels
// (u is not kept by the forward pass: u = v Ainv + binv)
// blockIdx.y = batch entry (a step of the level): v / gv / Ainv / binv advance by their strides, part by gridDim.x * NOUT
template <int C>
__global__ __launch_bounds__(256) void k_affine_wgrad(const float* __restrict__ v, const float* __restrict__ gv, int Q, const float* __restrict__ Ainv,
                                                     const float* __restrict__ binv, double* __restrict__ part, ptrdiff_t v_bs, ptrdiff_t gv_bs,
                                                     ptrdiff_t a_bs, ptrdiff_t b_bs) {
  __shared__ float us[64][C + 1];
  __shared__ float gs[64][C + 1];
  constexpr int NOUT = C * C + C;
  v += (ptrdiff_t)blockIdx.y * v_bs; gv += (ptrdiff_t)blockIdx.y * gv_bs; Ainv += (ptrdiff_t)blockIdx.y * a_bs; binv += (ptrdiff_t)blockIdx.y * b_bs;
  part += (size_t)blockIdx.y * gridDim.x * NOUT;
  constexpr int PER = (NOUT + 255) / 256;
  double acc[PER];
#pragma unroll
  for (int o = 0; o < PER; ++o) acc[o] = 0.0;
  const int per_block = (Q + gridDim.x - 1) / gridDim.x;
  const int q_begin = blockIdx.x * per_block, q_end = q_begin + per_block < Q ? q_begin + per_block : Q;
  for (int q0 = q_begin; q0 < q_end; q0 += 64) {
    __syncthreads();
    if (threadIdx.x < 64) {
      const int q = q0 + threadIdx.x;
      if (q < q_end) {
        float x[C], u[C];
#pragma unroll
        for (int k = 0; k < C; ++k) x[k] = v[(size_t)q * C + k];
        affine_cc<C>(Ainv, binv, x, u);
#pragma unroll
        for (int k = 0; k < C; ++k) { us[threadIdx.x][k] = u[k]; gs[threadIdx.x][k] = gv[(size_t)q * C + k]; }
      } else {
#pragma unroll
        for (int k = 0; k < C; ++k) { us[threadIdx.x][k] = 0.0f; gs[threadIdx.x][k] = 0.0f; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < PER; ++o) {
      const int idx = threadIdx.x + 256 * o;
      if (idx < C * C) {
        const int i = idx / C, j = idx % C;
        float t = 0.0f;
        for (int p = 0; p < 64; ++p) t = fmaf(us[p][i], gs[p][j], t);
        acc[o] += (double)t;
      } else if (idx < NOUT) {
        const int j = idx - C * C;
        float t = 0.0f;
        for (int p = 0; p < 64; ++p) t += gs[p][j];
        acc[o] += (double)t;
      }
    }
  }
#pragma unroll
  for (int o = 0; o < PER; ++o) {
    const int idx = threadIdx.x + 256 * o;
    if (idx < NOUT) part[(size_t)blockIdx.x * NOUT + idx] = acc[o];
  }
}

// ---- prior (flow_builder.py:131-139): d sum_n logN / d loc = sum_n (z - loc) / s^2,  d / d log s = sum_n ((z - loc)^2 / s^2 - 1)
__global__ __launch_bounds__(256) void k_prior_wgrad(const float* __restrict__ z, int N, int E, const float* __restrict__ loc,
                                                    const float* __restrict__ log_scale, float scale, float* __restrict__ dloc, float* __restrict__ dls) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const float l = loc[e], v = log_scale[e], inv2 = expf(-2.0f * v);
  double a = 0.0, b = 0.0;
  for (int n = 0; n < N; ++n) {
    const float d = z[(size_t)n * E + e] - l;
    a += (double)(d * inv2);
    b += (double)(d * d * inv2 - 1.0f);
  }
  dloc[e] = scale * (float)a;
  dls[e] = scale * (float)b;
}

// ---- optimizers (train_utils.py:23-41: tfk.optimizers.Adam(lr) / Adamax(lr), Keras defaults beta1 0.9, beta2 0.999, eps 1e-7)
//   adamax:  m = b1 m + (1 - b1) g;  u = max(b2 u, |g|);  p -= lr / (1 - b1^t) * m / (u + eps)
//   adam:    m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)
__global__ __launch_bounds__(256) void k_optimizer(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  size_t n, int adamax, float lr_t, float b1, float b2, float eps) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const float ge = g[e];
  const float me = b1 * m[e] + (1.0f - b1) * ge;
  float ve;
  if (adamax) {
    ve = fmaxf(b2 * v[e], fabsf(ge));
    p[e] -= lr_t * me / (ve + eps);
  } else {
    ve = b2 * v[e] + (1.0f - b2) * ge * ge;
    p[e] -= lr_t * me / (sqrtf(ve) + eps);
  }
  m[e] = me;
  v[e] = ve;
}

// ---- device-side refresh of the exact-fp32 kernel images after an optimizer step ---------------------------------------
// img[i] = map[i] < 0 ? 0 : params[map[i]]: the packed images are permutations (with zero padding) of the three conv kernels;
// the map is the host packer's own output on index-coded tensors, one map per level (glowk.hip: build_repack_map).
__global__ __launch_bounds__(256) void k_repack_f32(const int* __restrict__ map, size_t n, const float* __restrict__ params, size_t param_stride,
                                                   float* __restrict__ img, size_t img_stride) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int s = map[i];
  if (s == -2) return;               // not a permutation of the conv kernels (the BatchNorm / bias block: k_fold_bn)
  const size_t step = blockIdx.y;    // one grid row per step of the level
  img[step * img_stride + i] = s < 0 ? 0.0f : params[step * param_stride + s];
}

// ep block of a step: [b1 | g1 | d1 | b2 | g2 | d2] from biases and BatchNorm tensors
__global__ __launch_bounds__(256) void k_fold_bn(const float* __restrict__ b1, const float* __restrict__ b2, const float* __restrict__ bn, size_t param_stride,
                                                int F, float eps, float* __restrict__ ep, size_t img_stride) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const size_t po = (size_t)blockIdx.y * param_stride;
  float* e = ep + (size_t)blockIdx.y * img_stride;
  float g1, d1, g2, d2;
  bn_fold(bn + po, F, 0, f, eps, g1, d1);
  bn_fold(bn + po, F, 1, f, eps, g2, d2);
  e[f] = b1[po + f]; e[F + f] = g1; e[2 * F + f] = d1;
  e[3 * F + f] = b2[po + f]; e[4 * F + f] = g2; e[5 * F + f] = d2;
}

// ---- device-side refresh of the fp16-split kernel images after an optimizer step -----------------------------------------
// The host packer (glowk_pack.h: pack_step) folds BatchNorm into the weights, scales every layer by a power of two taken from
// its largest element, splits into fp16 hi + lo and lays the halves out in MFMA fragment order.  Here the layout is a MAP made
// by the packer itself in map mode (one int per half: source code | lo bit, see F16Codes), and the arithmetic is redone on the
// device in the packer's own precision and order (fp64 where it uses fp64, separate multiply and add where the x86 build has
// no fused ones), so that the refreshed images are bit for bit the host-packed ones -- which is what the test asserts.
// All kernels take one grid row (blockIdx.y) per step of the level.
struct F16Prep {
  const float* params;     // first step's parameter block (TrainOff layout), param_stride floats between steps
  size_t param_stride;
  const float* ep;         // first step's folded block [b1 | g1 | d1 | b2 | g2 | d2] in the arena, img_stride floats between steps
  size_t img_stride;
  size_t oK1, oK2, oK3, ob1, ob2;   // TrainOff offsets
  int c, F;
  size_t cA, cB, cC, cD, cE, cG, cTot;   // F16Codes
  float* src;              // [steps][cTot] scaled fp32 sources
  int* S;                  // [steps][6] power-of-two scales of the six source arrays
  float* scales;           // [steps][8]: sc1 sc2 sc3 scb1 scb2 scb3 xlim_f xlim_b (what the host packer hands back in scales3)
};

__device__ __forceinline__ int gain_exp(float g) { int e = 0; if (g != 0.0f) frexp((double)g, &e); return e; }
__device__ __forceinline__ double gain_mant(float g) { int e; return frexp((double)g, &e); }

__global__ __launch_bounds__(256) void k_f16_sources(F16Prep a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.cTot) return;
  const int F = a.F, c = a.c, ci = c / 2;
  const float* p = a.params + (size_t)blockIdx.y * a.param_stride;
  const float* ep = a.ep + (size_t)blockIdx.y * a.img_stride;
  const float *K1 = p + a.oK1, *K2 = p + a.oK2, *K3 = p + a.oK3, *g1 = ep + F, *g2 = ep + 4 * (size_t)F;
  float v;
  if (i < a.cB) {                       // K1f[kk][f] = ldexp(K1[kk][f] | b1[f], e1[f])
    const int kk = (int)(i / F), f = (int)(i % F);
    v = ldexpf(kk < 9 * ci ? K1[(size_t)kk * F + f] : ep[f], gain_exp(g1[f]));
  } else if (i < a.cC) {                // K2f[fi][fo] = (float) ldexp(K2 m1[fi], e2[fo])
    const size_t j = i - a.cB;
    const int fi = (int)(j / F), fo = (int)(j % F);
    v = (float)ldexp(__dmul_rn((double)K2[j], gain_mant(g1[fi])), gain_exp(g2[fo]));
  } else if (i < a.cD) {                // K3f[tap][f][co] = (float)(K3 m2[f])
    const size_t j = i - a.cC;
    const int f = (int)((j / c) % F);
    v = (float)__dmul_rn((double)K3[j], gain_mant(g2[f]));
  } else if (i < a.cE) {                // W3b[kk = (tap, co)][f] = K3[tap][f][co] g2[f]   (float product)
    const size_t j = i - a.cD;
    const int kk = (int)(j / F), f = (int)(j % F), tap = kk / c, co = kk % c;
    v = __fmul_rn(K3[((size_t)tap * F + f) * c + co], g2[f]);
  } else if (i < a.cG) {                // W2b[f2][f1] = K2[f1][f2] g1[f1]
    const size_t j = i - a.cE;
    const int f2 = (int)(j / F), f1 = (int)(j % F);
    v = __fmul_rn(K2[(size_t)f1 * F + f2], g1[f1]);
  } else {
    v = K1[i - a.cG];
  }
  a.src[(size_t)blockIdx.y * a.cTot + i] = v;
}

// S = 14 - exponent(max |w|) of each of the six source arrays (grid: 6 x steps, 1024 threads)
__global__ __launch_bounds__(1024) void k_f16_absmax(F16Prep a) {
  __shared__ float red[16];
  const size_t lo[6] = {a.cA, a.cB, a.cC, a.cD, a.cE, a.cG}, hi[6] = {a.cB, a.cC, a.cD, a.cE, a.cG, a.cTot};
  const int r = blockIdx.x;
  const float* s = a.src + (size_t)blockIdx.y * a.cTot;
  float m = 0.0f;
  for (size_t i = lo[r] + threadIdx.x; i < hi[r]; i += 1024) m = fmaxf(m, fabsf(s[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    int e = 0;
    if (m > 0.0f) frexpf(m, &e);
    a.S[(size_t)blockIdx.y * 6 + r] = 14 - e;
  }
}

// epH block of a step: [conv2 accumulator init (F) | per-row constants of P (32 NMT)], and the six kernel scales
__global__ __launch_bounds__(256) void k_f16_consts(F16Prep a, float* __restrict__ eph0, int NMT) {
  const int F = a.F, c = a.c, m = blockIdx.x * 256 + threadIdx.x;
  const float* p = a.params + (size_t)blockIdx.y * a.param_stride;
  const float* ep = a.ep + (size_t)blockIdx.y * a.img_stride;
  float* eh = eph0 + (size_t)blockIdx.y * a.img_stride;
  const int* S = a.S + (size_t)blockIdx.y * 6;
  const float act = GLOWK_ACT_SCALE;
  if (m < F) {            // b2f[fo] = b2 + sum_fi K2[fi][fo] d1[fi]  (fp64, fi ascending, multiply and add separately)
    const float* K2 = p + a.oK2;
    double b = (double)ep[3 * (size_t)F + m];
    for (int fi = 0; fi < F; ++fi) b = __dadd_rn(b, __dmul_rn((double)K2[(size_t)fi * F + m], (double)ep[2 * (size_t)F + fi]));
    eh[m] = (float)ldexp(__dmul_rn((double)act, b), gain_exp(ep[4 * (size_t)F + m]) + S[1]);
  } else if (m < F + 32 * NMT) {
    const int k = m - F;
    double pb = 0.0;
    if (k < 9 * c) {
      const float* K3 = p + a.oK3;
      const int tap = k / c, co = k % c;
      for (int f = 0; f < F; ++f) pb = __dadd_rn(pb, __dmul_rn((double)K3[((size_t)tap * F + f) * c + co], (double)ep[5 * (size_t)F + f]));
    }
    eh[m] = (float)pb;
  }
  if (m == 0) {
    float* sc = a.scales + (size_t)blockIdx.y * 8;
    sc[0] = ldexpf(1.0f, -S[0]); sc[1] = ldexpf(1.0f, -S[1]); sc[2] = ldexpf(1.0f, -S[2]) / act;
    sc[3] = ldexpf(1.0f, -S[3]); sc[4] = ldexpf(1.0f, -S[4]); sc[5] = ldexpf(1.0f, -S[5]) / act;
  }
}

// the range-guard limits of a step (pack_step: "range guard, forward / backward network"), one workgroup of F threads per step
// and direction (blockIdx.x = 0 forward, 1 backward)
__global__ __launch_bounds__(512) void k_f16_limits(F16Prep a) {
  __shared__ double sh[512];
  __shared__ double red[512];
  const int F = a.F, c = a.c, ci = c / 2, f = threadIdx.x;
  const float* p = a.params + (size_t)blockIdx.y * a.param_stride;
  const float* ep = a.ep + (size_t)blockIdx.y * a.img_stride;
  const double LIM = 60000.0 / (double)GLOWK_ACT_SCALE;
  float* sc = a.scales + (size_t)blockIdx.y * 8;
  if (blockIdx.x == 0) {
    const float *K1 = p + a.oK1, *K2 = p + a.oK2, *g1 = ep + F, *g2 = ep + 4 * (size_t)F;
    double n1 = 0.0;
    if (f < F)
      for (int kk = 0; kk < 9 * ci; ++kk) n1 = __dadd_rn(n1, fabs((double)K1[(size_t)kk * F + f]));
    if (f < F) sh[f] = n1;
    __syncthreads();
    double cand = LIM;
    auto tighten = [&](double A, double B) {
      const double room = LIM - B;
      if (room <= 0.0) cand = 0.0;
      else if (A > 0.0) cand = fmin(cand, room / A);
    };
    if (f < F) {
      tighten(ldexp(n1, gain_exp(g1[f])), ldexp(fabs((double)ep[f]), gain_exp(g1[f])));
      double A2 = 0.0, B2 = 0.0;
      for (int fi = 0; fi < F; ++fi) {
        const double ga = fabs((double)g1[fi]);
        const double aa = __dmul_rn(ga, sh[fi]), bb = __dadd_rn(__dmul_rn(ga, fabs((double)ep[fi])), fabs((double)ep[2 * (size_t)F + fi]));
        const double w = fabs((double)K2[(size_t)fi * F + f]);
        A2 = __dadd_rn(A2, __dmul_rn(w, aa));
        B2 = __dadd_rn(B2, __dmul_rn(w, bb));
      }
      tighten(ldexp(A2, gain_exp(g2[f])), ldexp(__dadd_rn(B2, fabs((double)ep[3 * (size_t)F + f])), gain_exp(g2[f])));
    }
    red[f] = f < F ? cand : LIM;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) { if (f < o) red[f] = fmin(red[f], red[f + o]); __syncthreads(); }
    if (f == 0) sc[6] = (float)(red[0] * (double)GLOWK_ACT_SCALE * (1.0 - 1e-6));
  } else {
    const float* s = a.src + (size_t)blockIdx.y * a.cTot;
    const float *W3b = s + a.cD, *W2b = s + a.cE;
    double a1 = 0.0;
    if (f < F)
      for (int kk = 0; kk < 9 * c; ++kk) a1 = __dadd_rn(a1, fabs((double)W3b[(size_t)kk * F + f]));
    sh[f] = f < F ? a1 : 0.0;
    __syncthreads();
    double a2 = 0.0;
    if (f < F)
      for (int f2 = 0; f2 < F; ++f2) a2 = __dadd_rn(a2, __dmul_rn(fabs((double)W2b[(size_t)f2 * F + f]), sh[f2]));
    red[f] = f < F ? fmax(a1, a2) : 0.0;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) { if (f < o) red[f] = fmax(red[f], red[f + o]); __syncthreads(); }
    if (f == 0) sc[7] = (float)(LIM / fmax(1.0, red[0]) * (double)GLOWK_ACT_SCALE * (1.0 - 1e-6));
  }
}

// IEEE binary32 -> binary16, round to nearest even, and back (bit-for-bit glowk_pack.h: f32_to_f16 / f16_to_f32)
__device__ __forceinline__ unsigned short dev_f32_to_f16(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ float dev_f16_to_f32(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }

// img16[pos] = hi or lo half of ldexp(src[code - 1], S[range of code]); map[pos] = code | lo << 30, or -1: leave alone
__device__ __forceinline__ unsigned short repack_f16_half(int mv, const F16Prep& a) {
  const int lo = (mv >> 30) & 1;
  const size_t code = (size_t)(mv & 0x3FFFFFFF);
  if (!code) return 0;
  const size_t i = code - 1;
  const int r = i < a.cB ? 0 : i < a.cC ? 1 : i < a.cD ? 2 : i < a.cE ? 3 : i < a.cG ? 4 : 5;
  const float ws = ldexpf(a.src[(size_t)blockIdx.y * a.cTot + i], a.S[(size_t)blockIdx.y * 6 + r]);
  const unsigned short hi = dev_f32_to_f16(ws);
  return lo ? dev_f32_to_f16(ws - dev_f16_to_f32(hi)) : hi;
}

// four consecutive halves per thread: one 16-byte read of the map, one 8-byte store (one half per thread -- 2-byte stores, 4 bytes of map
// per 2 bytes of image -- ran at 0.4 TB/s: 590 us per level and step batch); a group with a "leave alone" entry stores half by half
__global__ __launch_bounds__(256) void k_repack_f16(const int* __restrict__ map, size_t n, F16Prep a, unsigned short* __restrict__ img16, size_t img_stride_halves) {
  const size_t pos = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (pos >= n) return;
  unsigned short* dst = img16 + (size_t)blockIdx.y * img_stride_halves + pos;
  if (pos + 4 <= n && ((img_stride_halves | (size_t)(reinterpret_cast<uintptr_t>(img16) >> 1)) & 3) == 0) {
    const int4 mv = *reinterpret_cast<const int4*>(map + pos);
    if (mv.x >= 0 && mv.y >= 0 && mv.z >= 0 && mv.w >= 0) {
      const unsigned h0 = repack_f16_half(mv.x, a), h1 = repack_f16_half(mv.y, a), h2 = repack_f16_half(mv.z, a), h3 = repack_f16_half(mv.w, a);
      *reinterpret_cast<uint2*>(dst) = uint2{h0 | (h1 << 16), h2 | (h3 << 16)};
      return;
    }
  }
  for (int e = 0; e < 4 && pos + e < n; ++e) {
    const int mv = map[pos + e];
    if (mv >= 0) dst[e] = repack_f16_half(mv, a);
  }
}
