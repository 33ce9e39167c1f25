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
// Weights stream global -> LDS by LDS-DMA (global_load_lds_dwordx4), double buffered, 64 KiB chunks,
// pre-packed on the host into the exact lane order the MFMA A operand wants (ds_read_b128, conflict free).
#pragma once
#include <hip/hip_runtime.h>
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
  const float4* K2p;     // [NF fi][16 r][NF/4][64 lanes] float4 (4 consecutive fo)
  const float4* K3p;     // [NMT][NF fo][4 r4][64 lanes] float4 (4 consecutive r)
  float* P;              // [9*CO][Q]
};

// one 64-lane LDS-DMA piece: LDS destination = wave-uniform base + lane*16, global source per lane
__device__ __forceinline__ void glds16(const float4* src_lane, float4* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(GLOWK_GPTR(src_lane), GLOWK_LPTR(lds_wave_base), 16, 0, 0);
}

template <int NPER>  // float4 per thread; chunk = NPER*256 float4, copied linearly
__device__ __forceinline__ void stage_chunk(const float4* __restrict__ src, float4* dst, int tid) {
  const int wbase = tid & ~63;
  const int lane = tid & 63;
#pragma unroll
  for (int it = 0; it < NPER; ++it) {
    const int base = it * 256 + wbase;
    glds16(src + base + lane, dst + base);
  }
}

template <int CI, int NF>
__global__ __launch_bounds__(256, 1) void k_net_f32(NetArgs a) {
  constexpr int CO = 2 * CI;
  constexpr int KS1 = (9 * CI) / 2;   // k-steps (k = 2) of conv1's contraction over (tap, cin)
  constexpr int M3 = 9 * CO;          // rows of P^T: (tap, cout)
  constexpr int NMT = (M3 + 31) / 32;
  constexpr int F = NF * 32;
  constexpr int CHUNK4 = NF * 256;    // float4 per staged chunk (NF * 4 KiB)
  static_assert(CI % 2 == 0 && NF % 4 == 0, "shape");

  extern __shared__ float4 lds4[];    // [2][CHUNK4] weight ring, then ep [6][F]
  float* epl = reinterpret_cast<float*>(lds4 + 2 * CHUNK4);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int pix = lane & 31;
  const int hh = lane >> 5;
  const int q = (blockIdx.x * 4 + wave) * 32 + pix;
  const bool qok = q < a.Q;

  for (int i = tid; i < 6 * F; i += 256) epl[i] = a.ep[i];
  stage_chunk<NF>(a.K2p, lds4, tid);

  // im2col column of this lane's pixel: xcol[ks] = vb[pixel + d(tap)][cin], k = 2*ks + hh = tap*CI + cin
  float xcol[KS1];
  {
    const int hw = a.h * a.w;
    const int rem = q % hw;
    const int i = rem / a.w, j = rem % a.w;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const int k = 2 * ks + hh;
      const int tap = k / CI, cin = k % CI;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const int ii = i + dy, jj = j + dx;
      const bool ok = qok && ii >= 0 && ii < a.h && jj >= 0 && jj < a.w;
      const long src = (long)(q + dy * a.w + dx) * a.in_stride + a.in_off + cin;
      xcol[ks] = ok ? a.vin[src] : 0.0f;
    }
  }

  f32x16 acc2[NF];
#pragma unroll
  for (int fo = 0; fo < NF; ++fo)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[fo][r] = 0.0f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // ep visible; chunk 0 landed

  for (int fi = 0; fi < NF; ++fi) {
    // ---- conv1 for hidden channels [32 fi, 32 fi + 32): A1^T tile, then bias + ReLU + BN1 ----
    f32x16 h1;
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] = 0.0f;
    {
      const float* k1 = a.K1p + (size_t)fi * KS1 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[ks * 64], xcol[ks], h1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = fi * 32 + mfma_row(r, hh);
      h1[r] = fmaf(epl[F + f], fmaxf(h1[r] + epl[f], 0.0f), epl[2 * F + f]);
    }
    if (fi > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // chunk fi landed in every wave's view; buffer (fi+1)&1 is no longer read
    }
    if (fi + 1 < NF) stage_chunk<NF>(a.K2p + (size_t)(fi + 1) * CHUNK4, lds4 + ((fi + 1) & 1) * CHUNK4, tid);
    else             stage_chunk<NF>(a.K3p, lds4 + (NF & 1) * CHUNK4, tid);

    // ---- conv2: acc2[fo] += K2^T[fo-tile, fi-tile] . H1^T tile ----
    const float4* buf = lds4 + (fi & 1) * CHUNK4;
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

  // ---- bias + ReLU + BN2 in place: acc2 becomes H2^T ----
#pragma unroll
  for (int fo = 0; fo < NF; ++fo)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = fo * 32 + mfma_row(r, hh);
      acc2[fo][r] = fmaf(epl[4 * F + f], fmaxf(acc2[fo][r] + epl[3 * F + f], 0.0f), epl[5 * F + f]);
    }

  // ---- conv3 as nine per-tap 1x1 convolutions: P^T[(tap,co) x 32px] = K3r^T . H2^T ----
  for (int mt = 0; mt < NMT; ++mt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (mt + 1 < NMT) stage_chunk<NF>(a.K3p + (size_t)(mt + 1) * CHUNK4, lds4 + ((NF + mt + 1) & 1) * CHUNK4, tid);
    const float4* buf = lds4 + ((NF + mt) & 1) * CHUNK4;
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
      if (m < M3 && qok) a.P[(size_t)m * a.Q + q] = acc3[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// light kernels: one 256-thread workgroup per sample (deterministic per-sample reductions, no atomics)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double* red /* [4] in LDS */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// y[co] = b[co] + sum_ci x[ci] * A[ci][co]   (A row-major [C][C]; uniform addresses -> scalar loads)
template <int C>
__device__ __forceinline__ void affine_cc(const float* __restrict__ A, const float* __restrict__ b, const float (&x)[C], float (&y)[C]) {
#pragma unroll
  for (int co = 0; co < C; ++co) y[co] = b[co];
#pragma unroll
  for (int ci = 0; ci < C; ++ci)
#pragma unroll
    for (int co = 0; co < C; ++co) y[co] = fmaf(x[ci], A[ci * C + co], y[co]);
}

struct PreArgs {
  float minval, maxval, alpha;
  int use_logit;
};

// SpecPreprocessing._forward on one element (flow_tfp_bijectors.py:372-379); ld accumulates the
// data-dependent part of the logit log-det (-log p - log(1-p), :394)
__device__ __forceinline__ float pre_fwd(float x, const PreArgs& p, float& ld) {
  float u = (x - p.minval) / (p.maxval - p.minval);
  if (p.use_logit) {
    u = (1.0f - 2.0f * p.alpha) * u + p.alpha;
    const float lp = logf(u), lq = logf(1.0f - u);
    ld += -lp - lq;
    return lp - lq;
  }
  return u - 0.5f;
}

__device__ __forceinline__ float pre_inv(float y, const PreArgs& p) {
  if (p.use_logit) {
    y = 1.0f / (1.0f + expf(-y));
    y = (y - p.alpha) / (1.0f - 2.0f * p.alpha);
  } else {
    y += 0.5f;
  }
  return y * (p.maxval - p.minval) + p.minval;
}

// x [N,H,W,Cin] -> preprocess -> squeeze -> (optional) ActNorm+1x1 of the first step -> v [N,H/2,W/2,C]
// logdet[n] = ld_const + data-dependent preprocessing log-det
template <int C>
__global__ __launch_bounds__(256) void k_in(const float* __restrict__ x, int H, int W, PreArgs pre, int do_pre,
                                           const float* __restrict__ A, const float* __restrict__ b,
                                           float* __restrict__ v, double* __restrict__ logdet, double ld_const) {
  __shared__ double red[4];
  constexpr int Cin = C / 4;
  const int n = blockIdx.x, h = H / 2, w = W / 2;
  float ld = 0.0f;
  for (int pp = threadIdx.x; pp < h * w; pp += 256) {
    const int i = pp / w, j = pp % w;
    float u[C], y[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
      const int cin = cc >> 2, da = (cc >> 1) & 1, db = cc & 1;
      const float xv = x[((size_t)(n * H + 2 * i + da) * W + (2 * j + db)) * Cin + cin];
      u[cc] = do_pre ? pre_fwd(xv, pre, ld) : xv;
    }
    float* dst = v + ((size_t)n * h * w + pp) * C;
    if (A) {
      affine_cc<C>(A, b, u, y);
#pragma unroll
      for (int cc = 0; cc < C; ++cc) dst[cc] = y[cc];
    } else {
#pragma unroll
      for (int cc = 0; cc < C; ++cc) dst[cc] = u[cc];
    }
  }
  if (logdet) {
    const double tot = block_sum_256((double)ld, red);
    if (threadIdx.x == 0) logdet[n] = ld_const + tot;
  }
}

// u [N,h,w,C] -> unsqueeze -> (optional) SpecPreprocessing inverse -> x [N,2h,2w,C/4]
template <int C>
__global__ __launch_bounds__(256) void k_out(const float* __restrict__ u, int h, int w, PreArgs pre, int do_pre,
                                            float* __restrict__ x) {
  constexpr int Cin = C / 4;
  const int n = blockIdx.x, H = 2 * h, W = 2 * w;
  for (int pp = threadIdx.x; pp < h * w; pp += 256) {
    const int i = pp / w, j = pp % w;
    const float* src = u + ((size_t)n * h * w + pp) * C;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
      const int cin = cc >> 2, da = (cc >> 1) & 1, db = cc & 1;
      const float val = src[cc];
      x[((size_t)(n * H + 2 * i + da) * W + (2 * j + db)) * Cin + cin] = do_pre ? pre_inv(val, pre) : val;
    }
  }
}

// elementwise SpecPreprocessing forward / inverse with per-sample log-det (test entry points)
__global__ __launch_bounds__(256) void k_pre_only(const float* __restrict__ x, int E, PreArgs pre, int inverse,
                                                 float* __restrict__ y, float* __restrict__ logdet, double ld_const) {
  __shared__ double red[4];
  const int n = blockIdx.x;
  float ld = 0.0f;
  for (int e = threadIdx.x; e < E; e += 256) {
    const float xv = x[(size_t)n * E + e];
    y[(size_t)n * E + e] = inverse ? pre_inv(xv, pre) : pre_fwd(xv, pre, ld);
  }
  if (logdet) {
    const double tot = block_sum_256((double)ld, red);
    if (threadIdx.x == 0) logdet[n] = (float)(ld_const + tot);
  }
}

struct CoupleArgs {
  const float* vin;     // [Q][C]: forward: v = 1x1(actnorm(u)); inverse: y
  const float* P;       // [9C][Q] per-tap partial conv3 outputs of the network evaluated on vin[.., C/2:]
  const float* b3;      // [C] conv3 bias
  const float* A;       // post affine [C][C] or null: forward = NEXT step's ActNorm+1x1, inverse = this step's inverse 1x1+ActNorm
  const float* b;       // [C]
  float* out;           // element (q, co) at out[q*out_stride + out_off + co]
  int out_stride, out_off;
  double* logdet;       // [N] += sum log_s (forward only; may be null)
  float* log_s_out;     // optional [Q][C/2] dumps of the network outputs (glowk_coupling_net)
  float* t_out;
  int Q, h, w;
  int inverse;
};

// gather conv3 (9 taps), split + tanh (flow_tfk_layers.py:80-84), affine coupling
// (flow_tfp_bijectors.py:134-148), per-sample log-det (:150-153), then the following per-pixel affine
template <int C>
__global__ __launch_bounds__(256) void k_couple(CoupleArgs a) {
  __shared__ double red[4];
  constexpr int CI = C / 2;
  const int n = blockIdx.x;
  const int hw = a.h * a.w;
  float lsum = 0.0f;
  for (int pp = threadIdx.x; pp < hw; pp += 256) {
    const int q = n * hw + pp;
    const int i = pp / a.w, j = pp % a.w;
    float v[C], o[C];
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = a.b3[c];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const int ii = i + dy, jj = j + dx;
      if (ii >= 0 && ii < a.h && jj >= 0 && jj < a.w) {
        const float* src = a.P + (size_t)(tap * C) * a.Q + (q + dy * a.w + dx);
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] += src[(size_t)c * a.Q];
      }
    }
    if (a.vin) {
#pragma unroll
      for (int c = 0; c < C; ++c) v[c] = a.vin[(size_t)q * C + c];
    }
    float y[C];
#pragma unroll
    for (int k = 0; k < CI; ++k) {
      const float log_s = tanhf(o[k]);
      const float t = o[CI + k];
      if (a.log_s_out) { a.log_s_out[(size_t)q * CI + k] = log_s; a.t_out[(size_t)q * CI + k] = t; }
      if (a.vin) {
        const float s = expf(log_s);
        y[k] = a.inverse ? (v[k] - t) / s : s * v[k] + t;
        y[CI + k] = v[CI + k];
      }
      lsum += log_s;
    }
    if (a.out) {
      float* dst = a.out + (size_t)q * a.out_stride + a.out_off;
      if (a.A) {
        float z[C];
        affine_cc<C>(a.A, a.b, y, z);
#pragma unroll
        for (int c = 0; c < C; ++c) dst[c] = z[c];
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) dst[c] = y[c];
      }
    }
  }
  if (a.logdet) {
    const double tot = block_sum_256((double)lsum, red);
    if (threadIdx.x == 0) a.logdet[n] += tot;
  }
}

// plain per-pixel affine (ActNorm+1x1 of a step applied to a materialised tensor): test entry / first steps
template <int C>
__global__ __launch_bounds__(256) void k_affine(const float* __restrict__ in, int Q, const float* __restrict__ A,
                                               const float* __restrict__ b, float* __restrict__ out) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= Q) return;
  float x[C], y[C];
#pragma unroll
  for (int c = 0; c < C; ++c) x[c] = in[(size_t)q * C + c];
  affine_cc<C>(A, b, x, y);
#pragma unroll
  for (int c = 0; c < C; ++c) out[(size_t)q * C + c] = y[c];
}

// end of a non-final block (flow_glow.py:104-106,177-182): o [N,h,w,C] -> first half row-major into the
// latent (plain reshape, not a squeeze), second half squeezed (+ first ActNorm+1x1 of the next block) -> v
template <int C>
__global__ __launch_bounds__(256) void k_split(const float* __restrict__ o, int h, int w,
                                              float* __restrict__ latent, int HWl, int Cl, int off, int Cz,
                                              const float* __restrict__ A, const float* __restrict__ b,
                                              float* __restrict__ vnext) {
  constexpr int CH = C / 2;     // channels of each half
  constexpr int C2 = 2 * C;     // channels of the next level (4 * CH)
  const int n = blockIdx.x, hw = h * w;
  const float* on = o + (size_t)n * hw * C;
  if (latent) {
    for (int e = threadIdx.x; e < hw * CH; e += 256) {
      const int pp = e / CH, k = e % CH;
      latent[((size_t)n * HWl + e / Cz) * Cl + off + e % Cz] = on[(size_t)pp * C + k];
    }
  }
  const int h2 = h / 2, w2 = w / 2;
  for (int pp = threadIdx.x; pp < h2 * w2; pp += 256) {
    const int i = pp / w2, j = pp % w2;
    float u[C2], y[C2];
#pragma unroll
    for (int cc = 0; cc < C2; ++cc) {
      const int cin = cc >> 2, da = (cc >> 1) & 1, db = cc & 1;
      u[cc] = on[(size_t)((2 * i + da) * w + 2 * j + db) * C + CH + cin];
    }
    float* dst = vnext + ((size_t)n * h2 * w2 + pp) * C2;
    if (A) {
      affine_cc<C2>(A, b, u, y);
#pragma unroll
      for (int cc = 0; cc < C2; ++cc) dst[cc] = y[cc];
    } else {
#pragma unroll
      for (int cc = 0; cc < C2; ++cc) dst[cc] = u[cc];
    }
  }
}

// inverse of k_split (flow_glow.py:110-116,187-195): y [N,h,w,C] = concat(reshape(latent slice), unsqueeze(unext))
// for the last level (unext == null) y = latent[..., off:off+C]
template <int C>
__global__ __launch_bounds__(256) void k_unsplit(const float* __restrict__ latent, int HWl, int Cl, int off, int Cz,
                                                const float* __restrict__ unext, int h, int w, float* __restrict__ y) {
  constexpr int CH = C / 2;
  const int n = blockIdx.x, hw = h * w;
  float* yn = y + (size_t)n * hw * C;
  if (!unext) {
    for (int e = threadIdx.x; e < hw * C; e += 256) {
      const int pp = e / C, k = e % C;
      yn[e] = latent[((size_t)n * HWl + pp) * Cl + off + k];
    }
    return;
  }
  for (int e = threadIdx.x; e < hw * CH; e += 256) {
    const int pp = e / CH, k = e % CH;
    yn[(size_t)pp * C + k] = latent[((size_t)n * HWl + e / Cz) * Cl + off + e % Cz];
  }
  constexpr int C2 = 2 * C;
  const int h2 = h / 2, w2 = w / 2;
  const float* un = unext + (size_t)n * h2 * w2 * C2;
  for (int e = threadIdx.x; e < h2 * w2 * C2; e += 256) {
    const int pp = e / C2, cc = e % C2;
    const int i = pp / w2, j = pp % w2;
    const int cin = cc >> 2, da = (cc >> 1) & 1, db = cc & 1;
    yn[(size_t)((2 * i + da) * w + 2 * j + db) * C + CH + cin] = un[e];
  }
}

// prior log-density (flow_builder.py:131-144) + accumulated log-det -> logp[n]
__global__ __launch_bounds__(256) void k_prior(const float* __restrict__ z, int E, const float* __restrict__ loc,
                                              const float* __restrict__ log_scale, const double* __restrict__ logdet,
                                              float* __restrict__ logp, float* __restrict__ logdet_out) {
  __shared__ double red[4];
  const int n = blockIdx.x;
  const float HALF_LOG_2PI = 0.91893853320467274178f;
  double acc = 0.0;
  if (logp) {
    for (int e = threadIdx.x; e < E; e += 256) {
      const float zv = z[(size_t)n * E + e];
      float lp;
      if (loc) {
        const float v = log_scale[e];
        const float d = (zv - loc[e]) / expf(v);
        lp = -0.5f * d * d - v - HALF_LOG_2PI;
      } else {
        lp = -0.5f * zv * zv - HALF_LOG_2PI;
      }
      acc += (double)lp;
    }
    const double tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) logp[n] = (float)(tot + (logdet ? logdet[n] : 0.0));
  }
  if (logdet_out && threadIdx.x == 0) logdet_out[n] = (float)logdet[n];
}

// z = loc + exp(log_scale) * eps (prior.sample given the standard-normal draw)
__global__ __launch_bounds__(256) void k_prior_sample(const float* __restrict__ eps, size_t total, int E,
                                                     const float* __restrict__ loc, const float* __restrict__ log_scale,
                                                     float* __restrict__ z) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int e = (int)(idx % E);
  z[idx] = loc ? loc[e] + expf(log_scale[e]) * eps[idx] : eps[idx];
}

// per-channel partial sums for ActNorm's data-dependent init (flow_tfp_bijectors.py:222-234):
// part[block][c] = sum over this block's pixels of x (mean == null) or (x - mean[c])^2
template <int C>
__global__ __launch_bounds__(256) void k_chan_stats(const float* __restrict__ x, int Q, const double* __restrict__ mean,
                                                   double* __restrict__ part) {
  __shared__ double red[4][C];
  double acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = 0.0;
  for (int q = blockIdx.x * 256 + threadIdx.x; q < Q; q += gridDim.x * 256) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const double v = (double)x[(size_t)q * C + c];
      if (mean) { const double d = v - mean[c]; acc[c] += d * d; } else acc[c] += v;
    }
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    double v = acc[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < C) part[(size_t)blockIdx.x * C + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
