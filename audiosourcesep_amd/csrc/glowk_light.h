// glowk device code, part 2: the light kernels (HBM / latency bound): preprocessing + squeeze, coupling + log-det, factor-out,
// prior, ActNorm statistics, and the light half of the input-gradient path.  One 256-thread workgroup per sample unless noted
// (deterministic per-sample reductions, no atomics).  Only glowk.hip includes this file; the per-shape translation units
// of the coupling-network kernels (glowk_net_inst.hip) see glowk_kernels.h alone, so editing a light kernel rebuilds one object.
#pragma once
#include "glowk_kernels.h"

// ------------------------------------------------------------------------------------------------
// light kernels: one 256-thread workgroup per sample (deterministic per-sample reductions, no atomics)
// ------------------------------------------------------------------------------------------------
// same for any workgroup of whole waves up to 1024 threads (deterministic: fixed order over the waves)
__device__ __forceinline__ double block_sum_any(double v, double* red /* [16] in LDS */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  const int nw = (int)blockDim.x >> 6;
  for (int w = 0; w < nw; ++w) t += red[w];
  return t;
}

__device__ __forceinline__ double block_sum_256(double v, double* red /* [4] in LDS */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// y[co] = b[co] + sum_ci x[ci] * A[ci][co]   (A row-major [C][C]; uniform addresses -> scalar loads)

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
  int np;               // number of partial P buffers (f16x3 kernels: one per pass over the hidden width), >= 1
  size_t pstride;       // floats between them
  const float* b3;      // [C] conv3 bias
  const float* A;       // post affine [C][C] or null: forward = NEXT step's ActNorm+1x1, inverse = this step's inverse 1x1+ActNorm
  const float* b;       // [C]
  float* out;           // element (q, co) at out[q*out_stride + out_off + co]
  int out_stride, out_off;
  double* logdet;       // [N] += sum log_s (forward only; may be null)
  float* log_s_out;     // optional [Q][C/2] dumps of the network outputs (glowk_coupling_net)
  float* t_out;
  float* o_save;        // saving pass (input-gradient / training): [Q][C/2] the pre-tanh log_s inputs (bias included) -- all the backward
                        // pass needs of the network's output; the per-tap outputs P themselves are scratch and are not kept
  int Q, h, w;
  int inverse;
  int* flag;            // sticky range flag of the handle: set to 1 when a network output is not finite (may be null)
};

// Second source of the range flag (the first is the split kernels' own range check, glowk_kernels.h: max3abs): a network
// output or a coupling result that is not finite -- an overflow of the LAST hidden layer (no ReLU follows it to swallow the
// NaN), a non-finite input tile, or a genuinely exploding flow.  The reference's callers assert on exactly this
// (run_basis_sep.py:183-191, train_glow.py:115-118).  Free here: these kernels wait for HBM.
__device__ __forceinline__ bool not_finite(float v) { return !(fabsf(v) <= 3.0e38f); }

// gather conv3 (9 taps), split + tanh (flow_tfk_layers.py:80-84), affine coupling
// (flow_tfp_bijectors.py:134-148), per-sample log-det (:150-153), then the following per-pixel affine
// QUAD: four lanes per pixel (small batches: the per-sample workgroups cannot fill the chip, so parallelism has to come from
// inside the pixel); otherwise one lane per pixel (large batches: fewer, fully used lanes -- 47 vs 80 us at 1024 tiles)
// one pixel of k_couple: gather the nine taps (QUAD: this lane's three, then the quad sum), bias, tanh / exp, coupling, the
// following affine, store.  Returns the pixel's sum of log_s on the lane that finished it (lane 0 of a quad), 0 elsewhere.
// LPP lanes per pixel: 1, 4 (lane r: taps r, r + 4, r + 8 of every partial) or 16 (the (tap, partial) pairs dealt round robin: the
// deep levels at small batches, where 64 pixels x 16 channels x 4 partials are ~200 dependent loads per lane of a 4-lane kernel)
template <int C, int LPP>
__device__ __forceinline__ float couple_pixel(const CoupleArgs& a, int q, int i, int j, bool live, int r4) {
  constexpr int CI = C / 2;
  constexpr bool QUAD = LPP == 4;
  float v[C], o[C];
#pragma unroll
  for (int c = 0; c < C; ++c) o[c] = 0.0f;
  if constexpr (LPP == 16) {
    for (int idx = r4; idx < 9 * a.np; idx += 16) {
      const int tap = idx / a.np, part = idx % a.np;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const int ii = i + dy, jj = j + dx;
      if (ii >= 0 && ii < a.h && jj >= 0 && jj < a.w) {
        const float* src = a.P + (size_t)part * a.pstride + (size_t)(tap * C) * a.Q + (q + dy * a.w + dx);
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] += src[(size_t)c * a.Q];
      }
    }
  } else {
#pragma unroll
  for (int u = 0; u < (QUAD ? 3 : 9); ++u) {
    const int tap = QUAD ? r4 + 4 * u : u;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const int ii = i + dy, jj = j + dx;
    if (tap < 9 && ii >= 0 && ii < a.h && jj >= 0 && jj < a.w) {
      const size_t off = (size_t)(tap * C) * a.Q + (q + dy * a.w + dx);
#pragma unroll
      for (int part = 0; part < 4; ++part)      // all partials' loads in flight together
        if (part < a.np) {
          const float* src = a.P + (size_t)part * a.pstride + off;
#pragma unroll
          for (int c = 0; c < C; ++c) o[c] += src[(size_t)c * a.Q];
        }
    }
  }
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
#pragma unroll
    for (int m = 1; m < LPP; m <<= 1) o[c] += __shfl_xor(o[c], m, 64);
    o[c] += a.b3[c];
  }
  if constexpr (LPP == 16) {
    // sixteen lanes per pixel: every lane holds the full o[] now (xor butterfly), so the per-pixel algebra is dealt out as well --
    // lane r owns channels r (and r + 16 at the 32-channel level): its own tanh / exp, an all-gather of the coupled vector over the
    // pixel's lanes, its own output channel(s) of the following affine, one coalesced store.  (On lane 0 alone this tail was ~700
    // dependent instructions per pixel at c = 16: 14.5 us per launch at 30 tiles.)
    constexpr int CPL = (C + 15) / 16;           // channels per lane
    float yo[CPL], ls_sum = 0.0f;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
      const int ch = r4 + 16 * e;
      float oc = 0.0f, ot = 0.0f;                // o[ch], o[CI + ch] by select chains (no dynamic register indexing)
#pragma unroll
      for (int c = 0; c < C; ++c) oc = (c == ch) ? o[c] : oc;
#pragma unroll
      for (int c = 0; c < CI; ++c) ot = (c == ch) ? o[CI + c] : ot;
      const bool mine = ch < C, first = ch < CI;
      const float vch = (a.vin && mine) ? a.vin[(size_t)q * C + ch] : 0.0f;
      float y = vch;
      if (first) {
        const float log_s = tanhf(oc);
        if (a.log_s_out && live) { a.log_s_out[(size_t)q * CI + ch] = log_s; a.t_out[(size_t)q * CI + ch] = ot; }
        if (a.o_save && live) a.o_save[(size_t)q * CI + ch] = oc;
        if (a.vin) {
          const float sc = expf(log_s);
          y = a.inverse ? (vch - ot) / sc : sc * vch + ot;
        }
        if (live) ls_sum += log_s;
      }
      bad |= mine && (not_finite(oc) | not_finite(y));
      yo[e] = y;
    }
    if (bad && live && a.flag) *a.flag = 1;
    if (a.out) {
      float z[CPL];
      const int lane0 = (int)(threadIdx.x & 63) & ~15;          // first lane of this pixel's group within the wave
      if (a.A) {
#pragma unroll
        for (int e = 0; e < CPL; ++e) z[e] = (r4 + 16 * e < C) ? a.b[r4 + 16 * e] : 0.0f;
#pragma unroll
        for (int ci = 0; ci < C; ++ci) {
          const float yc = __shfl(yo[ci / 16], lane0 + (ci & 15), 64);       // the all-gather: every lane of the group executes it
#pragma unroll
          for (int e = 0; e < CPL; ++e)
            if (r4 + 16 * e < C) z[e] = fmaf(yc, a.A[ci * C + r4 + 16 * e], z[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < CPL; ++e) z[e] = yo[e];
      }
      if (live) {
        float* dst = a.out + (size_t)q * a.out_stride + a.out_off;
#pragma unroll
        for (int e = 0; e < CPL; ++e)
          if (r4 + 16 * e < C) dst[r4 + 16 * e] = z[e];
      }
    }
    return ls_sum;
  }
  if (r4 != 0 || !live) return 0.0f;
  if (a.flag) {
    bool bad = false;
#pragma unroll
    for (int c = 0; c < C; ++c) bad |= not_finite(o[c]);
    if (bad) *a.flag = 1;
  }
  if (a.o_save) {
#pragma unroll
    for (int k = 0; k < CI; ++k) a.o_save[(size_t)q * CI + k] = o[k];
  }
  if (a.vin) {
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = a.vin[(size_t)q * C + c];
  }
  float y[C];
  float lsum = 0.0f;
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
  if (a.flag && a.vin) {
    bool bad = false;
#pragma unroll
    for (int c = 0; c < C; ++c) bad |= not_finite(y[c]);
    if (bad) *a.flag = 1;
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
  return lsum;
}

template <int C, bool QUAD>
__global__ __launch_bounds__(1024) void k_couple(CoupleArgs a) {   // 256 .. 1024 threads
  __shared__ double red[16];
  const int n = blockIdx.x;
  const int hw = a.h * a.w;
  float lsum = 0.0f;
  // four lanes per pixel: lane r of the quad gathers taps r, r + 4, r + 8 (all partial buffers), the quad adds up, lane 0
  // does the coupling.  (One lane per pixel left the deep levels -- 64 pixels per sample -- with 64 busy lanes per workgroup
  // and hundreds of dependent-address loads each.)
  const int r4 = QUAD ? (threadIdx.x & 3) : 0;
  const int qpb = QUAD ? (int)blockDim.x >> 2 : (int)blockDim.x;   // pixels per workgroup and iteration
  for (int pp0 = QUAD ? threadIdx.x >> 2 : threadIdx.x; pp0 < (hw + qpb - 1) / qpb * qpb; pp0 += qpb) {
    const bool live = pp0 < hw;
    const int pp = live ? pp0 : hw - 1;
    lsum += couple_pixel<C, QUAD ? 4 : 1>(a, n * hw + pp, pp / a.w, pp % a.w, live, r4);
  }
  if (a.logdet) {
    const double tot = block_sum_any((double)lsum, red);
    if (threadIdx.x == 0) a.logdet[n] += tot;
  }
}

// The same on a FLAT grid over the pixels (256 / LPP per workgroup: 64 at four lanes per pixel, 16 at sixteen) for batches whose
// per-sample workgroups would leave most CUs idle (the reference's 30 / 32 tiles): needs h w a multiple of the workgroup's pixels, so
// that they belong to one sample.  Its share of the sample's log-det goes to a slot of its own -- one slot per 16 pixels of a level:
// slot[n * sample_stride + slot_base + first 16-pixel unit of the workgroup]; no two workgroups add to one address, the order of the
// final sum (k_ld_fold) is fixed.
template <int C, int LPP>
__global__ __launch_bounds__(256) void k_couple_flat(CoupleArgs a, double* slot, int sample_stride, int slot_base) {
  __shared__ double red[4];
  constexpr int PPB = 256 / LPP;
  const int hw = a.h * a.w;
  const int q0 = (int)blockIdx.x * PPB + ((int)threadIdx.x / LPP);
  const bool live = q0 < a.Q;
  const int q = live ? q0 : a.Q - 1;
  const int pp = q % hw;
  const float lsum = couple_pixel<C, LPP>(a, q, pp / a.w, pp % a.w, live, threadIdx.x & (LPP - 1));
  if (slot) {
    const double tot = block_sum_256((double)lsum, red);
    const int first = (int)blockIdx.x * PPB;
    if (threadIdx.x == 0) slot[(size_t)(first / hw) * sample_stride + slot_base + (first % hw) / 16] = tot;
  }
}

// logdet[n] += the sample's slots, in slot order (k_couple_flat)
__global__ __launch_bounds__(64) void k_ld_fold(double* __restrict__ logdet, const double* __restrict__ slot, int nslots) {
  const int n = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < nslots; i += 64) acc += slot[(size_t)n * nslots + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if (threadIdx.x == 0) logdet[n] += acc;
}

// the rows the fused kernel could not finish (see fused_couple): one workgroup per sample.  Also adds the sample's log-det:
// the partial sums of the fused workgroups plus the edge pixels' own log_s.
struct EdgeArgs {
  const float* vin;      // [Q][4]
  const float* edge;     // [workgroup][4][FUSE_EW][4]
  const double* ldpart;
  const float *A, *b;
  float* out;
  int out_stride, out_off, inverse;
  double* logdet;        // [N] += (may be null)
  float* osave;          // saving pass: [Q][2] pre-tanh log_s inputs of the edge pixels (complete sums), or null
  int h, w;
  int* flag;
  int pxw;               // pixels per workgroup of the fused kernel that left `edge` / `ldpart`: 256 (k_net_h3s) or 128 (k_net_h3c)
};

__global__ __launch_bounds__(256) void k_couple_edge(EdgeArgs a) {
  __shared__ double red[4];
  const int n = blockIdx.x, w = a.w, hw = a.h * a.w;
  const int pxw = a.pxw;                                              // pixels per workgroup of the fused kernel (256, or 128: co-resident form)
  const int nb = hw >= pxw ? hw / pxw : 1, rows = pxw / w;
  float lsum = 0.0f;
  const int nedge = hw > pxw ? 2 * w * (nb - 1) : 0;
  for (int t = threadIdx.x; t < nedge; t += 256) {
    const int b = 1 + t / (2 * w), lower = (t / w) & 1, j = t % w;
    const size_t wgA = (size_t)n * nb + b - 1, wgB = wgA + 1;         // the workgroups above and below the boundary
    // upper row = last row of A: its partial sums (slot 1) + B's contribution upwards (slot 2); lower row = first row of B: slot 0 + A's slot 3
    const float4 part = *reinterpret_cast<const float4*>(a.edge + (((lower ? wgB : wgA) * 4 + (lower ? 0 : 1)) * FUSE_EW + j) * 4);
    const float4 halo = *reinterpret_cast<const float4*>(a.edge + (((lower ? wgA : wgB) * 4 + (lower ? 3 : 2)) * FUSE_EW + j) * 4);
    const size_t q = (size_t)n * hw + (size_t)(b * rows - 1 + lower) * w + j;
    const float o[4] = {part.x + halo.x, part.y + halo.y, part.z + halo.z, part.w + halo.w};
    const float4 v4 = *reinterpret_cast<const float4*>(a.vin + q * 4);
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
    if (a.osave) { a.osave[q * 2] = o[0]; a.osave[q * 2 + 1] = o[1]; }
    float y[4];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float log_s = tanhf(o[k]), sc = expf(log_s), tt = o[2 + k];
      y[k] = a.inverse ? (v[k] - tt) / sc : sc * v[k] + tt;
      y[2 + k] = v[2 + k];
      lsum += log_s;
      bad |= fz_not_finite(o[k]) | fz_not_finite(tt) | fz_not_finite(y[k]);
    }
    if (bad && a.flag) *a.flag = 1;
    float z[4];
    if (a.A) {
#pragma unroll
      for (int co = 0; co < 4; ++co) z[co] = a.b[co];
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int co = 0; co < 4; ++co) z[co] = fmaf(y[ci], a.A[ci * 4 + co], z[co]);
    } else {
#pragma unroll
      for (int co = 0; co < 4; ++co) z[co] = y[co];
    }
    *reinterpret_cast<float4*>(a.out + q * a.out_stride + a.out_off) = float4{z[0], z[1], z[2], z[3]};
  }
  if (a.logdet) {
    double d = (double)lsum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = red[0] + red[1] + red[2] + red[3];
      // the fused workgroups' per-wave partials (32 pixels each) of this sample, in pixel order
      const size_t i0 = (size_t)n * (hw / 32);
      for (int x = 0; x < hw / 32; ++x) tot += a.ldpart[i0 + x];
      a.logdet[n] += tot;
    }
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

// fp64 sum of n fp32 values in a fixed order (one workgroup of 1024 threads: thread t adds elements t, t + 1024, ...; then the
// waves, then the 16 wave sums): the summed log-likelihood the trainers reduce over (train_glow.py:29-31, 52-54).  Deterministic
// and independent of everything but n, so a shard's sum is bitwise repeatable; accumulate != 0 adds to *out (chunked batches);
// scale: the -1 / global_batch of tf.nn.compute_average_loss when the caller wants the loss itself.
__global__ __launch_bounds__(1024) void k_sum_f64(const float* __restrict__ v, size_t n, double* __restrict__ out, int accumulate, double scale) {
  __shared__ double red[16];
  double acc = 0.0;
  for (size_t e = threadIdx.x; e < n; e += 1024) acc += (double)v[e];
  const double tot = block_sum_any(acc, red);
  if (threadIdx.x == 0) *out = accumulate ? *out + scale * tot : scale * tot;
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

// ------------------------------------------------------------------------------------------------------------------
// input-gradient path (compute_grad_logprob, run_basis_sep.py:73-79; derivation SURVEY appendix A.5)
// ------------------------------------------------------------------------------------------------------------------
// one pixel's row of N floats (N = 2 or a multiple of 4) as 8- / 16-byte accesses; the callers guarantee the alignment (host check in
// launch_bwd_light: every [Q][C] array of the one-lane-per-pixel form starts on a 16-byte boundary)
template <int N>
__device__ __forceinline__ void ld_row(const float* __restrict__ p, float (&d)[N]) {
  if constexpr (N == 2) { const float2 t = *reinterpret_cast<const float2*>(p); d[0] = t.x; d[1] = t.y; }
  else {
#pragma unroll
    for (int e = 0; e < N / 4; ++e) {
      const float4 t = reinterpret_cast<const float4*>(p)[e];
      d[4 * e] = t.x; d[4 * e + 1] = t.y; d[4 * e + 2] = t.z; d[4 * e + 3] = t.w;
    }
  }
}
template <int N>
__device__ __forceinline__ void st_row(float* __restrict__ p, const float (&d)[N]) {
  if constexpr (N == 2) *reinterpret_cast<float2*>(p) = make_float2(d[0], d[1]);
  else {
#pragma unroll
    for (int e = 0; e < N / 4; ++e) reinterpret_cast<float4*>(p)[e] = make_float4(d[4 * e], d[4 * e + 1], d[4 * e + 2], d[4 * e + 3]);
  }
}

struct BwdArgs {
  // (1) gradient wrt the tensor v_s entering step s's coupling:
  //     merge: g_v = [g_va, g_yb + sum_tap Pg[tap, cin][q - d(tap)]]   (ghalf_in, Pg of step s), or
  //     direct: g_v = gv_direct[q*gvd_stride + gvd_off + .]             (Pg == null)
  const float* ghalf_in;
  const float* Pg;
  int npg;                 // number of partial Pg buffers (f16x3 backward kernel: one per pass), >= 1
  size_t pgstride;
  const float* gv_direct;
  int gvd_stride, gvd_off;
  // (2) through step s's fused ActNorm + 1x1: g_y = g_v . A^T (null: g_y = g_v)
  const float* A;
  // (3) coupling backward of the step that produced y (forward order: the step before s), or none (v == null)
  const float* v;          // [Q][C] saved coupling input
  const float* osave;      // [Q][C/2] saved pre-tanh log_s inputs of that coupling (CoupleArgs::o_save)
  float* g_o;              // [Q][C] gradient wrt the network output o = [pre-tanh log_s, t]
  float* ghalf_out;        // [Q][C] [g_va, g_yb]
  float* gu_out;           // [Q][C] g_y itself when no coupling follows
  int Q, h, w;
  int* flag;               // sticky range flag (see CoupleArgs); may be null
  float* gv_out;           // training: [Q][C] the merged g_v of (1) (the weight gradients of step s's ActNorm + 1x1 need it), or null
  // training in the split arithmetic: the backward network and the weight-gradient GEMMs are linear in g_o, so g_o is written
  // times a power of two `go_scale` chosen by the host from the gradient magnitudes of the previous sweep (dynamic scaling: it
  // keeps every value the split kernels and GEMMs convert to fp16 inside the static range bound whatever the loss scale is),
  // the per-tap network gradients Pg that come back carry it and are multiplied by `pg_scale` = 1 / go_scale here, and the
  // assembly of the weight gradients divides by it.  gmax (optional): atomic maximum of |g_o| (unscaled, float bits) per launch.
  float go_scale, pg_scale;
  unsigned* gmax;          // [16] words per level (the workgroups spread their atomics over them)
};

// LPP lanes per pixel: 1 (large grids: every load of the planar Pg arrays and every 16-byte row access is fully coalesced),
// 4 (lane r gathers taps r, r + 4, r + 8 of every partial) or 16 -- the (tap, partial) pairs dealt round
// robin -- for the deep levels at small batches: 64 pixels x 8 or 16 channels per sample leave a 4-lane kernel with ~200 dependent
// loads per lane on a few dozen workgroups (15-16 us per launch at 30 tiles against a ~5 us floor)
template <int C, int LPP>
__global__ __launch_bounds__(256) void k_bwd_light(BwdArgs a) {
  constexpr int CI = C / 2;
  constexpr int PPB = 256 / LPP;     // pixels per workgroup
  const int hw = a.h * a.w;
  // Flat grid over the Q pixels (nothing is reduced per sample); lane 0 of a pixel's group does the per-pixel algebra.
  const int r4 = threadIdx.x & (LPP - 1);
  {
    const int q0 = (int)blockIdx.x * PPB + ((int)threadIdx.x / LPP);
    const bool live = q0 < a.Q;
    const int q = live ? q0 : a.Q - 1;
    const int pp = q % hw;
    const int i = pp / a.w, j = pp % a.w;
    float gsum[CI], o[CI];   // merged network gradient (second half of g_v); log_s half of the saved network output
#pragma unroll
    for (int c = 0; c < CI; ++c) { gsum[c] = 0.0f; o[c] = 0.0f; }
    if constexpr (LPP <= 4) {
#pragma unroll
    for (int u = 0; u < (9 + LPP - 1) / LPP; ++u) {
      const int tap = r4 + LPP * u;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      if (tap < 9 && a.Pg) {
        const int ii = i - dy, jj = j - dx;   // Pg[q'] contributes at q' + d(tap)
        if (ii >= 0 && ii < a.h && jj >= 0 && jj < a.w) {
          const size_t po = (size_t)(tap * CI) * a.Q + (q - dy * a.w - dx);
#pragma unroll
          for (int part = 0; part < 4; ++part)
            if (part < a.npg) {
#pragma unroll
              for (int c = 0; c < CI; ++c) gsum[c] += a.Pg[(size_t)part * a.pgstride + po + (size_t)c * a.Q];
            }
        }
      }
    }
    } else {
      // (tap, partial) pairs dealt round robin over the pixel's LPP lanes: 9 * npg pairs
      const int npm = a.npg;
      for (int idx = r4; idx < 9 * npm; idx += LPP) {
        const int tap = idx / npm, part = idx % npm;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        if (a.Pg && part < a.npg) {
          const int ii = i - dy, jj = j - dx;
          if (ii >= 0 && ii < a.h && jj >= 0 && jj < a.w) {
            const float* src = a.Pg + (size_t)part * a.pgstride + (size_t)(tap * CI) * a.Q + (q - dy * a.w - dx);
#pragma unroll
            for (int c = 0; c < CI; ++c) gsum[c] += src[(size_t)c * a.Q];
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CI; ++c) {
#pragma unroll
      for (int m = 1; m < LPP; m <<= 1) gsum[c] += __shfl_xor(gsum[c], m, 64);
    }
    float gomax = 0.0f;      // largest |g_o| this lane writes (unscaled)
    if constexpr (LPP == 16) {
      // sixteen lanes per pixel: the per-pixel algebra is dealt out too -- lane r owns channel r (r, r + 16 at the 32-channel level):
      // its entry of the merged g_v, an all-gather over the pixel's lanes, its own row of g_y = g_v . A^T, and for r < c/2 one tanh /
      // exp of the coupling backward (on lane 0 alone: ~600 dependent instructions per pixel at c = 16)
      constexpr int CPL = (C + 15) / 16;
      const int lane0 = (int)(threadIdx.x & 63) & ~15;
      float gvo[CPL];
      bool bad = false;
#pragma unroll
      for (int e = 0; e < CPL; ++e) {
        const int ch = r4 + 16 * e;
        float g = 0.0f;
        if (ch < C) {
          if (a.Pg) {
            float gs = 0.0f;                       // gsum[ch - CI] by a select chain
#pragma unroll
            for (int c = 0; c < CI; ++c) gs = (c == ch - CI) ? gsum[c] : gs;
            g = a.ghalf_in[(size_t)q * C + ch] + (ch >= CI ? gs * a.pg_scale : 0.0f);
            bad |= ch >= CI && not_finite(gs);
          } else {
            g = a.gv_direct[(size_t)q * a.gvd_stride + a.gvd_off + ch];
          }
          if (a.gv_out && live) a.gv_out[(size_t)q * C + ch] = g;
        }
        gvo[e] = g;
      }
      float gyo[CPL];
      if (a.A) {
#pragma unroll
        for (int e = 0; e < CPL; ++e) gyo[e] = 0.0f;
#pragma unroll
        for (int co = 0; co < C; ++co) {
          const float gc = __shfl(gvo[co / 16], lane0 + (co & 15), 64);
#pragma unroll
          for (int e = 0; e < CPL; ++e)
            if (r4 + 16 * e < C) gyo[e] = fmaf(gc, a.A[(r4 + 16 * e) * C + co], gyo[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < CPL; ++e) gyo[e] = gvo[e];
      }
      if (a.v) {
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
          const int ch = r4 + 16 * e;
          if (ch < CI) {
            const float ok_ = a.osave[(size_t)q * CI + ch];
            const float log_s = tanhf(ok_), sc = expf(log_s), va = a.v[(size_t)q * C + ch], gya = gyo[e];
            const float g_pre = (gya * sc * va + 1.0f) * (1.0f - log_s * log_s);
            bad |= not_finite(ok_);
            if (live) {
              gomax = nan_max(gomax, nan_max(fabsf(g_pre), fabsf(gya)));   // (NaN-keeping: fmaxf would drop it)
              a.g_o[(size_t)q * C + ch] = g_pre * a.go_scale;
              a.g_o[(size_t)q * C + CI + ch] = gya * a.go_scale;
              a.ghalf_out[(size_t)q * C + ch] = gya * sc;
            }
          } else if (ch < C && live) {
            a.ghalf_out[(size_t)q * C + ch] = gyo[e];
          }
        }
      } else if (live) {
#pragma unroll
        for (int e = 0; e < CPL; ++e)
          if (r4 + 16 * e < C) a.gu_out[(size_t)q * C + r4 + 16 * e] = gyo[e];
      }
      if (bad && live && a.flag) *a.flag = 1;
    } else
    if (r4 == 0 && live) {
#pragma unroll
    for (int c = 0; c < CI; ++c) gsum[c] *= a.pg_scale;
    if (a.flag) {
      bool bad = false;
#pragma unroll
      for (int c = 0; c < CI; ++c) bad |= not_finite(gsum[c]) | not_finite(o[c]);
      if (bad) *a.flag = 1;
    }
    float gv[C];
    if (a.Pg) {
      if constexpr (LPP == 1) ld_row<C>(a.ghalf_in + (size_t)q * C, gv);
      else {
#pragma unroll
        for (int c = 0; c < C; ++c) gv[c] = a.ghalf_in[(size_t)q * C + c];
      }
#pragma unroll
      for (int c = 0; c < CI; ++c) gv[CI + c] += gsum[c];
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) gv[c] = a.gv_direct[(size_t)q * a.gvd_stride + a.gvd_off + c];
    }
    if (a.gv_out) {
      if constexpr (LPP == 1) st_row<C>(a.gv_out + (size_t)q * C, gv);
      else {
#pragma unroll
        for (int c = 0; c < C; ++c) a.gv_out[(size_t)q * C + c] = gv[c];
      }
    }
    float gy[C];
    if (a.A) {
#pragma unroll
      for (int ci = 0; ci < C; ++ci) {
        float acc = 0.0f;
#pragma unroll
        for (int co = 0; co < C; ++co) acc = fmaf(gv[co], a.A[ci * C + co], acc);
        gy[ci] = acc;
      }
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) gy[c] = gv[c];
    }
    if (a.v) {
      float va_[CI], go_[C], gh_[C];
      if constexpr (LPP == 1) { ld_row<CI>(a.osave + (size_t)q * CI, o); ld_row<CI>(a.v + (size_t)q * C, va_); }
      else {
#pragma unroll
        for (int c = 0; c < CI; ++c) { o[c] = a.osave[(size_t)q * CI + c]; va_[c] = a.v[(size_t)q * C + c]; }   // only the log_s half of the network output is needed
      }
#pragma unroll
      for (int k = 0; k < CI; ++k) {
        const float log_s = tanhf(o[k]);
        const float sc = expf(log_s);
        const float va = va_[k];
        const float gya = gy[k];
        const float g_ls = gya * sc * va + 1.0f;          // + 1: d(sum log_s)/d log_s (flow_tfp_bijectors.py:150-153)
        const float g_pre = g_ls * (1.0f - log_s * log_s); // through tanh
        gomax = nan_max(gomax, nan_max(fabsf(g_pre), fabsf(gya)));   // (NaN-keeping: fmaxf would drop it)
        go_[k] = g_pre * a.go_scale;
        go_[CI + k] = gya * a.go_scale;                   // g_t
        gh_[k] = gya * sc;                                // g_va
        gh_[CI + k] = gy[CI + k];                         // g_yb (the network's contribution is merged by the next call)
      }
      if constexpr (LPP == 1) { st_row<C>(a.g_o + (size_t)q * C, go_); st_row<C>(a.ghalf_out + (size_t)q * C, gh_); }
      else {
        float* go = a.g_o + (size_t)q * C;
        float* gh = a.ghalf_out + (size_t)q * C;
#pragma unroll
        for (int c = 0; c < C; ++c) { go[c] = go_[c]; gh[c] = gh_[c]; }
      }
    } else {
      if constexpr (LPP == 1) st_row<C>(a.gu_out + (size_t)q * C, gy);
      else {
        float* gu = a.gu_out + (size_t)q * C;
#pragma unroll
        for (int c = 0; c < C; ++c) gu[c] = gy[c];
      }
    }
    }
    if (a.gmax) {            // (uniform branch; every lane is here: idle ones carry 0)
      // workgroup maximum first, then ONE atomic per workgroup, spread over 16 words (thousands of waves hammering one address
      // serialise in the L2: the launch took 3x as long)
      __shared__ float wmax[4];
      if (!(gomax == gomax)) gomax = __uint_as_float(0x7f800000u);     // NaN counts as "not finite"
#pragma unroll
      for (int o2 = 32; o2 > 0; o2 >>= 1) gomax = fmaxf(gomax, __shfl_xor(gomax, o2, 64));
      if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = gomax;
      __syncthreads();
      if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (m > 0.0f) atomicMax(a.gmax + (blockIdx.x & 15), __float_as_uint(m));
      }
    }
  }
}

// gradient of the prior log-density wrt the latent (flow_builder.py:131-144): -(z - loc)/sigma^2 or -z
__global__ __launch_bounds__(256) void k_prior_grad(const float* __restrict__ z, size_t total, int E, const float* __restrict__ loc,
                                                   const float* __restrict__ log_scale, float* __restrict__ gz) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int e = (int)(idx % E);
  const float zv = z[idx];
  gz[idx] = loc ? -(zv - loc[e]) * expf(-2.0f * log_scale[e]) : -zv;
}

// backward of k_split: g_o [N,h,w,C] = concat(reshape(gz slice), unsqueeze(g_unext))
template <int C>
__global__ __launch_bounds__(256) void k_bwd_split(const float* __restrict__ gz, int HWl, int Cl, int off, int Cz,
                                                  const float* __restrict__ gun, int h, int w, float* __restrict__ go) {
  constexpr int CH = C / 2;
  constexpr int C2 = 2 * C;
  const int n = blockIdx.x, hw = h * w;
  float* gn = go + (size_t)n * hw * C;
  for (int e = threadIdx.x; e < hw * CH; e += 256) {
    const int pp = e / CH, k = e % CH;
    gn[(size_t)pp * C + k] = gz[((size_t)n * HWl + e / Cz) * Cl + off + e % Cz];
  }
  const int h2 = h / 2, w2 = w / 2;
  const float* un = gun + (size_t)n * h2 * w2 * C2;
  for (int e = threadIdx.x; e < h2 * w2 * C2; e += 256) {
    const int pp = e / C2, cc = e % C2;
    const int i = pp / w2, j = pp % w2;
    const int cin = cc >> 2, da = (cc >> 1) & 1, db = cc & 1;
    gn[(size_t)((2 * i + da) * w + 2 * j + db) * C + CH + cin] = un[e];
  }
}

// backward of k_in: g_u [N,h,w,C] -> unsqueeze -> through SpecPreprocessing (incl. its own log-det term) -> g_x [N,2h,2w,C/4]
template <int C>
__global__ __launch_bounds__(256) void k_bwd_in(const float* __restrict__ gu, const float* __restrict__ x, int h, int w, PreArgs pre,
                                               float* __restrict__ gx) {
  constexpr int Cin = C / 4;
  const int n = blockIdx.x, H = 2 * h, W = 2 * w;
  const float inv = 1.0f / (pre.maxval - pre.minval);
  for (int pp = threadIdx.x; pp < h * w; pp += 256) {
    const int i = pp / w, j = pp % w;
    const float* src = gu + ((size_t)n * h * w + pp) * C;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
      const int cin = cc >> 2, da = (cc >> 1) & 1, db = cc & 1;
      const size_t xi = ((size_t)(n * H + 2 * i + da) * W + (2 * j + db)) * Cin + cin;
      float g = src[cc];
      if (pre.use_logit) {
        // y = log p - log(1-p), ld = -log p - log(1-p), p = (1-2a) u + a, u = (x-min)/(max-min)
        const float u = (x[xi] - pre.minval) * inv;
        const float p = (1.0f - 2.0f * pre.alpha) * u + pre.alpha;
        const float dp = (1.0f - 2.0f * pre.alpha) * inv;
        g = g * dp * (1.0f / p + 1.0f / (1.0f - p)) + dp * (-1.0f / p + 1.0f / (1.0f - p));
      } else {
        g = g * inv;
      }
      gx[xi] = g;
    }
  }
}
