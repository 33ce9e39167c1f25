/* glowk -- C ABI of the MI355X (gfx950) Glow forward / inverse / log-prob engine.
 *
 * The reference (SamArgt/AudioSourceSep) has no native boundary: its "operator API" for this path is
 * the Python duck type of tfd.TransformedDistribution / tfb.Bijector returned by
 * flow_models/flow_builder.py:60-146 (build_glow).  Each entry point below names the reference
 * interface it replaces; audiosourcesep_amd/flow_models/ binds them with ctypes and re-creates that
 * duck type (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every tensor is NHWC, float32, contiguous; "dev" pointers are device (HBM) addresses owned by the
 *     caller (e.g. torch storages), "host" pointers are ordinary host memory;
 *   - every compute call is asynchronous on the hipStream_t passed as `stream` (void*, 0 = null stream);
 *   - return value: 0 = OK, non-zero = error (enum glowk_status); glowk_last_error() returns a thread-local message;
 *   - one handle per device; a handle is not thread safe; the engine owns packed weights + workspace;
 *   - every entry point that touches the GPU selects the handle's device for the duration of the call and restores the
 *     caller's current device before returning.
 *   - `level` counts blocks from 0 (glowBlock1 = 0); `step` is the creation index k of glowStep_k
 *     (flow_glow.py:44-49).  The forward pass applies steps K-1 ... 0 (tfb.Chain order, :51-52).
 */
#ifndef GLOWK_H
#define GLOWK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLOWK_VERSION 400

/* Arguments of build_glow (flow_builder.py:60-61) + SpecPreprocessing kwargs (flow_tfp_bijectors.py:365). */
typedef struct glowk_config {
  int32_t H, W, C;     /* data_shape */
  int32_t L, K, F;     /* L in {2,3,4}; K steps per block; F = n_filters in {128, 256, 384, 512} (glowk_create rejects others) */
  int32_t learntop;    /* 1: learnable diagonal Gaussian prior (flow_builder.py:131-141), 0: N(0,1) (:142-144) */
  int32_t use_logit;   /* SpecPreprocessing(use_logit=...) */
  float minval, maxval, alpha;
  float bn_eps;        /* Keras BatchNormalization epsilon (1e-3) */
} glowk_config;

typedef struct glowk_handle glowk_handle;

/* Tensors of one flow step in the reference's own layouts (creation order of SURVEY appendix A.3). */
enum glowk_tensor_id {
  GLOWK_ACTNORM_LOG_SCALE = 0, /* [c]          ActNorm.log_scale   flow_tfp_bijectors.py:236 */
  GLOWK_ACTNORM_SHIFT = 1,     /* [c]          ActNorm.shift       :239 */
  GLOWK_INV1X1_P = 2,          /* [c,c]        Invertible1x1Conv.P :281 */
  GLOWK_INV1X1_SIGN_S = 3,     /* [c]          sign_S              :285 */
  GLOWK_INV1X1_L = 4,          /* [c,c]        L                   :289 */
  GLOWK_INV1X1_LOG_S = 5,      /* [c]          log_S               :291 */
  GLOWK_INV1X1_U = 6,          /* [c,c]        U                   :293 */
  GLOWK_CONV1_KERNEL = 7,      /* [3,3,c/2,F]  HWIO                flow_tfk_layers.py:56-60 */
  GLOWK_CONV1_BIAS = 8,        /* [F] */
  GLOWK_BN1_GAMMA = 9,         /* [F]          batch_norm_1        :61 */
  GLOWK_BN1_BETA = 10,
  GLOWK_BN1_MEAN = 11,
  GLOWK_BN1_VAR = 12,
  GLOWK_CONV2_KERNEL = 13,     /* [1,1,F,F]                        :63-65 */
  GLOWK_CONV2_BIAS = 14,       /* [F] */
  GLOWK_BN2_GAMMA = 15,        /* [F]          batch_norm_2        :66 */
  GLOWK_BN2_BETA = 16,
  GLOWK_BN2_MEAN = 17,
  GLOWK_BN2_VAR = 18,
  GLOWK_CONV3_KERNEL = 19,     /* [3,3,F,c]                        :68-70 */
  GLOWK_CONV3_BIAS = 20,       /* [c] */
  GLOWK_INV1X1_P_INV = 21,     /* [c,c]        P_inv, the stored inverse of P that _inverse multiplies by  :282-284,313.  Left unset
                                  (all zeros) the engine uses inv(P), which is what the reference initialises it to. */
  GLOWK_NUM_STEP_TENSORS = 22,
  /* prior (level = -1, step = 0), flow_builder.py:131-139 */
  GLOWK_PRIOR_LOC = 100,       /* [H/2^L, W/2^L, C*4^L] */
  GLOWK_PRIOR_LOG_SCALE = 101  /* same shape: log of scale_diag (TransformedVariable(.., Exp())) */
};

/* Arithmetic of the coupling-network contractions (the >95 % of the FLOPs), for every compute entry point of the handle
 * (forward, inverse, log_prob, log_prob_grad, param_grad, sample, the per-step calls).  Shapes without a split-kernel instance
 * run the exact kernels in every mode (at the supported widths: the 32-channel level of 4-level models at n_filters 128;
 * glowk_kernel_families tells which family every launch took). */
enum glowk_precision {
  GLOWK_PREC_F32 = 0,     /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (default) */
  GLOWK_PREC_F16X3 = 1,   /* error-compensated split: x = hi + lo in fp16, 3 fp16 MFMAs per product, fp32 accumulate:
                             fp32-class results (~5e-8 relative on log_prob) at ~3x the speed; assumes hidden activations
                             below 16 376 in magnitude (any normalised flow) */
  GLOWK_PREC_F16X2 = 2    /* throughput mode of the plain forward direction (forward, log_prob, inverse, sample): weights
                             hi + lo, activations rounded to fp16 once, 2 fp16 MFMAs per product -- log_prob ~1e-5 relative
                             (inside the 1e-4 bar, no longer fp32-class); log_prob_grad runs the F16X3 kernels in this mode,
                             and so do shapes without a two-term instance */
};

enum glowk_status {
  GLOWK_OK = 0,
  GLOWK_ERR = 1,         /* bad argument / unsupported shape / HIP error: see glowk_last_error() */
  GLOWK_ERR_RANGE = 2    /* a split-precision call left the fp16 range (see glowk_range_policy); outputs are not usable */
};

/* Range guard of the split arithmetics (F16X3 / F16X2).  Every value that is split into fp16 hi + lo must stay below
 * 65504 / 4 in magnitude; beyond that hi = inf, lo = -inf, the next contraction is NaN -- and the ReLU after it turns that NaN
 * into 0: a finite, WRONG result that no test of the outputs can see.  The guard is therefore STATIC: glowk_finalize_weights
 * (and the device-side refresh after glowk_apply_gradients) bounds every split value of a flow step by the L1 norms of its
 * BatchNorm-folded weights as a linear function of the largest coupling-network INPUT, and solves for the input magnitude
 * `xlim` up to which no split can overflow; the network kernels compare the inputs they gather against it (prologue, once per
 * pixel; nothing in the MFMA loops) and raise a sticky per-handle flag on the device.  It is a worst-case bound: it may send a
 * legitimate call to the exact kernels, it cannot miss an overflow.  Second, free source of the same flag: a non-finite network
 * output, coupling result or gradient seen by the coupling / gradient kernels (non-finite input tiles, a genuinely exploding
 * flow -- what the reference's callers assert on, run_basis_sep.py:183-191, train_glow.py:115-118).
 * What a compute call (forward, inverse, log_prob, log_prob_sum, log_prob_grad, param_grad, sample, step_*, coupling_net) does
 * with the flag:
 *   GLOWK_RANGE_ERROR (default)  after its launches the call waits for the stream, reads the flag and, if set, clears it and
 *                                returns GLOWK_ERR_RANGE: the outputs of the call are not usable;
 *   GLOWK_RANGE_FALLBACK         same check, but the call is re-run inside the engine on the exact fp32 kernels and returns
 *                                that result (0); glowk_range_status counts the re-runs;
 *   GLOWK_RANGE_IGNORE           no wait, no read: calls stay fully asynchronous (hipGraph capture); the caller polls
 *                                glowk_range_status.
 * Calls in GLOWK_PREC_F32 never check: there a non-finite result is the reference's own result. */
enum glowk_range_policy { GLOWK_RANGE_IGNORE = 0, GLOWK_RANGE_ERROR = 1, GLOWK_RANGE_FALLBACK = 2 };

int glowk_version(void);
const char* glowk_last_error(void);
/* Diagnostic switches (environment variables GLOWK_HALF_OFF, GLOWK_NO_FUSE, GLOWK_BWD_LIGHT_4, ...: one launch form forced for an
 * A/B timing or a form-against-form parity test; none is needed for normal use) are read when the library is loaded, not per
 * launch; a process that changes its environment afterwards calls this to have them read again.  (No reference counterpart.) */
void glowk_reload_env(void);
/* Diagnostic, builds with -DGLOWK_STAMPS only (the product kernels carry no stamps: the buffer then stays zero): the first call arms
 * in-kernel time stamps (later coupling-network launches of the instrumented kernels leave the
 * constant-rate 100 MHz counter of workgroup (0, 0)'s first lane at their phase boundaries); every call waits for the device and
 * copies the n <= 64 stamp words of the LAST such launch.  Current device.  (No reference counterpart.) */
int glowk_debug_stamps(unsigned long long* out, int n);

/* --- construction: replaces build_glow (flow_builder.py:60-146) --------------------------------- */
int glowk_create(const glowk_config* cfg, int device, glowk_handle** out);
int glowk_destroy(glowk_handle* h);
/* number of elements tensor `id` of (level, step) holds, or 0 if the id/level is invalid */
size_t glowk_tensor_size(const glowk_handle* h, int level, int tensor_id);
/* copy one tensor host -> engine / engine -> host: replaces assigning / reading flow.variables
 * (train_utils.py:67-68 uses them as the checkpoint root) */
int glowk_set_tensor(glowk_handle* h, int level, int step, int tensor_id, const float* host, size_t n);
int glowk_get_tensor(const glowk_handle* h, int level, int step, int tensor_id, float* host, size_t n);
/* assemble W = P L U and W^-1 (flow_tfp_bijectors.py:300-303,309-315), fold ActNorm into the 1x1,
 * fold BN into per-channel affines, pack the conv kernels into MFMA operand order, upload. Must be
 * called after the last glowk_set_tensor and before any compute call. Synchronous. */
int glowk_finalize_weights(glowk_handle* h);
/* ActNorm data-dependent initialisation (flow_tfp_bijectors.py:222-234) as build_glow drives it: the
 * minibatch x [N,H,W,C] is preprocessed, squeezed and pushed through the steps one by one; before each
 * step its ActNorm log_scale/shift are set from the per-channel mean / population std (+1e-8) of the
 * tensor reaching it (GlowBlock.__init__, flow_glow.py:40-49).  All other tensors must be set and
 * glowk_finalize_weights called first; the new ActNorm tensors can be read back with glowk_get_tensor.
 *   runtime_order = 0: steps are visited in creation order 0..K-1 exactly like the reference constructor
 *                      (although tfb.Chain later applies them K-1..0);
 *   runtime_order = 1: steps are visited in the order the forward pass applies them (K-1..0), which keeps
 *                      every step's input normalised at run time (used for the synthetic benchmark weights);
 *   raw_minibatch_quirk = 1: blocks 2+ of the 3- and 4-level graphs are initialised from the RAW preprocessed
 *                      minibatch reinterpreted by Squeeze's reshape, as GlowBijector_3blocks/_4blocks do
 *                      (flow_glow.py:162-165,171-174); 0: from the propagated second half.
 * Synchronous (reads statistics back per step). */
int glowk_actnorm_data_init(glowk_handle* h, const float* x_dev, int N, int runtime_order, int raw_minibatch_quirk, void* stream);
int glowk_set_precision(glowk_handle* h, int precision);
int glowk_get_precision(const glowk_handle* h);
int glowk_set_range_policy(glowk_handle* h, int policy);
int glowk_get_range_policy(const glowk_handle* h);
/* waits for `stream`, reports whether the sticky range flag is set (and clears it) and how many calls were re-run on the
 * fp32 kernels so far; either output may be NULL */
int glowk_range_status(glowk_handle* h, int* tripped, int64_t* fallbacks, void* stream);
/* The margin of the static bound, measured: between glowk_range_probe_begin and glowk_range_probe_end every split-kernel launch
 * of the handle also records the largest coupling-network input it gathered (one atomic per wave, in the kernels' prologue);
 * glowk_range_probe_end waits for `stream` and returns, over all those launches, the largest  input / limit  of the forward
 * networks -- below 1 the guard did not fire, 0 means no split launch ran -- and, for the backward (gradient) networks, which
 * are linear and normalise every pixel's gradient vector by a power of two before the split (so that no gradient magnitude can
 * leave the range), the static ratio (smallest usable normalisation) / (what the weights' worst-case bound allows).  A
 * diagnostic (bench.py reports it for the trained BASIS priors); off by default. */
int glowk_range_probe_begin(glowk_handle* h);
int glowk_range_probe_end(glowk_handle* h, float* fwd_ratio, float* bwd_ratio, void* stream);
/* device memory (bytes) the engine allocates for batches up to N: the forward/inverse workspace, plus -- with_grad != 0 --
 * the per-step saves and gradient scratch of glowk_log_prob_grad in the handle's current precision.  glowk_reserve allocates
 * exactly that up front, so that no later compute call of that kind with that batch size (or a smaller one in the same
 * launch regime) allocates or synchronises the device (required before hipGraph capture).  Buffers only ever grow. */
size_t glowk_workspace_bytes(const glowk_handle* h, int N, int with_grad);
int glowk_reserve(glowk_handle* h, int N, int with_grad);
/* the largest batch ONE call accepts (2^28 elements / (H*W*C): indices within a call are 32-bit); every batch entry point
 * rejects a larger N with GLOWK_ERR.  Tiles are independent, so a caller with more tiles loops over chunks
 * (audiosourcesep_amd/engine.py does). */
int glowk_max_tiles(const glowk_handle* h);

/* --- the hot path ------------------------------------------------------------------------------- */
/* Chain([glow, prepro]).forward(x) and its forward_log_det_jacobian (flow_builder.py:127;
 * flow_glow.py:102-108,119-126 / 176-185,198-209 / 268-282,298-313): x [N,H,W,C] -> z [N,Hl,Wl,Cl],
 * logdet [N] (may be NULL). */
int glowk_forward(glowk_handle* h, const float* x_dev, int N, float* z_dev, float* logdet_dev, void* stream);
/* Chain.inverse(z): z -> x (flow_glow.py:110-117 / 187-196 / 284-296) */
int glowk_inverse(glowk_handle* h, const float* z_dev, int N, float* x_dev, void* stream);
/* TransformedDistribution.log_prob(x) (train_glow.py:30, run_basis_sep.py:77): logp [N];
 * z_dev may be NULL or receives the latent */
int glowk_log_prob(glowk_handle* h, const float* x_dev, int N, float* logp_dev, float* z_dev, void* stream);
/* glowk_log_prob that also leaves sum_n logp[n] on the device as ONE fp64 value (accumulate != 0: added to *sum_dev -- batches
 * evaluated in chunks): the summed log-likelihood of train_glow.py:29-31 / :52-54, which is what the ranks all-reduce (one
 * element, RCCL).  The sum is taken in a fixed order (k_sum_f64), so it is bitwise repeatable for a given shard. */
int glowk_log_prob_sum(glowk_handle* h, const float* x_dev, int N, float* logp_dev, float* z_dev, double* sum_dev, int accumulate,
                       void* stream);
/* the reduction alone: *out_dev = (accumulate ? *out_dev : 0) + scale * sum of n floats, fp64, fixed order; no handle needed
 * (scale = -1 / global batch gives the replica's share of tf.nn.compute_average_loss, train_glow.py:29-31) */
int glowk_sum_f64(const float* v_dev, size_t n, double* out_dev, int accumulate, double scale, void* stream);
/* compute_grad_logprob (run_basis_sep.py:73-79): logp [N] and d sum(logp) / dx [N,H,W,C] */
int glowk_log_prob_grad(glowk_handle* h, const float* x_dev, int N, float* logp_dev, float* dx_dev, void* stream);
/* TransformedDistribution.sample(n) (train_glow.py:74) with the standard-normal draw supplied by the
 * caller: eps [N,Hl,Wl,Cl] -> x = chain.inverse(loc + exp(log_scale) * eps) */
int glowk_sample(glowk_handle* h, const float* eps_dev, int N, float* x_dev, void* stream);
/* prior.log_prob(z) alone: [N] */
int glowk_prior_log_prob(glowk_handle* h, const float* z_dev, int N, float* logp_dev, void* stream);

/* --- measurement ---------------------------------------------------------------------------------- */
/* Per-kernel HIP-event timing of the coupling-network kernel (k_net), recorded on the stream each launch
 * goes to.  glowk_profile_begin arms it, every later compute call brackets its k_net launches with
 * events, glowk_profile_end synchronises, sums the elapsed times per level and disarms.  Used by bench.py
 * for the roofline object; off by default (no events, no overhead). */
typedef struct glowk_profile {
  double net_ms[4];          /* summed k_net duration per level */
  int64_t net_launches[4];
} glowk_profile;
int glowk_profile_begin(glowk_handle* h);
/* number of flow steps so far that ran as ONE kernel -- coupling network + affine coupling + next step's ActNorm / 1x1 fused
 * (flow_glow.py:21-22 as a single launch) -- rather than as network kernel + coupling kernel */
int64_t glowk_fused_steps(const glowk_handle* h);
/* coupling-network launches of the handle so far, by kernel family: out7[0] the exact fp32 kernel (k_net_f32), [1] the split
 * kernel on v_mfma_f32_32x32x16_f16 (k_net_h3), [2] on 16x16x32 (k_net_h3s), [3] its 128-pixel half-wave form, [4] the fused
 * network + coupling kernel, [5] of these ([2] or [4]) the launches that took the co-resident form (k_net_h3c: four-wave / 128-pixel
 * workgroups, two to a CU), [6] of [3] the launches that took the small-grid form with all conv1 blocks first (k_net_h3q).  A handle
 * in a split arithmetic whose out7[0] stays put ran no level on the exact kernels. */
int glowk_kernel_families(const glowk_handle* h, int64_t* out7);
int glowk_profile_end(glowk_handle* h, glowk_profile* out);

/* --- sub-bijectors, as exercised one by one by unittest_flow_models.py:124-186 --------------------- */
/* Squeeze._forward / _inverse (flow_tfp_bijectors.py:170-180); no handle needed */
int glowk_squeeze(const float* x_dev, int N, int H, int W, int C, float* y_dev, void* stream);
int glowk_unsqueeze(const float* y_dev, int N, int h, int w, int c4, float* x_dev, void* stream);
/* SpecPreprocessing forward / inverse / fldj (flow_tfp_bijectors.py:372-396) */
int glowk_preprocess_forward(glowk_handle* h, const float* x_dev, int N, float* y_dev, float* logdet_dev, void* stream);
int glowk_preprocess_inverse(glowk_handle* h, const float* y_dev, int N, float* x_dev, void* stream);
/* GlowStep forward (+fldj) / inverse (flow_glow.py:24-31) of step `step` of block `level`:
 * u, y are [N,h,w,c] of that level; logdet [N] may be NULL */
int glowk_step_forward(glowk_handle* h, int level, int step, const float* u_dev, int N, float* y_dev, float* logdet_dev, void* stream);
int glowk_step_inverse(glowk_handle* h, int level, int step, const float* y_dev, int N, float* u_dev, void* stream);
/* ShiftAndLogScaleConvNet.call (flow_tfk_layers.py:73-84) of one step: xb [N,h,w,c/2] -> log_s, t [N,h,w,c/2] */
int glowk_coupling_net(glowk_handle* h, int level, int step, const float* xb_dev, int N, float* log_s_dev, float* t_dev, void* stream);

/* --- training step (train_glow.py:29-44, train_noisy_glow.py:31-44) -------------------------------------------------------
 * The reference's train_step is  loss = sum(-flow.log_prob(X)) / global_batch;  gradients = tape.gradient(loss,
 * flow.trainable_variables);  optimizer.apply_gradients(...)  under MirroredStrategy (per-replica gradients summed by an
 * all-reduce).  Here: glowk_param_grad fills ONE flat fp32 vector with  scale * d sum_n log_prob(x_n) / d theta  for this rank's
 * tiles (scale = -1 / global batch), the caller all-reduces that vector across ranks (RCCL; torch.distributed in the mirror),
 * glowk_apply_gradients takes the optimizer step on the engine's device-resident master copy of the parameters and refreshes
 * the packed kernel images on the device.  The sweep runs in the handle's arithmetic: with GLOWK_PREC_F16X3 / F16X2 on the split
 * kernels (hidden tensors stored in their scaled units, undone in the gradient assembly; weight-gradient GEMMs in the same
 * three-product fp16 split; fp32-class gradients, ~3x the exact sweep) wherever every level has training instances, under the range guard (a tripped sweep is repeated on the exact kernels
 * unless the policy is GLOWK_RANGE_ERROR); otherwise, and with GLOWK_PREC_F32, on the exact fp32 kernels.  glowk_apply_gradients
 * refreshes the exact images always and the fp16 hi/lo images (BatchNorm folds, power-of-two scales, epilogue constants,
 * range-guard limits; bit for bit the host packer's) when the handle is in a split arithmetic.
 * Layout of the vector: glowk_param_offset.  It holds every tf.Variable of the flow except the frozen P, P_inv, sign_S; the
 * BatchNorm moving mean / variance (non-trainable, never updated by the reference: the layers are called without training=)
 * are carried with zero gradient.
 * Memory: the sweep keeps, per pixel and flow step, the two hidden activations of the coupling network (4 KB at n_filters 512:
 * 5.6 GB for 32 tiles of 64x64, K = 32, L = 3) and, a level at a time, their two gradients (as much again for the largest
 * level) when a quarter / a third of the free device memory holds them; otherwise it re-runs each step's forward network and
 * works step by step -- slower, same results to fp32 rounding. */
size_t glowk_param_vector_size(glowk_handle* h);
/* where tensor `tensor_id` of (level, step) -- or a prior tensor (level, step ignored) -- sits in the vector */
int glowk_param_offset(glowk_handle* h, int level, int step, int tensor_id, size_t* offset, size_t* count);
/* x [N,H,W,C] -> grad_dev [glowk_param_vector_size] (overwritten) and, if logp_dev != NULL, log_prob [N].  The results are
 * ordered on `stream` like those of every other call.  The call returns once the data-gradient sweep has finished on the device (the
 * host's share -- the c x c fp64 chain rule of ActNorm / 1x1 -- then runs beside the last weight-gradient GEMMs; its results go up on
 * an internal stream that `stream` waits for by event); since version 400 it does NOT join `stream`: synchronise it before reading
 * grad_dev from the host. */
int glowk_param_grad(glowk_handle* h, const float* x_dev, int N, float scale, float* logp_dev, float* grad_dev, void* stream);
/* one optimizer step: optimizer 0 = Adam, 1 = Adamax (train_utils.py:23-41; Keras defaults beta_1 0.9, beta_2 0.999, epsilon 1e-7).
 * glowk_get_tensor / flow.variables see the new values. */
int glowk_apply_gradients(glowk_handle* h, const float* grad_dev, int optimizer, float lr, void* stream);

/* --- BASIS: the annealed-Langevin update around two log_prob_grad calls (run_basis_sep.py:152-181, dB branch) ------------- */
/* One step of basis_inner_loop for two sources, in place, as ONE kernel:
 *     mix = g(x1, x2) (:133-141),  (m1, m2) = grad_g(x1, x2) (:143-147),
 *     x_k <- x_k + eta (g_k + lambda_recon m_k (mixed - mix)) + sqrt(2 eta) N(0, I)        (:163-164, :180-181)
 * with g_k = compute_grad_logprob(x_k, model_k) supplied by the caller (glowk_log_prob_grad).  All tensors hold n floats.
 * The normal draws come from the engine's counter-based device RNG (Philox4x32-10 keyed by `seed`, counter = element, `step`,
 * source) unless eps1_dev / eps2_dev supply them (tests replay the oracle's draws; the reference draws unseeded).
 * `offset` (a multiple of 4) is the position of element 0 in that stream: a rank that holds tiles [a, b) of the mixture passes
 * a * H * W * C and draws exactly what one process would have drawn for those tiles, so a sharded run is the unsharded one.
 * nonfinite_dev (optional, one int on the device): set to 1 when a gradient, the mixture or an updated value is not finite --
 * the reference's debug asserts (:183-191). */
int glowk_basis_update(float* x1_dev, float* x2_dev, const float* g1_dev, const float* g2_dev, const float* mixed_dev, size_t n,
                       float eta, float lambda_recon, const float* eps1_dev, const float* eps2_dev, uint64_t seed, uint64_t step,
                       uint64_t offset, int* nonfinite_dev, void* stream);
/* g(x1, x2) alone: the mixture of two sources in dB, sum in power (:133-141) */
int glowk_basis_mix(const float* x1_dev, const float* x2_dev, float* out_dev, size_t n, void* stream);
/* the device RNG itself: out[e] = the draw glowk_basis_update makes for element e of (seed, step, which); uniform != 0 gives
 * U(0, 1) from the same stream instead of N(0, 1) (the chain's initial state, run_basis_sep.py:360-361); `offset` (a multiple
 * of 4): out[0] is element `offset` of the stream, as for glowk_basis_update */
int glowk_random(float* out_dev, size_t n, uint64_t seed, uint64_t step, int which, int uniform, uint64_t offset, void* stream);
/* out = x + sigma * N(0, I), the draws being those of glowk_random(seed, step, which, offset): the input noise of the
 * noise-conditioned training step (train_noisy_glow.py:31, X + tf.random.normal(X.shape) * noise).  out_dev may equal x_dev. */
int glowk_add_noise(const float* x_dev, float* out_dev, size_t n, float sigma, uint64_t seed, uint64_t step, int which, uint64_t offset,
                    void* stream);

/* --- host utility ----------------------------------------------------------------------------------------------------------- */
/* CRC-32C (Castagnoli) of a host buffer: the checksum of TFRecord frames (datasets/preprocessing.py:197-271) and of TensorFlow
 * checkpoint bundles (train_utils.py:62-75), whose tensors are too large for an interpreted byte loop */
uint32_t glowk_crc32c(const void* host_data, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* GLOWK_H */
