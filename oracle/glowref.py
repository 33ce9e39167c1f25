"""CPU ORACLE (test infrastructure only) -- NumPy restatement of the reference Glow path.

This file is the parity checker for the HIP engine.  It is NOT part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import it.  The product path (``audiosourcesep_amd``) never imports anything from
``oracle/`` and fails loudly when the HIP library is missing.

Pinning status: *partially pinned*.  TensorFlow 2.2 / TFP 0.9 are not installable in the
build container (ordinary ``ModuleNotFoundError``), so the reference itself cannot be
executed.  The reference's own unit tests (``unittest_flow_models.py``) hold known answers
for the primitive bijectors with a *toy* coupling network (``2*log2``/``4*log2`` log-dets,
ActNorm scale exactly 2) and invertibility / ``fldj == -ildj`` cases for ActNorm, 1x1 conv,
GlowStep, GlowBlock and the 2/3-level Glow; ``tests/test_oracle_reference_cases.py`` checks
this file against every one of them.  ``ShiftAndLogScaleConvNet`` (incl. BatchNorm mode),
the prior and ``TransformedDistribution.log_prob`` have no golden vectors in the reference:
for those this restatement follows the source text line by line and is cross-checked by
independent means (autodiff Jacobian ``slogdet``, a second torch-CPU restatement built on
``F.conv2d``, central differences).  That part is "parity unpinned" in the prompt's sense.

Every function cites the reference ``file:line`` (relative to the reference root) it follows.
All tensors are NHWC.  The dtype of the computation is the dtype of the arrays handed in
(use ``cast_params`` to move a parameter dict to float64 / float32).

Parameter container: a flat ``dict[str, np.ndarray]`` keyed
``b{level}/s{k}/<tensor>`` with k the *creation* index of the step (``glowStep_k``,
flow_glow.py:44-49) plus ``prior/loc`` and ``prior/log_scale``; ``cfg`` is a plain dict
``{H,W,C,L,K,F,learntop,minval,maxval,use_logit,alpha,bn_eps}``.
"""
import numpy as np

TENSORS_STEP = (
    "actnorm/log_scale", "actnorm/shift",
    "inv1x1/P", "inv1x1/sign_S", "inv1x1/L", "inv1x1/log_S", "inv1x1/U",   # (+ optional "inv1x1/P_inv", :282-284)
    "nn/conv1/kernel", "nn/conv1/bias",
    "nn/bn1/gamma", "nn/bn1/beta", "nn/bn1/mean", "nn/bn1/var",
    "nn/conv2/kernel", "nn/conv2/bias",
    "nn/bn2/gamma", "nn/bn2/beta", "nn/bn2/mean", "nn/bn2/var",
    "nn/conv3/kernel", "nn/conv3/bias",
)


def default_cfg(**kw):
    cfg = dict(H=64, W=64, C=1, L=3, K=32, F=512, learntop=True, minval=-100.0, maxval=20.0,
               use_logit=False, alpha=1e-10, bn_eps=1e-3)
    cfg.update(kw)
    return cfg


def cast_params(params, dtype):
    return {k: np.asarray(v, dtype=dtype) for k, v in params.items()}


def level_shapes(cfg):
    """[(h, w, c)] of the tensor each GlowBlock's steps act on (flow_glow.py:63-77, 93-99, 153-174)."""
    H, W, C = cfg["H"], cfg["W"], cfg["C"]
    out = []
    h, w, c = H, W, C
    for lvl in range(cfg["L"]):
        h, w, c = h // 2, w // 2, c * 4
        out.append((h, w, c))
        c = c // 2  # factor out half before the next block
    return out


def latent_shape(cfg):
    """flow_builder.py:64-75."""
    s = 2 ** cfg["L"]
    return (cfg["H"] // s, cfg["W"] // s, cfg["C"] * s * s)


# --------------------------------------------------------------------------------------
# index maps (bit exact)
# --------------------------------------------------------------------------------------
def squeeze(x):
    """Squeeze._forward, flow_tfp_bijectors.py:170-174."""
    N, H, W, C = x.shape
    x = x.reshape(N, H // 2, 2, W // 2, 2, C)
    x = x.transpose(0, 1, 3, 5, 2, 4)
    return x.reshape(N, H // 2, W // 2, C * 4)


def unsqueeze(y):
    """Squeeze._inverse, flow_tfp_bijectors.py:176-180."""
    N, h, w, c4 = y.shape
    C = c4 // 4
    y = y.reshape(N, h, w, C, 2, 2)
    y = y.transpose(0, 1, 4, 2, 5, 3)
    return y.reshape(N, h * 2, w * 2, C)


# --------------------------------------------------------------------------------------
# SpecPreprocessing (flow_tfp_bijectors.py:364-396)
# --------------------------------------------------------------------------------------
def spec_pre_forward(x, cfg):
    """:372-379."""
    dt = x.dtype
    mn, mx = dt.type(cfg["minval"]), dt.type(cfg["maxval"])
    x = (x - mn) / (mx - mn)
    if cfg["use_logit"]:
        a = dt.type(cfg["alpha"])
        x = (dt.type(1.0) - dt.type(2.0) * a) * x + a
        x = np.log(x) - np.log(dt.type(1.0) - x)
    else:
        x = x - dt.type(0.5)
    return x


def spec_pre_inverse(y, cfg):
    """:381-388."""
    dt = y.dtype
    mn, mx = dt.type(cfg["minval"]), dt.type(cfg["maxval"])
    if cfg["use_logit"]:
        a = dt.type(cfg["alpha"])
        y = dt.type(1.0) / (dt.type(1.0) + np.exp(-y))
        y = (y - a) / (dt.type(1.0) - dt.type(2.0) * a)
    else:
        y = y + dt.type(0.5)
    return y * (mx - mn) + mn


def spec_pre_fldj(x, cfg):
    """:390-396 -> [N]."""
    dt = x.dtype
    mn, mx = dt.type(cfg["minval"]), dt.type(cfg["maxval"])
    u = (x - mn) / (mx - mn)
    log_det = np.log(np.ones_like(u) / (mx - mn))
    if cfg["use_logit"]:
        a = dt.type(cfg["alpha"])
        p = (dt.type(1.0) - dt.type(2.0) * a) * u + a
        log_det = log_det + (-np.log(p) - np.log(dt.type(1.0) - p) + np.log(dt.type(1.0) - dt.type(2.0) * a))
    return log_det.sum(axis=(1, 2, 3))


# --------------------------------------------------------------------------------------
# ActNorm (flow_tfp_bijectors.py:202-253)
# --------------------------------------------------------------------------------------
def actnorm_init(minibatch):
    """Data-dependent init, normalize='channel' (:222-234): population std + 1e-8 (added in fp32 there)."""
    dt = minibatch.dtype
    mean = minibatch.mean(axis=(0, 1, 2))
    std = minibatch.std(axis=(0, 1, 2)) + dt.type(10 ** (-8))
    scale = dt.type(1.0) / std
    return np.log(scale), -mean / std  # log_scale, shift


def actnorm_forward(x, log_scale, shift):
    """:242-243."""
    return x * np.exp(log_scale) + shift


def actnorm_inverse(y, log_scale, shift):
    """:246-247."""
    return (y - shift) / np.exp(log_scale)


def actnorm_fldj(x, log_scale):
    """:250-253 -> [N] (same value repeated)."""
    _, h, w, _ = x.shape
    return np.repeat(h * w * log_scale.sum(), x.shape[0])


# --------------------------------------------------------------------------------------
# Invertible1x1Conv (flow_tfp_bijectors.py:256-322)
# --------------------------------------------------------------------------------------
def inv1x1_init(c, rng):
    """:271-278 -- QR of a Gaussian matrix, scipy LU, diag pulled out of U."""
    import scipy.linalg
    w = np.linalg.qr(rng.standard_normal((c, c)))[0]
    p, l, u = scipy.linalg.lu(w)
    s = np.diag(u)
    return dict(P=p, sign_S=np.sign(s), log_S=np.log(np.abs(s)), L=l, U=np.triu(u, k=1))


def inv1x1_weight(P, L, U, sign_S, log_S):
    """:300-303 -- W = P (L*mask + I) (U*mask^T + diag(sign*exp(log_S)))."""
    c = P.shape[0]
    dt = P.dtype
    l_mask = np.tril(np.ones((c, c), dtype=dt), -1)
    Lm = L * l_mask + np.eye(c, dtype=dt)
    Um = U * l_mask.T + np.diag(sign_S * np.exp(log_S))
    return P @ (Lm @ Um)


def inv1x1_weight_inv(P, L, U, sign_S, log_S, P_inv=None):
    """:309-315 -- W^-1 = U^-1 L^-1 P_inv; P_inv is the stored variable, initialised to inv(P) (:282-284)."""
    c = P.shape[0]
    dt = P.dtype
    l_mask = np.tril(np.ones((c, c), dtype=dt), -1)
    Lm = L * l_mask + np.eye(c, dtype=dt)
    Um = U * l_mask.T + np.diag(sign_S * np.exp(log_S))
    return np.linalg.inv(Um) @ (np.linalg.inv(Lm) @ (np.linalg.inv(P) if P_inv is None else P_inv))


def inv1x1_forward(x, W):
    """:304-305 -- conv2d with a [1,1,c,c] filter == per-pixel row-vector x matrix."""
    return x @ W


def inv1x1_fldj(x, log_S):
    """:319-322."""
    _, h, w, _ = x.shape
    return np.repeat(h * w * log_S.sum(), x.shape[0])


# --------------------------------------------------------------------------------------
# ShiftAndLogScaleConvNet (flow_tfk_layers.py:31-84)
# --------------------------------------------------------------------------------------
def conv2d_same(x, kernel, bias):
    """tfk.layers.Conv2D(padding='same', stride 1): cross-correlation, HWIO kernel, zero pad."""
    N, H, W, Ci = x.shape
    kh, kw, _, Co = kernel.shape
    ph, pw = kh // 2, kw // 2
    xp = np.zeros((N, H + 2 * ph, W + 2 * pw, Ci), dtype=x.dtype)
    xp[:, ph:ph + H, pw:pw + W, :] = x
    cols = np.empty((N, H, W, kh * kw * Ci), dtype=x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            t = dy * kw + dx
            cols[..., t * Ci:(t + 1) * Ci] = xp[:, dy:dy + H, dx:dx + W, :]
    out = cols.reshape(-1, kh * kw * Ci) @ kernel.reshape(kh * kw * Ci, Co)
    return out.reshape(N, H, W, Co) + bias


def batchnorm_inference(x, gamma, beta, mean, var, eps):
    """tfk.layers.BatchNormalization called without training= (flow_tfk_layers.py:76,78) ->
    inference form (Keras semantics, SURVEY F8e/A.7 -- taken on trust, all four tensors explicit)."""
    return gamma * (x - mean) / np.sqrt(var + x.dtype.type(eps)) + beta


def convnet(xb, p, pre, eps):
    """ShiftAndLogScaleConvNet.call, flow_tfk_layers.py:73-84 -> (log_s, t)."""
    x = np.maximum(conv2d_same(xb, p[pre + "nn/conv1/kernel"], p[pre + "nn/conv1/bias"]), 0)   # :56-60,75
    x = batchnorm_inference(x, p[pre + "nn/bn1/gamma"], p[pre + "nn/bn1/beta"],
                            p[pre + "nn/bn1/mean"], p[pre + "nn/bn1/var"], eps)                  # :61,76
    x = np.maximum(conv2d_same(x, p[pre + "nn/conv2/kernel"], p[pre + "nn/conv2/bias"]), 0)     # :63-65,77
    x = batchnorm_inference(x, p[pre + "nn/bn2/gamma"], p[pre + "nn/bn2/beta"],
                            p[pre + "nn/bn2/mean"], p[pre + "nn/bn2/var"], eps)                  # :66,78
    x = conv2d_same(x, p[pre + "nn/conv3/kernel"], p[pre + "nn/conv3/bias"])                    # :68-70,79
    c = x.shape[-1] // 2
    return np.tanh(x[..., :c]), x[..., c:]                                                       # :80-84


# --------------------------------------------------------------------------------------
# AffineCouplingLayerSplit (flow_tfp_bijectors.py:124-153); nn(xb) -> (log_s, t)
# --------------------------------------------------------------------------------------
def coupling_forward(x, nn):
    """:134-140 -> (y, fldj[N]) (fldj per :150-153)."""
    c = x.shape[-1] // 2
    xa, xb = x[..., :c], x[..., c:]
    log_s, t = nn(xb)
    ya = np.exp(log_s) * xa + t
    return np.concatenate([ya, xb], axis=-1), log_s.sum(axis=(1, 2, 3))


def coupling_inverse(y, nn):
    """:142-148."""
    c = y.shape[-1] // 2
    ya, yb = y[..., :c], y[..., c:]
    log_s, t = nn(yb)
    xa = (ya - t) / np.exp(log_s)
    return np.concatenate([xa, yb], axis=-1)


# --------------------------------------------------------------------------------------
# GlowStep / GlowBlock (flow_glow.py:9-77)
# --------------------------------------------------------------------------------------
def _step_nn(p, pre, cfg, nn_override):
    if nn_override is not None:
        return nn_override
    return lambda xb: convnet(xb, p, pre, cfg["bn_eps"])


def step_forward(x, p, pre, cfg, nn_override=None):
    """GlowStep: Chain([coupling, inv1x1, actnorm]) applied right-to-left (flow_glow.py:21-22) -> (y, fldj[N])."""
    a = actnorm_forward(x, p[pre + "actnorm/log_scale"], p[pre + "actnorm/shift"])
    W = inv1x1_weight(p[pre + "inv1x1/P"], p[pre + "inv1x1/L"], p[pre + "inv1x1/U"],
                      p[pre + "inv1x1/sign_S"], p[pre + "inv1x1/log_S"])
    v = inv1x1_forward(a, W)
    y, ld3 = coupling_forward(v, _step_nn(p, pre, cfg, nn_override))
    ld = actnorm_fldj(x, p[pre + "actnorm/log_scale"]) + inv1x1_fldj(a, p[pre + "inv1x1/log_S"]) + ld3
    return y, ld


def step_inverse(y, p, pre, cfg, nn_override=None):
    """Chain.inverse: coupling^-1, then inv1x1^-1, then actnorm^-1."""
    v = coupling_inverse(y, _step_nn(p, pre, cfg, nn_override))
    Winv = inv1x1_weight_inv(p[pre + "inv1x1/P"], p[pre + "inv1x1/L"], p[pre + "inv1x1/U"],
                             p[pre + "inv1x1/sign_S"], p[pre + "inv1x1/log_S"], p.get(pre + "inv1x1/P_inv"))
    a = v @ Winv
    return actnorm_inverse(a, p[pre + "actnorm/log_scale"], p[pre + "actnorm/shift"])


def block_forward(x, p, lvl, cfg, nn_override=None):
    """GlowBlock: Chain(glow_steps + [squeeze]) (flow_glow.py:51-52): squeeze first, then steps K-1 ... 0."""
    u = squeeze(x)
    ld = np.zeros(x.shape[0], dtype=x.dtype)
    for k in reversed(range(cfg["K"])):
        u, l = step_forward(u, p, "b%d/s%d/" % (lvl, k), cfg, nn_override)
        ld = ld + l
    return u, ld


def block_inverse(y, p, lvl, cfg, nn_override=None):
    """Chain.inverse: steps 0 ... K-1 inverted, then unsqueeze."""
    u = y
    for k in range(cfg["K"]):
        u = step_inverse(u, p, "b%d/s%d/" % (lvl, k), cfg, nn_override)
    return unsqueeze(u)


# --------------------------------------------------------------------------------------
# GlowBijector_{2,3,4}blocks (flow_glow.py:80-329)
# --------------------------------------------------------------------------------------
def glow_forward(x, p, cfg, nn_override=None):
    """_forward (:102-108 / :176-185 / :268-282) + _forward_log_det_jacobian (:119-126 / :198-209 / :298-313).
    Factored-out halves are plain row-major reshapes to the latent's spatial size, not squeezes."""
    N = x.shape[0]
    Hl, Wl, _ = latent_shape(cfg)
    zs = []
    ld = np.zeros(N, dtype=x.dtype)
    hcur = x
    for lvl in range(cfg["L"]):
        o, l = block_forward(hcur, p, lvl, cfg, nn_override)
        ld = ld + l
        if lvl < cfg["L"] - 1:
            c = o.shape[-1] // 2
            z, hcur = o[..., :c], o[..., c:]
            zs.append(np.ascontiguousarray(z).reshape(N, Hl, Wl, -1))
        else:
            zs.append(o)
    return np.concatenate(zs, axis=-1), ld


def glow_inverse(z, p, cfg, nn_override=None):
    """_inverse (:110-117 / :187-196 / :284-296)."""
    N = z.shape[0]
    shapes = level_shapes(cfg)
    L = cfg["L"]
    # channel extents of z1..zL in the latent
    _, _, Cl = latent_shape(cfg)
    widths = []
    rem = Cl
    for lvl in range(L - 1):
        widths.append(rem // 2)
        rem = rem // 2
    widths.append(rem)
    offs = np.cumsum([0] + widths)
    parts = [z[..., offs[i]:offs[i + 1]] for i in range(L)]
    h = block_inverse(parts[L - 1], p, L - 1, cfg, nn_override)
    for lvl in reversed(range(L - 1)):
        hh, ww, cc = shapes[lvl]
        zl = np.ascontiguousarray(parts[lvl]).reshape(N, hh, ww, cc // 2)
        h = block_inverse(np.concatenate([zl, h], axis=-1), p, lvl, cfg, nn_override)
    return h


# --------------------------------------------------------------------------------------
# prior + TransformedDistribution (flow_builder.py:116-144)
# --------------------------------------------------------------------------------------
def prior_log_prob(z, p, cfg):
    """learntop: Independent(MultivariateNormalDiag(loc, scale_diag=exp(v)), 2) (:131-139) summed over
    [h,w,c]; else Normal(0,1) over the latent (:142-144)."""
    dt = z.dtype
    half_log_2pi = dt.type(0.5 * np.log(2.0 * np.pi))
    if cfg["learntop"]:
        loc, v = p["prior/loc"], p["prior/log_scale"]
        e = (z - loc) / np.exp(v)
        lp = -dt.type(0.5) * e * e - v - half_log_2pi
    else:
        lp = -dt.type(0.5) * z * z - half_log_2pi
    return lp.sum(axis=(1, 2, 3))


def bijector_forward(x, p, cfg, nn_override=None):
    """Chain([glow, prepro]).forward and its fldj: the map data -> latent (flow_builder.py:127)."""
    y = spec_pre_forward(x, cfg)
    z, ld = glow_forward(y, p, cfg, nn_override)
    return z, ld + spec_pre_fldj(x, cfg)


def bijector_inverse(z, p, cfg, nn_override=None):
    return spec_pre_inverse(glow_inverse(z, p, cfg, nn_override), cfg)


def log_prob(x, p, cfg, nn_override=None):
    """TransformedDistribution(prior, Invert(chain)).log_prob(x) = prior.log_prob(F(x)) + fldj_F(x)
    (flow_builder.py:129,140-141; tfp 0.9 semantics)."""
    z, ld = bijector_forward(x, p, cfg, nn_override)
    return prior_log_prob(z, p, cfg) + ld


def sample_from_eps(eps, p, cfg, nn_override=None):
    """flow.sample: z = loc + exp(v)*eps (or eps), x = chain.inverse(z)."""
    if cfg["learntop"]:
        z = p["prior/loc"] + np.exp(p["prior/log_scale"]) * eps
    else:
        z = eps
    return bijector_inverse(z.astype(eps.dtype), p, cfg, nn_override)


# --------------------------------------------------------------------------------------
# build-time initialisation (flow_builder.py:116-125, flow_glow.py:40-49,153-174)
# --------------------------------------------------------------------------------------
def _glorot_uniform(rng, shape):
    """Keras default kernel_initializer for Conv2D (glorot_uniform): limit = sqrt(6/(fan_in+fan_out))."""
    rf = int(np.prod(shape[:-2]))
    fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


def actnorm_data_init(p, minibatch, cfg, runtime_order=False, raw_minibatch_quirk=True):
    """Fill every step's ActNorm tensors of ``p`` (in place) the way the reference constructors do
    (flow_builder.py:121-125, flow_glow.py:40-49,93-99,153-174): the preprocessed minibatch is squeezed and
    pushed through the steps; each ActNorm is initialised (flow_tfp_bijectors.py:222-234) from the tensor that
    reaches it, then ``minibatch_updated = glow_step.forward(minibatch_updated)`` (:48).
    runtime_order=False visits steps in creation order 0..K-1 like the reference; True visits them in the
    order tfb.Chain applies them (K-1..0).  Quirk kept when raw_minibatch_quirk:
    GlowBijector_3blocks/4blocks hand the *raw* preprocessed minibatch to blocks 2+
    (flow_glow.py:162-165,171-174); Squeeze's reshape(-1, ...) reinterprets it as more, smaller samples.
    GlowBijector_2blocks propagates properly (:96-99)."""
    K, L = cfg["K"], cfg["L"]
    dt = p["b0/s0/inv1x1/P"].dtype
    mb0 = spec_pre_forward(np.asarray(minibatch, dtype=dt), cfg)
    shapes = level_shapes(cfg)
    quirk = raw_minibatch_quirk and L > 2
    prev_out = None
    for lvl in range(L):
        h, w, c = shapes[lvl]
        if lvl == 0:
            src = mb0
        elif quirk:
            src = mb0.reshape(-1, h * 2, w * 2, c // 4)
        else:
            src = prev_out[..., prev_out.shape[-1] // 2:]
        u = squeeze(src)
        for idx in range(K):
            k = K - 1 - idx if runtime_order else idx
            pre = "b%d/s%d/" % (lvl, k)
            p[pre + "actnorm/log_scale"], p[pre + "actnorm/shift"] = actnorm_init(u)
            u, _ = step_forward(u, p, pre, cfg)
        if lvl < L - 1 and not quirk:
            if runtime_order:
                prev_out = u
            else:
                prev_out, _ = block_forward(src, p, lvl, cfg)   # block.forward(minibatch) (flow_glow.py:96,159)
    return p


def init_params(minibatch, cfg, rng, dtype=np.float64):
    """What build_glow leaves in flow.variables right after construction: QR/LU 1x1 weights, Keras default
    conv init with conv3 zero (flow_tfk_layers.py:68-70), BN (1,0,0,1), prior (0, log 1), and ActNorm
    initialised from the minibatch (actnorm_data_init, reference order and quirk)."""
    p = {}
    F, K, L = cfg["F"], cfg["K"], cfg["L"]
    for lvl, (h, w, c) in enumerate(level_shapes(cfg)):
        for k in range(K):
            pre = "b%d/s%d/" % (lvl, k)
            w1 = inv1x1_init(c, rng)
            for name in ("P", "sign_S", "L", "log_S", "U"):
                p[pre + "inv1x1/" + name] = w1[name].astype(dtype)
            ci = c // 2
            p[pre + "nn/conv1/kernel"] = _glorot_uniform(rng, (3, 3, ci, F)).astype(dtype)
            p[pre + "nn/conv1/bias"] = np.zeros(F, dtype)
            p[pre + "nn/conv2/kernel"] = _glorot_uniform(rng, (1, 1, F, F)).astype(dtype)
            p[pre + "nn/conv2/bias"] = np.zeros(F, dtype)
            p[pre + "nn/conv3/kernel"] = np.zeros((3, 3, F, c), dtype)
            p[pre + "nn/conv3/bias"] = np.zeros(c, dtype)
            for bn in ("bn1", "bn2"):
                p[pre + "nn/%s/gamma" % bn] = np.ones(F, dtype)
                p[pre + "nn/%s/beta" % bn] = np.zeros(F, dtype)
                p[pre + "nn/%s/mean" % bn] = np.zeros(F, dtype)
                p[pre + "nn/%s/var" % bn] = np.ones(F, dtype)
    Hl, Wl, Cl = latent_shape(cfg)
    p["prior/loc"] = np.zeros((Hl, Wl, Cl), dtype)
    p["prior/log_scale"] = np.zeros((Hl, Wl, Cl), dtype)
    return actnorm_data_init(p, minibatch, cfg, runtime_order=False, raw_minibatch_quirk=True)
