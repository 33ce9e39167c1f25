"""Drop-in for ``flow_models/flow_builder.py:build_glow`` of the reference (flow_builder.py:60-146).

``build_glow(minibatch, data_shape, L, K, n_filters, learntop, l2_reg, mirrored_strategy, data_type, **kwargs)``
keeps the reference's signature and returns a :class:`GlowFlow` that quacks like the
``tfd.TransformedDistribution`` the scripts use: ``log_prob``, ``sample``, ``bijector``, ``variables``,
``trainable_variables`` (train_glow.py:30,39-43,74; run_basis_sep.py:77).  All arithmetic runs in the HIP engine
behind the C ABI; tensors are torch CUDA tensors (NHWC float32).
"""
import numpy as np
import scipy.linalg
import torch

from ..config import GlowConfig
from ..engine import GlowEngine
from .flow_glow import GlowFlow


def _glorot_uniform(rng, shape):
    """Keras default ``kernel_initializer`` of Conv2D (flow_tfk_layers.py:56-65 leave it at the default)."""
    rf = int(np.prod(shape[:-2]))
    lim = np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def initial_variables(cfg: GlowConfig, rng):
    """Every tensor build_glow creates except ActNorm (which is data dependent): QR -> LU 1x1 weights
    (flow_tfp_bijectors.py:271-294), Keras-default conv kernels with conv3 zero (flow_tfk_layers.py:68-70),
    BatchNormalization (gamma 1, beta 0, moving mean 0, moving variance 1), prior loc 0 / scale 1
    (flow_builder.py:131-139).  Host-side NumPy; no arithmetic of the hot path happens here."""
    p = {}
    F = cfg.F
    for lvl, (h, w, c) in enumerate(cfg.level_shapes()):
        for k in range(cfg.K):
            pre = "b%d/s%d/" % (lvl, k)
            q = np.linalg.qr(rng.standard_normal((c, c)))[0]        # flow_tfp_bijectors.py:271-278
            perm, lower, upper = scipy.linalg.lu(q)
            diag = np.diag(upper)
            p[pre + "actnorm/log_scale"] = np.zeros(c, np.float32)
            p[pre + "actnorm/shift"] = np.zeros(c, np.float32)
            p[pre + "inv1x1/P"] = perm.astype(np.float32)
            p[pre + "inv1x1/P_inv"] = np.linalg.inv(perm).astype(np.float32)      # :282-284
            p[pre + "inv1x1/sign_S"] = np.sign(diag).astype(np.float32)
            p[pre + "inv1x1/log_S"] = np.log(np.abs(diag)).astype(np.float32)
            p[pre + "inv1x1/L"] = lower.astype(np.float32)
            p[pre + "inv1x1/U"] = np.triu(upper, k=1).astype(np.float32)
            p[pre + "nn/conv1/kernel"] = _glorot_uniform(rng, (3, 3, c // 2, F))
            p[pre + "nn/conv1/bias"] = np.zeros(F, np.float32)
            p[pre + "nn/conv2/kernel"] = _glorot_uniform(rng, (1, 1, F, F))
            p[pre + "nn/conv2/bias"] = np.zeros(F, np.float32)
            p[pre + "nn/conv3/kernel"] = np.zeros((3, 3, F, c), np.float32)
            p[pre + "nn/conv3/bias"] = np.zeros(c, np.float32)
            for bn in ("bn1", "bn2"):
                p[pre + "nn/%s/gamma" % bn] = np.ones(F, np.float32)
                p[pre + "nn/%s/beta" % bn] = np.zeros(F, np.float32)
                p[pre + "nn/%s/mean" % bn] = np.zeros(F, np.float32)
                p[pre + "nn/%s/var" % bn] = np.ones(F, np.float32)
    Hl, Wl, Cl = cfg.latent_shape()
    p["prior/loc"] = np.zeros((Hl, Wl, Cl), np.float32)
    p["prior/log_scale"] = np.zeros((Hl, Wl, Cl), np.float32)
    return p


def build_glow(minibatch, data_shape, L=3, K=32, n_filters=512, learntop=True, l2_reg=None,
               mirrored_strategy=None, data_type="image", seed=None, device=None, precision=None, actnorm_init="reference", **kwargs):
    """Same arguments as the reference (flow_builder.py:60-61).

    * ``L`` outside {2,3,4} raises ``ValueError("L should be 2, 3 or 4")`` (:76-77).
    * ``data_type``: anything but ``"image"`` selects SpecPreprocessing(**kwargs) with kwargs
      ``minval, maxval, alpha, use_logit`` (:116-119).  ``"image"`` (ImgPreprocessing: fresh uniform noise on every
      call, flow_tfp_bijectors.py:337) is outside the accelerated path and raises ``NotImplementedError``.
    * ``l2_reg`` only attaches Keras regularizers whose losses the reference never adds to the objective
      (SURVEY section 3.1): accepted and ignored.
    * ``mirrored_strategy``: the reference only uses it to place variables; here each process owns one GPU
      (see audiosourcesep_amd.distributed) so it is accepted and ignored.
    * ``precision`` (extension): ``None`` / ``"f32"`` = exact fp32 kernels, ``"f16x3"`` = the fp16-split kernels (fp32-class
      accuracy, ~3x the speed).  ``seed`` / ``device`` are extensions too.
    * ``minibatch`` drives the data-dependent ActNorm init exactly like the reference constructor, including the
      raw-minibatch quirk of the 3/4-level graphs (flow_glow.py:162-165), on the GPU.
    * ``actnorm_init`` (extension): ``"reference"`` visits the steps in constructor order 0..K-1 although ``tfb.Chain`` applies them
      K-1..0 -- the flow that comes out is NOT normalised (log_prob per dimension of the order of -1e5 on its own
      minibatch) and only training repairs it; ``"runtime"`` visits them in the order the forward pass applies them, from the
      propagated tensors: every step's input is normalised from the first call on.  Short training runs (the noise-conditioned
      priors of the BASIS chain test) use ``"runtime"``.
    """
    if actnorm_init not in ("reference", "runtime"):
        raise ValueError("actnorm_init must be 'reference' or 'runtime'")
    if L not in (2, 3, 4):
        raise ValueError("L should be 2, 3 or 4")
    if data_type == "image":
        raise NotImplementedError("ImgPreprocessing (random dequantisation) is outside the MI355X hot path; "
                                  "use data_type='melspec' with minval/maxval/use_logit/alpha")
    H, W, C = [int(v) for v in data_shape]
    if "minval" not in kwargs or "maxval" not in kwargs:
        raise TypeError("SpecPreprocessing needs minval and maxval")          # flow_tfp_bijectors.py:365
    cfg = GlowConfig(H=H, W=W, C=C, L=L, K=K, F=int(n_filters), learntop=bool(learntop),
                     minval=float(kwargs["minval"]), maxval=float(kwargs["maxval"]),
                     use_logit=bool(kwargs.get("use_logit", True)), alpha=float(kwargs.get("alpha", 1e-10)))
    rng = np.random.default_rng(seed)
    eng = GlowEngine(cfg, device=device)
    eng.load_params(initial_variables(cfg, rng))
    if not torch.is_tensor(minibatch):
        minibatch = torch.as_tensor(np.asarray(minibatch, dtype=np.float32))
    if tuple(minibatch.shape[1:]) != (H, W, C):
        raise ValueError("minibatch must be [N, %d, %d, %d]" % (H, W, C))     # ActNorm asserts, flow_tfp_bijectors.py:218-220
    eng.actnorm_data_init(minibatch, runtime_order=actnorm_init == "runtime", raw_minibatch_quirk=actnorm_init == "reference")
    flow = GlowFlow(eng)
    if precision is not None:
        flow.set_precision(precision)
    return flow
