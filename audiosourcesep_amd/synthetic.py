"""Seeded synthetic weights and mel tiles.

There is no network for datasets or checkpoints, and the reference ships no Glow checkpoint, so the
benchmark and the parity tests run on these generators.  Weights are *not* the reference's
zero-initialised conv3 (that makes every coupling the identity); see the distributions below.
Host-side NumPy only; nothing here touches the GPU.
"""
import numpy as np
import scipy.linalg

from .config import GlowConfig


def step_prefix(level, k):
    """Key prefix of ``glowStep_k`` (creation index, flow_glow.py:44-49) of block ``level``."""
    return "b%d/s%d/" % (level, k)


def synthetic_params(cfg: GlowConfig, seed=2024, dtype=np.float32, conv3_std=0.005):
    """Flat ``{name: ndarray}`` with the reference's tensor layouts (HWIO conv kernels, [in,out] 1x1).

    conv3 kernels are N(0, 0.005^2), not SURVEY section 8(d)'s 0.01: measured on the GPU
    (scripts/scan_synth.py), at 0.01 the 96-step flow sits past a stability threshold -- 6-8 % of held-out tiles
    saturate tanh and explode to |z| > 1e2 (in the fp32 *oracle* too) -- while at 0.005 all 1024 held-out tiles
    stay at |z| < 20 and round-trip through inverse() to < 2e-4 dB."""
    rng = np.random.default_rng(seed)
    p = {}
    F = cfg.F
    for lvl, (h, w, c) in enumerate(cfg.level_shapes()):
        ci = c // 2
        for k in range(cfg.K):
            pre = step_prefix(lvl, k)
            p[pre + "actnorm/log_scale"] = rng.normal(0, 0.1, c)
            p[pre + "actnorm/shift"] = rng.normal(0, 0.1, c)
            # exactly the reference's init path (flow_tfp_bijectors.py:271-278), then perturbed
            wq = np.linalg.qr(rng.standard_normal((c, c)))[0]
            pm, lm, um = scipy.linalg.lu(wq)
            s = np.diag(um)
            mask = np.tril(np.ones((c, c)), -1)
            p[pre + "inv1x1/P"] = pm
            p[pre + "inv1x1/sign_S"] = np.sign(s)
            p[pre + "inv1x1/log_S"] = np.log(np.abs(s))
            p[pre + "inv1x1/L"] = lm + mask * rng.normal(0, 0.01, (c, c))
            p[pre + "inv1x1/U"] = np.triu(um, 1) + mask.T * rng.normal(0, 0.01, (c, c))
            p[pre + "nn/conv1/kernel"] = rng.normal(0, 0.05, (3, 3, ci, F))
            p[pre + "nn/conv1/bias"] = rng.normal(0, 0.01, F)
            p[pre + "nn/conv2/kernel"] = rng.normal(0, 0.05, (1, 1, F, F))
            p[pre + "nn/conv2/bias"] = rng.normal(0, 0.01, F)
            p[pre + "nn/conv3/kernel"] = rng.normal(0, conv3_std, (3, 3, F, c))
            p[pre + "nn/conv3/bias"] = rng.normal(0, 0.01, c)
            for bn in ("bn1", "bn2"):
                p[pre + "nn/%s/gamma" % bn] = 1.0 + rng.normal(0, 0.05, F)
                p[pre + "nn/%s/beta" % bn] = rng.normal(0, 0.05, F)
                p[pre + "nn/%s/mean" % bn] = rng.normal(0, 0.05, F)
                p[pre + "nn/%s/var" % bn] = 1.0 + rng.uniform(0, 0.1, F)
    Hl, Wl, Cl = cfg.latent_shape()
    p["prior/loc"] = rng.normal(0, 0.1, (Hl, Wl, Cl))
    p["prior/log_scale"] = rng.normal(0, 0.1, (Hl, Wl, Cl))
    return {k: np.ascontiguousarray(v, dtype=dtype) for k, v in p.items()}


def calibrated_engine(cfg: GlowConfig, device=None, init_tiles=64, seed=2024, init_seed=77):
    """Engine with the synthetic weights and ActNorm set by data-dependent init on the GPU.

    Random ActNorm tensors make a K=32 flow numerically meaningless (activations grow ~5 % per step and
    saturate tanh by the third level), so -- like any real Glow -- the benchmark weights get their ActNorm from
    the reference's own mechanism (flow_tfp_bijectors.py:222-234) on a synthetic minibatch, visiting the steps
    in the order the forward pass applies them so every step's input is normalised at run time.
    Returns (engine, params) with params holding the ActNorm values the engine computed (for the oracle)."""
    from .engine import GlowEngine
    params = synthetic_params(cfg, seed=seed)
    eng = GlowEngine(cfg, device=device)
    eng.load_params(params)
    eng.actnorm_data_init(synthetic_mel_tiles(init_tiles, cfg, seed=init_seed), runtime_order=True, raw_minibatch_quirk=False)
    params.update(eng.actnorm_params())
    return eng, params


def synthetic_mel_tiles(n, cfg: GlowConfig, seed=1234, dtype=np.float32):
    """dB mel tiles ``clip(-45 + 18 g, minval, maxval)`` with g unit-variance Gaussian noise AR(1)-smoothed
    (rho 0.96 along time/W, 0.79 along mel/H) -- the statistics of the shipped real tiles
    (basis_sep_results/.../results.npz)."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, cfg.H, cfg.W))
    for axis, rho in ((2, 0.96), (1, 0.79)):
        g = np.moveaxis(g, axis, 0)
        s = np.sqrt(1.0 - rho * rho)
        for i in range(1, g.shape[0]):
            g[i] = rho * g[i - 1] + s * g[i]
        g = np.moveaxis(g, 0, axis)
    x = np.clip(-45.0 + 18.0 * g, cfg.minval, cfg.maxval)
    x = np.repeat(x[..., None], cfg.C, axis=-1)
    return np.ascontiguousarray(x, dtype=dtype)
