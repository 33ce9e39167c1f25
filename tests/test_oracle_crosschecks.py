"""Independent checks of the CPU oracle that do not rely on its own formulas.

The reference's tests never exercise ShiftAndLogScaleConvNet, the prior or log_prob (toy network only), and
TF cannot run here, so these are what stands behind those parts of the oracle:
 * index maps bit exact against the explicit formula of appendix A.1,
 * NumPy restatement == torch-CPU restatement (different convolution code),
 * forward log-det == slogdet of the autodiff Jacobian,
 * inverse(forward(x)) == x with the real conv net and non-zero conv3,
 * autograd input gradient == central differences.
"""
import numpy as np
import pytest
import torch

from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import synthetic_params, synthetic_mel_tiles
from oracle import glowref as R
from oracle import glowref_torch as RT


def small_cfg(L=2, **kw):
    s = 2 ** L
    base = dict(H=2 * s, W=s, C=1, L=L, K=2, F=8)
    base.update(kw)
    return GlowConfig(**base)


def test_squeeze_formula_bit_exact():
    x = np.arange(2 * 6 * 4 * 3, dtype=np.float32).reshape(2, 6, 4, 3)
    y = R.squeeze(x)
    N, H, W, C = x.shape
    for n in range(N):
        for i in range(H // 2):
            for j in range(W // 2):
                for c in range(C):
                    for a in range(2):
                        for b in range(2):
                            assert y[n, i, j, 4 * c + 2 * a + b] == x[n, 2 * i + a, 2 * j + b, c]
    np.testing.assert_array_equal(R.unsqueeze(y), x)
    np.testing.assert_array_equal(RT.squeeze(torch.from_numpy(x)).numpy(), y)


@pytest.mark.parametrize("L", [2, 3, 4])
def test_factor_out_layout_bit_exact(L):
    """Latent layout of the 2/3/4-level graphs (flow_glow.py:102-108,176-185,268-282) with identity steps:
    K=0 makes every block a pure squeeze, so z is a pure permutation of x that we rebuild by hand."""
    s = 2 ** L
    cfg = R.default_cfg(H=s, W=2 * s, C=1, L=L, K=0, F=2)
    N = 2
    x = np.arange(N * s * 2 * s, dtype=np.float64).reshape(N, s, 2 * s, 1)
    z, ld = R.glow_forward(x, {}, cfg)
    assert z.shape == (N,) + R.latent_shape(cfg)
    assert np.all(ld == 0)
    Hl, Wl, _ = R.latent_shape(cfg)
    parts, h = [], x
    for lvl in range(L):
        o = R.squeeze(h)
        if lvl < L - 1:
            c = o.shape[-1] // 2
            parts.append(o[..., :c].reshape(N, Hl, Wl, -1))
            h = o[..., c:]
        else:
            parts.append(o)
    np.testing.assert_array_equal(z, np.concatenate(parts, -1))
    np.testing.assert_array_equal(R.glow_inverse(z, {}, cfg), x)
    assert sorted(z.ravel().tolist()) == sorted(x.ravel().tolist())


@pytest.mark.parametrize("L,learntop,use_logit", [(2, True, False), (3, True, False), (3, False, True), (4, True, False)])
def test_numpy_vs_torch_restatement(L, learntop, use_logit):
    cfg = small_cfg(L, learntop=learntop, use_logit=use_logit, alpha=1e-4)
    p = synthetic_params(cfg, dtype=np.float64)
    x = synthetic_mel_tiles(3, cfg, dtype=np.float64)
    lp = R.log_prob(x, p, cfg.as_dict())
    lpt, zt = RT.log_prob(torch.from_numpy(x), RT.to_torch(p), cfg.as_dict())
    np.testing.assert_allclose(lp, lpt.numpy(), rtol=1e-12)
    z, _ = R.bijector_forward(x, p, cfg.as_dict())
    np.testing.assert_allclose(z, zt.numpy(), rtol=1e-10, atol=1e-12)
    lpt2, _ = RT.log_prob(torch.from_numpy(x), RT.to_torch(p), cfg.as_dict(), evals_per_step=2)
    np.testing.assert_allclose(lpt2.numpy(), lpt.numpy(), rtol=1e-14)


@pytest.mark.parametrize("L", [2, 3])
def test_round_trip_real_network(L):
    cfg = small_cfg(L)
    p = synthetic_params(cfg, dtype=np.float64)
    x = synthetic_mel_tiles(2, cfg, dtype=np.float64)
    z, _ = R.bijector_forward(x, p, cfg.as_dict())
    np.testing.assert_allclose(R.bijector_inverse(z, p, cfg.as_dict()), x, atol=1e-9)
    # sample() path: eps -> z -> x, and forward brings it back
    eps = np.random.default_rng(0).standard_normal((2,) + cfg.latent_shape())
    xs = R.sample_from_eps(eps, p, cfg.as_dict())
    zs, _ = R.bijector_forward(xs, p, cfg.as_dict())
    np.testing.assert_allclose(zs, p["prior/loc"] + np.exp(p["prior/log_scale"]) * eps, atol=1e-9)
    # fp32 mode stays invertible to ~1e-4 in dB units (range 120)
    p32, x32 = R.cast_params(p, np.float32), x.astype(np.float32)
    z32, _ = R.bijector_forward(x32, p32, cfg.as_dict())
    assert z32.dtype == np.float32
    np.testing.assert_allclose(R.bijector_inverse(z32, p32, cfg.as_dict()), x32, atol=2e-3)


def test_fldj_matches_jacobian_slogdet():
    """log|det dF/dx| from autodiff on a tiny flow == the oracle's summed log-det terms."""
    cfg = GlowConfig(H=4, W=4, C=1, L=2, K=2, F=8)
    p = synthetic_params(cfg, dtype=np.float64)
    x = synthetic_mel_tiles(1, cfg, dtype=np.float64)
    pt = RT.to_torch(p)

    def f(xflat):
        _, z = RT.log_prob(xflat.reshape(1, 4, 4, 1), pt, cfg.as_dict())
        return z.reshape(-1)

    J = torch.autograd.functional.jacobian(f, torch.from_numpy(x).reshape(-1))
    _, logabsdet = np.linalg.slogdet(J.numpy())
    _, ld = R.bijector_forward(x, p, cfg.as_dict())
    np.testing.assert_allclose(ld[0], logabsdet, rtol=1e-10)


def test_input_gradient_vs_central_differences():
    cfg = GlowConfig(H=4, W=4, C=1, L=2, K=2, F=8)
    p = synthetic_params(cfg, dtype=np.float64)
    x = synthetic_mel_tiles(2, cfg, dtype=np.float64)
    lp, g = RT.log_prob_and_grad(x, p, cfg.as_dict())
    np.testing.assert_allclose(lp, R.log_prob(x, p, cfg.as_dict()), rtol=1e-12)
    eps = 1e-5
    num = np.zeros_like(x)
    for idx in np.ndindex(*x.shape):
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps
        xm[idx] -= eps
        num[idx] = (R.log_prob(xp, p, cfg.as_dict()).sum() - R.log_prob(xm, p, cfg.as_dict()).sum()) / (2 * eps)
    np.testing.assert_allclose(g, num, rtol=1e-6, atol=1e-8)


def test_data_dependent_init_normalises_first_step():
    """After build-time init the output of the first-created ActNorm of block 1 has zero mean / unit std per
    channel on the init minibatch (flow_tfp_bijectors.py:222-234), and conv3 is zero (coupling = identity)."""
    cfg = GlowConfig(H=8, W=8, C=1, L=3, K=2, F=4)
    mb = synthetic_mel_tiles(8, cfg, dtype=np.float64)
    p = R.init_params(mb, cfg.as_dict(), np.random.default_rng(0))
    u = R.squeeze(R.spec_pre_forward(mb, cfg.as_dict()))
    a = R.actnorm_forward(u, p["b0/s0/actnorm/log_scale"], p["b0/s0/actnorm/shift"])
    np.testing.assert_allclose(a.mean(axis=(0, 1, 2)), 0, atol=1e-9)
    np.testing.assert_allclose(a.std(axis=(0, 1, 2)), 1, atol=1e-6)
    assert not p["b0/s0/nn/conv3/kernel"].any()
    # parameter count of SURVEY appendix A.4 for the shipped configs
    from audiosourcesep_amd.config import CONFIG_B
    per_step = {4: 294940, 8: 322648, 16: 378160}
    n = sum(CONFIG_B.K * per_step[c] for (_, _, c) in CONFIG_B.level_shapes())
    assert abs(n - 31.9e6) < 0.1e6
