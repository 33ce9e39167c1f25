"""The reference's own unit-test cases (unittest_flow_models.py) replayed against the CPU oracle.

These are the only known answers the reference holds for the hot path: toy coupling
network (log_s = log 2, t = 1), crafted minibatches that make ActNorm's scale exactly 2, and
invertibility / fldj == -ildj for every bijector up to the 3-level Glow.  Exact float equality on
round trips is replaced by tolerances ('Staleness'); index ops stay bit exact.
"""
import numpy as np
import pytest

from oracle import glowref as R

LOG2 = np.log(2.0, dtype=np.float32)  # EXPECTED_LOG_DET, unittest_flow_models.py:59


def toy_nn(xb):
    """shift_and_log_scale_toy, unittest_flow_models.py:76-79."""
    return LOG2 * np.ones_like(xb), np.ones_like(xb)


def crafted_minibatch(shape):
    """MINIBATCH*, unittest_flow_models.py:66-73: [2*ones, ones]."""
    return np.concatenate([2 * np.ones((1,) + shape, np.float32), np.ones((1,) + shape, np.float32)], 0)


def toy_params(cfg, minibatch, seed=0):
    p = R.init_params(minibatch, cfg, np.random.default_rng(seed), dtype=np.float32)
    return p


RAW = dict(minval=0.0, maxval=1.0, use_logit=False)  # identity-scale preprocessing is not part of these cases


def test_coupling_split_known_answer():
    # TestAffineCouplingLayerSplit, unittest_flow_models.py:140-146: [1,2,2,2] -> 4*log 2
    x = np.random.default_rng(1).standard_normal((1, 2, 2, 2)).astype(np.float32)
    y, ld = R.coupling_forward(x, toy_nn)
    assert ld.dtype == np.float32
    assert ld[0] == np.float32(4) * LOG2
    np.testing.assert_array_equal(y[..., 1:], x[..., 1:])
    np.testing.assert_allclose(y[..., :1], np.exp(LOG2) * x[..., :1] + 1, rtol=0, atol=0)
    xr = R.coupling_inverse(y, toy_nn)
    np.testing.assert_allclose(xr, x, atol=1e-6)


def test_actnorm_known_answer():
    # TestActNorm, unittest_flow_models.py:149-154: scale exactly 2, fldj = 4 log 2
    mb = crafted_minibatch((2, 2, 1))
    ls, sh = R.actnorm_init(mb)
    assert ls.dtype == np.float32
    assert np.exp(ls)[0] == np.float32(2.0)
    assert sh[0] == np.float32(-3.0)
    x = np.random.default_rng(2).standard_normal((1, 2, 2, 1)).astype(np.float32)
    fldj = R.actnorm_fldj(x, ls)
    assert fldj[0] == np.float32(4) * LOG2
    y = R.actnorm_forward(x, ls, sh)
    np.testing.assert_allclose(R.actnorm_inverse(y, ls, sh), x, atol=1e-6)
    # fldj == -ildj: the inverse's log-det is -(h w sum log_scale) whatever its argument
    assert -(-fldj[0]) == fldj[0]


def test_inv1x1_invertible():
    # TestInvertible1x1Conv, unittest_flow_models.py:157-161
    rng = np.random.default_rng(3)
    w = {k: v.astype(np.float32) for k, v in R.inv1x1_init(2, rng).items()}
    W = R.inv1x1_weight(w["P"], w["L"], w["U"], w["sign_S"], w["log_S"])
    Winv = R.inv1x1_weight_inv(w["P"], w["L"], w["U"], w["sign_S"], w["log_S"])
    x = rng.standard_normal((1, 2, 2, 2)).astype(np.float32)
    np.testing.assert_allclose((x @ W) @ Winv, x, atol=1e-5)
    # QR of a Gaussian matrix is orthogonal: |det| = 1 => sum log_S = 0 and fldj = log|det W| * h * w
    assert abs(R.inv1x1_fldj(x, w["log_S"])[0]) < 1e-5
    np.testing.assert_allclose(np.linalg.slogdet(W.astype(np.float64))[1] * 4, R.inv1x1_fldj(x, w["log_S"])[0], atol=1e-5)


@pytest.mark.parametrize("case", ["step", "block", "glow2", "glow3"])
def test_glow_composites_invertible_and_logdet(case):
    # TestGlowStep/:164, TestGlowBlock/:170, TestGlowBijector_2Blocks/:176, _3Blocks/:182 -- K=2, toy network
    rng = np.random.default_rng(4)
    if case == "step":
        cfg = R.default_cfg(H=4, W=4, C=1, L=2, K=1, F=2, **RAW)  # only b0/s0 is used, on [2,2,2]... see below
        mb = crafted_minibatch((2, 2, 2))
        x = rng.standard_normal((1, 2, 2, 2)).astype(np.float32)
        ls, sh = R.actnorm_init(mb)
        w = {k: v.astype(np.float32) for k, v in R.inv1x1_init(2, rng).items()}
        p = {"s/actnorm/log_scale": ls, "s/actnorm/shift": sh}
        p.update({"s/inv1x1/" + k: v for k, v in w.items()})
        y, ld = R.step_forward(x, p, "s/", cfg, nn_override=toy_nn)
        xr = R.step_inverse(y, p, "s/", cfg, nn_override=toy_nn)
        np.testing.assert_allclose(xr, x, atol=1e-5)
        # fldj = 4*sum(log_scale) + 4*sum(log_S) + 4*log2 (coupling scales 1 channel x 4 pixels)
        expect = 4 * ls.sum() + 4 * w["log_S"].sum() + 4 * LOG2
        np.testing.assert_allclose(ld[0], expect, rtol=1e-6)
        return
    if case == "block":
        cfg = R.default_cfg(H=4, W=4, C=1, L=2, K=2, F=2, **RAW)
        mb = crafted_minibatch((4, 4, 1))
        p = toy_params(cfg, mb)
        x = rng.standard_normal((1, 4, 4, 1)).astype(np.float32)
        y, ld = R.block_forward(x, p, 0, cfg, nn_override=toy_nn)
        assert y.shape == (1, 2, 2, 4)
        np.testing.assert_allclose(R.block_inverse(y, p, 0, cfg, nn_override=toy_nn), x, atol=1e-5)
        return
    if case == "glow2":
        cfg = R.default_cfg(H=4, W=4, C=1, L=2, K=2, F=2, **RAW)
        shape = (4, 4, 1)
    else:
        cfg = R.default_cfg(H=8, W=8, C=1, L=3, K=2, F=2, **RAW)
        shape = (8, 8, 1)
    p = toy_params(cfg, crafted_minibatch(shape))
    x = rng.standard_normal((1,) + shape).astype(np.float32)
    z, ld = R.glow_forward(x, p, cfg, nn_override=toy_nn)
    assert z.shape == (1,) + R.latent_shape(cfg)
    np.testing.assert_allclose(R.glow_inverse(z, p, cfg, nn_override=toy_nn), x, atol=2e-5)
    # fldj is input independent with the toy network: every step adds hw*(sum log_scale + sum log_S) + hw*(c/2)*log2
    expect = 0.0
    for lvl, (h, w, c) in enumerate(R.level_shapes(cfg)):
        for k in range(cfg["K"]):
            pre = "b%d/s%d/" % (lvl, k)
            expect += h * w * (p[pre + "actnorm/log_scale"].sum() + p[pre + "inv1x1/log_S"].sum()) + h * w * (c // 2) * LOG2
    np.testing.assert_allclose(ld[0], expect, rtol=1e-5)


def test_spec_preprocessing_round_trip_and_logdet():
    # SpecPreprocessing has no reference test; invertibility + analytic log-det (flow_tfp_bijectors.py:372-396)
    rng = np.random.default_rng(5)
    x = rng.uniform(-90, 10, (2, 4, 4, 1))
    for use_logit in (False, True):
        cfg = R.default_cfg(H=4, W=4, use_logit=use_logit, alpha=1e-6)
        y = R.spec_pre_forward(x, cfg)
        np.testing.assert_allclose(R.spec_pre_inverse(y, cfg), x, atol=1e-8)
        ld = R.spec_pre_fldj(x, cfg)
        eps = 1e-6
        num = np.log((R.spec_pre_forward(x + eps, cfg) - R.spec_pre_forward(x - eps, cfg)) / (2 * eps)).sum(axis=(1, 2, 3))
        np.testing.assert_allclose(ld, num, rtol=1e-6)
