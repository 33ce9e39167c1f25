"""GPU: (1) the engine against the committed golden vectors (data only; no oracle call), (2) the Python mirror of the
reference API -- build_glow(...) and the object it returns -- used the way the reference's scripts and unit tests use it."""
import ast
import glob
import os

import numpy as np
import pytest
import torch

from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import synthetic_params, synthetic_mel_tiles
from oracle import glowref as R

pytestmark = pytest.mark.gpu
PRIMARY_FAMILIES = ("f32", "h3_32x32x16", "h3s_16x16x32", "h3s_half", "fused")   # (co_resident / small_grid_q count subsets of these)
FILES = sorted(f for f in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz"))
               if not os.path.basename(f).startswith(("real_", "basis_real_")))   # (those two hold tiles, not oracle vectors)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_engine_reproduces_golden(path):
    from audiosourcesep_amd.engine import GlowEngine
    g = dict(np.load(path))
    cfg = GlowConfig(**ast.literal_eval(str(g["cfg"][0])))
    eng = GlowEngine(cfg, device=0)
    eng.load_params(synthetic_params(cfg, seed=int(g["seed_w"])))
    z, ld = eng.forward(dev(g["x"]))
    np.testing.assert_allclose(z.cpu().numpy(), g["z"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(ld.cpu().numpy(), g["logdet"], rtol=1e-6)
    np.testing.assert_allclose(eng.log_prob(dev(g["x"])).cpu().numpy(), g["log_prob"], rtol=1e-6)   # bar: 1e-4
    np.testing.assert_allclose(eng.sample_from_eps(dev(g["eps"])).cpu().numpy(), g["x_sample"], atol=2e-3)
    y, lds = eng.step_forward(0, 0, dev(g["u_step"]))
    np.testing.assert_allclose(y.cpu().numpy(), g["y_step"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lds.cpu().numpy(), g["ld_step"], rtol=1e-5, atol=1e-4)
    c = g["u_step"].shape[-1]
    log_s, t = eng.coupling_net(0, 0, dev(g["u_step"][..., c // 2:]))
    np.testing.assert_allclose(log_s.cpu().numpy(), g["log_s"], atol=2e-5)
    np.testing.assert_allclose(t.cpu().numpy(), g["t"], atol=2e-5)


# ---- the reference API -------------------------------------------------------------------------------
MEL = dict(data_type="melspec", minval=-100.0, maxval=20.0, use_logit=False, alpha=1e-10)


def test_build_glow_signature_and_errors():
    from audiosourcesep_amd.flow_models import flow_builder
    mb = synthetic_mel_tiles(4, GlowConfig(H=16, W=16, L=2, K=2, F=128))
    with pytest.raises(ValueError, match="L should be 2, 3 or 4"):          # flow_builder.py:76-77
        flow_builder.build_glow(mb, [16, 16, 1], L=5, K=2, n_filters=128, **MEL)
    with pytest.raises(ValueError):
        flow_builder.build_glow(mb[:, :8], [16, 16, 1], L=2, K=2, n_filters=128, **MEL)   # ActNorm shape asserts
    with pytest.raises(NotImplementedError):
        flow_builder.build_glow(mb, [16, 16, 1], L=2, K=2, n_filters=128, data_type="image")


@pytest.mark.parametrize("L", [2, 3])
def test_build_glow_matches_reference_construction(L):
    """build_glow(minibatch, ...) then log_prob / sample / bijector protocol, against the oracle evaluated on the
    variables the flow reports (flow.variables is the checkpoint root in the reference, train_utils.py:67-68)."""
    from audiosourcesep_amd.flow_models import flow_builder
    s = 2 ** L
    shape = [4 * s, 2 * s, 1]
    cfg = GlowConfig(H=shape[0], W=shape[1], C=1, L=L, K=2, F=128)
    mb = synthetic_mel_tiles(8, cfg, seed=3)
    flow = flow_builder.build_glow(mb, shape, L=L, K=2, n_filters=128, learntop=True, l2_reg=None,
                                   mirrored_strategy=None, seed=11, **MEL)
    names = [v.name for v in flow.variables]
    assert len(names) == L * 2 * 22 + 2 and len(flow.trainable_variables) == L * 2 * 15 + 2   # 22 per step incl. P_inv (flow_tfp_bijectors.py:281-294)
    # flow.variables follows ONE order, the derived tf.Module traversal of tf_checkpoint.variable_order (variable i = checkpoint
    # key variables/<i>, train_utils.py:67-68); creation order (flow_glow.py:15-20) is the named alternative
    from audiosourcesep_amd.tf_checkpoint import variable_order
    assert names == variable_order(cfg)
    assert [n.split("/", 2)[2] for n in names[:3]] == ["actnorm/log_scale", "actnorm/shift", "nn/conv1/kernel"]
    created = [v.name for v in flow.variables_in_creation_order]
    assert sorted(created) == sorted(names)
    assert [n.split("/", 2)[2] for n in created[:8]] == ["actnorm/log_scale", "actnorm/shift", "inv1x1/P", "inv1x1/P_inv", "inv1x1/sign_S",
                                                         "inv1x1/L", "inv1x1/log_S", "inv1x1/U"]
    sd = flow.state_dict()
    np.testing.assert_allclose(sd["b0/s1/inv1x1/P_inv"], np.linalg.inv(sd["b0/s1/inv1x1/P"]), atol=1e-6)
    with pytest.raises(KeyError):
        flow.load_state_dict({k: v for k, v in sd.items() if not k.endswith("conv2/bias")})      # strict by default
    with pytest.raises(KeyError):
        flow.load_state_dict(dict(sd, stray=np.zeros(1)))
    flow.load_state_dict({k: v for k, v in sd.items() if not k.endswith("P_inv")})                # the derived tensor may be absent
    p = R.cast_params(flow.state_dict(), np.float64)
    # right after construction conv3 is zero => every coupling is the identity and log_prob is the prior of an
    # ActNorm/1x1 chain; data-dependent init must equal the oracle's (reference order + raw-minibatch quirk)
    assert not p["b0/s0/nn/conv3/kernel"].any()
    ref_init = R.actnorm_data_init(dict(p), mb.astype(np.float64), cfg.as_dict())
    for k in range(2):
        np.testing.assert_allclose(p["b0/s%d/actnorm/log_scale" % k], ref_init["b0/s%d/actnorm/log_scale" % k], atol=2e-5)
        np.testing.assert_allclose(p["b%d/s%d/actnorm/shift" % (L - 1, k)], ref_init["b%d/s%d/actnorm/shift" % (L - 1, k)], atol=2e-4)
    x = synthetic_mel_tiles(3, cfg)
    lp = flow.log_prob(dev(x))
    np.testing.assert_allclose(lp.cpu().numpy(), R.log_prob(x.astype(np.float64), p, cfg.as_dict()), rtol=1e-5)
    # "train" a little: assign non-zero conv3 through the Variable views, as a restored checkpoint would
    rng = np.random.default_rng(0)
    for v in flow.variables:
        if v.name.endswith("conv3/kernel"):
            v.assign(rng.normal(0, 0.01, v.shape).astype(np.float32))
    p = R.cast_params(flow.state_dict(), np.float64)
    lp = flow.log_prob(dev(x))
    np.testing.assert_allclose(lp.cpu().numpy(), R.log_prob(x.astype(np.float64), p, cfg.as_dict()), rtol=1e-5)
    # the precision switch (extension): same numbers to fp32 class in the fp16-split arithmetic, and back
    # (the 3-level flow initialised in the reference's order is not normalised at run time -- SURVEY F8a -- and its inputs
    # exceed what the range guard can vouch for: the call is then re-run on the fp32 kernels, with a warning)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        lp16 = flow.set_precision("f16x3").log_prob(dev(x))
    np.testing.assert_allclose(lp16.cpu().numpy(), lp.cpu().numpy(), rtol=2e-6)
    assert torch.equal(flow.set_precision("f32").log_prob(dev(x)), lp)
    with pytest.raises(ValueError):
        flow.set_precision("bf16")
    # tfb.Invert semantics: flow.bijector.inverse is data -> latent, forward is latent -> data
    z = flow.bijector.inverse(dev(x))
    assert tuple(z.shape[1:]) == cfg.latent_shape() == tuple(flow.chain.forward_event_shape(shape))
    np.testing.assert_allclose(flow.bijector.forward(z).cpu().numpy(), x, atol=2e-3)
    fldj = flow.chain.forward_log_det_jacobian(dev(x), event_ndims=3)
    ildj = flow.chain.inverse_log_det_jacobian(z, event_ndims=3)
    np.testing.assert_allclose(fldj.cpu().numpy(), -ildj.cpu().numpy(), rtol=1e-5)       # unittest_flow_models.py:39-46
    # one GlowStep through the view objects (TestGlowStep, :164-168)
    step = flow.chain.glow.blocks[0].steps[1]
    h, w, c = cfg.level_shapes()[0]
    u = dev(np.random.default_rng(1).standard_normal((2, h, w, c)))
    np.testing.assert_allclose(step.inverse(step.forward(u)).cpu().numpy(), u.cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(step.forward_log_det_jacobian(u).cpu().numpy(),
                               -step.inverse_log_det_jacobian(step.forward(u)).cpu().numpy(), rtol=1e-5, atol=1e-5)
    # sample(n): shape, finiteness, and it is the inverse image of a latent (forward brings back N(loc, scale) draws)
    xs = flow.sample(5, seed=7)
    assert tuple(xs.shape) == (5,) + tuple(shape) and torch.isfinite(xs).all()
    assert torch.equal(xs, flow.sample(5, seed=7))
    # save / restore round trip of the checkpoint container
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        flow.save(os.path.join(d, "ckpt.npz"))
        flow2 = flow_builder.build_glow(mb, shape, L=L, K=2, n_filters=128, seed=99, **MEL)
        flow2.restore(os.path.join(d, "ckpt.npz"))
        assert torch.equal(flow2.log_prob(dev(x)), lp)


# ---- input gradient (compute_grad_logprob, run_basis_sep.py:73-79) -----------------------------------------
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_input_gradient_reproduces_golden(path):
    """d sum(log_prob) / dx from glowk_log_prob_grad against the fp64 autograd vectors of the golden files."""
    from audiosourcesep_amd.engine import GlowEngine
    g = dict(np.load(path))
    cfg = GlowConfig(**ast.literal_eval(str(g["cfg"][0])))
    eng = GlowEngine(cfg, device=0)
    eng.load_params(synthetic_params(cfg, seed=int(g["seed_w"])))
    lp, dx = eng.log_prob_grad(dev(g["x"]))
    np.testing.assert_allclose(lp.cpu().numpy(), g["log_prob"], rtol=1e-6)
    scale = np.abs(g["grad"]).max()
    np.testing.assert_allclose(dx.cpu().numpy(), g["grad"], atol=2e-4 * scale, rtol=2e-3)
    # log_prob computed by the saving forward pass is bitwise the plain one
    assert torch.equal(lp, eng.log_prob(dev(g["x"])))


def test_input_gradient_autograd_contract():
    """The BASIS contract: x.requires_grad -> flow.log_prob(x).sum().backward() fills x.grad (tf.GradientTape in the
    reference); checked against the torch-CPU oracle's autograd on a calibrated flow with a ragged batch."""
    from audiosourcesep_amd.flow_models.flow_glow import GlowFlow
    from audiosourcesep_amd.synthetic import calibrated_engine
    from oracle import glowref_torch as RT
    cfg = GlowConfig(H=16, W=16, C=1, L=3, K=4, F=128)
    eng, params = calibrated_engine(cfg, device=0, init_tiles=16)
    flow = GlowFlow(eng)
    x = synthetic_mel_tiles(5, cfg, seed=8)
    xt = dev(x).requires_grad_(True)
    lp = flow.log_prob(xt)
    lp.sum().backward()
    lp_ref, g_ref = RT.log_prob_and_grad(x.astype(np.float64), params, cfg.as_dict())
    np.testing.assert_allclose(lp.detach().cpu().numpy(), lp_ref, rtol=1e-6)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), g_ref, atol=2e-4 * np.abs(g_ref).max(), rtol=2e-3)
    # weighted sum: grad_output is honoured
    xt2 = dev(x).requires_grad_(True)
    w = torch.arange(1, 6, device="cuda", dtype=torch.float32)
    (flow.log_prob(xt2) * w).sum().backward()
    np.testing.assert_allclose(xt2.grad.cpu().numpy(), g_ref * np.arange(1, 6).reshape(-1, 1, 1, 1),
                               atol=1e-3 * np.abs(g_ref).max(), rtol=2e-3)


def test_input_gradient_in_f16x3_arithmetic():
    """glowk_log_prob_grad with the fp16x3 kernels (forward with saves and backward network; the c = 16 level's backward
    falls back to exact fp32) against the fp64 autograd of the oracle and against the exact-fp32 kernels."""
    from audiosourcesep_amd import _lib
    from audiosourcesep_amd.synthetic import calibrated_engine
    from oracle import glowref_torch as RT
    cfg = GlowConfig(H=16, W=16, C=1, L=3, K=4, F=128)
    eng, params = calibrated_engine(cfg, device=0, init_tiles=16)
    x = synthetic_mel_tiles(21, cfg, seed=9)                 # ragged: 21 tiles -> partial workgroups at every level
    lp32, g32 = eng.log_prob_grad(dev(x))
    eng.set_precision(_lib.PREC_F16X3)
    lp16, g16 = eng.log_prob_grad(dev(x))
    # the saving pass runs the 32x32x16 kernel, the plain one the 16x16x32 kernel: same products, another summation order
    np.testing.assert_allclose(lp16.cpu().numpy(), eng.log_prob(dev(x)).cpu().numpy(), rtol=1e-6)
    lp_ref, g_ref = RT.log_prob_and_grad(x.astype(np.float64), params, cfg.as_dict())
    scale = np.abs(g_ref).max()
    np.testing.assert_allclose(lp16.cpu().numpy(), lp_ref, rtol=1e-6)
    np.testing.assert_allclose(g16.cpu().numpy(), g_ref, atol=2e-4 * scale, rtol=2e-3)
    np.testing.assert_allclose(g16.cpu().numpy(), g32.cpu().numpy(), atol=5e-5 * scale, rtol=1e-3)
    # a batch large enough for the one-workgroup-per-256-pixels launch at level 0 (small grids launch one workgroup per
    # hidden half): the two launch shapes must agree with the exact-fp32 kernels alike
    xl = dev(synthetic_mel_tiles(640, cfg, seed=10))
    lpl16, gl16 = eng.log_prob_grad(xl)
    eng.set_precision(_lib.PREC_F32)
    lpl32, gl32 = eng.log_prob_grad(xl)
    np.testing.assert_allclose(lpl16.cpu().numpy(), lpl32.cpu().numpy(), rtol=2e-6)
    np.testing.assert_allclose(gl16.cpu().numpy(), gl32.cpu().numpy(), atol=2e-4 * float(gl32.abs().max()), rtol=2e-3)   # the bar vs fp64


def test_input_gradient_f16x3_full_width():
    """n_filters = 512 (the width the 4-pass kernel forms exist for: level-3 backward with K = 144, split launches of small
    grids): gradient in both arithmetics against the fp64 autograd of the oracle, 64x64 tiles, L = 3, K = 2."""
    from audiosourcesep_amd import _lib
    from audiosourcesep_amd.synthetic import calibrated_engine
    from oracle import glowref_torch as RT
    cfg = GlowConfig(H=64, W=64, C=1, L=3, K=2, F=512)
    eng, params = calibrated_engine(cfg, device=0, init_tiles=8)
    x = synthetic_mel_tiles(3, cfg, seed=11)
    lp_ref, g_ref = RT.log_prob_and_grad(x.astype(np.float64), params, cfg.as_dict())
    scale = np.abs(g_ref).max()
    errs = {}
    for prec in (_lib.PREC_F32, _lib.PREC_F16X3):
        eng.set_precision(prec)
        lp, g = eng.log_prob_grad(dev(x))
        np.testing.assert_allclose(lp.cpu().numpy(), lp_ref, rtol=1e-6)
        d = np.abs(g.cpu().numpy() - g_ref) / scale
        errs[prec] = float(d.max())
        # Either arithmetic may decide a ReLU whose pre-activation is within rounding of zero differently from fp64: one such flip
        # moves a few dozen gradient entries around ONE pixel neighbourhood of ONE tile by ~1e-3 of the maximum (scripts/
        # grad_flip_probe.py: seeds 11-16 of this very shape -- most agree to 6e-7, fp32 flips in tile 2 at seed 11, the split
        # kernels in tile 1).  Everything outside such a neighbourhood is fp32-class; anything systematic is not tolerated.
        big = np.argwhere(d > 2e-4)
        if len(big):
            assert len(set(big[:, 0].tolist())) == 1 and big[:, 1].max() - big[:, 1].min() < 16 and len(big) < 200 and d.max() < 2e-2, (prec, d.max(), len(big))
        clean = d.copy()
        if len(big):
            t = int(big[0, 0])
            clean[t, max(0, big[:, 1].min() - 2):big[:, 1].max() + 3] = 0.0
        assert clean.max() < 2e-4, (prec, clean.max())
        np.testing.assert_allclose(eng.log_prob(dev(x)).cpu().numpy(), lp_ref, rtol=1e-6)
    print("max |grad - fp64 autograd| / max|grad|: fp32 kernels %.2e, fp16x3 kernels %.2e" % (errs[_lib.PREC_F32], errs[_lib.PREC_F16X3]))
    # a batch whose level-1 grid takes the unsplit launch while the deeper levels split 2- and 4-way
    xl = dev(synthetic_mel_tiles(80, cfg, seed=12))
    lp16, g16 = eng.log_prob_grad(xl)
    eng.set_precision(_lib.PREC_F32)
    lp32, g32 = eng.log_prob_grad(xl)
    np.testing.assert_allclose(lp16.cpu().numpy(), lp32.cpu().numpy(), rtol=2e-6)
    np.testing.assert_allclose(g16.cpu().numpy(), g32.cpu().numpy(), atol=1e-3 * float(g32.abs().max()), rtol=5e-3)


def test_gradient_path_is_bitwise_repeatable():
    """Regression for a missing barrier at the first step of k_net_f32 (block 0's small-conv operands could be overwritten by
    the first weight DMA while a slow wave was still reading them): 1 % of the exact-fp32 gradient calls differed from run to
    run in a few tiles.  400 repeats of one call, both arithmetics, every result bitwise equal to the first."""
    from audiosourcesep_amd import _lib
    from audiosourcesep_amd.synthetic import calibrated_engine
    cfg = GlowConfig(H=64, W=64, C=1, L=2, K=3, F=512)
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=16, seed=5)
    x = dev(synthetic_mel_tiles(200, cfg, seed=300))
    for prec in (_lib.PREC_F32, _lib.PREC_F16X3):
        eng.set_precision(prec)
        lp0, g0 = eng.log_prob_grad(x)
        for _ in range(400 if prec == _lib.PREC_F32 else 100):
            lp, g = eng.log_prob_grad(x)
            assert torch.equal(lp, lp0) and torch.equal(g, g0)


def test_input_gradient_four_levels_full_width():
    """L = 4 at n_filters = 512: the last level's backward pass is the K = 288 small convolution whose operands pass through
    a half-block LDS buffer (two DMA phases per step).  Gradient against the fp64 autograd of the oracle, and 100 repeats of
    one call bitwise equal (the extra barriers of that path are what a race would get past)."""
    from audiosourcesep_amd import _lib
    from audiosourcesep_amd.synthetic import calibrated_engine
    from oracle import glowref_torch as RT
    cfg = GlowConfig(H=32, W=32, C=1, L=4, K=2, F=512)
    eng, params = calibrated_engine(cfg, device=0, init_tiles=8)
    x = synthetic_mel_tiles(3, cfg, seed=21)
    lp_ref, g_ref = RT.log_prob_and_grad(x.astype(np.float64), params, cfg.as_dict())
    scale = np.abs(g_ref).max()
    for prec in (_lib.PREC_F32, _lib.PREC_F16X3):
        eng.set_precision(prec)
        before = eng.kernel_families()
        lp, g = eng.log_prob_grad(dev(x))
        fam = {k: v - before[k] for k, v in eng.kernel_families().items()}
        np.testing.assert_allclose(lp.cpu().numpy(), lp_ref, rtol=1e-6)
        assert float(np.abs(g.cpu().numpy() - g_ref).max() / scale) < 1e-3
        if prec == _lib.PREC_F16X3:
            # round 3: EVERY level of the gradient path on the split kernels -- the 32-channel level's saving forward pass and
            # its K = 288 backward network through the half-wave form of the 16x16x32 family (one 16-pixel half per wave: nine
            # k-steps of fragments fit the registers); not one launch of the exact fp32 kernel (4 levels x 2 steps x 2 directions)
            assert fam["f32"] == 0 and sum(fam[k] for k in PRIMARY_FAMILIES) == 4 * 2 * 2 and fam["h3s_half"] >= 4, fam
        else:
            assert fam["f32"] == 16 and sum(fam[k] for k in PRIMARY_FAMILIES) == 16, fam
    # a larger batch: the 32-channel level still has only its half-wave instances (whatever the grid), the others their usual forms
    eng.set_precision(_lib.PREC_F16X3)
    xm = dev(synthetic_mel_tiles(600, cfg, seed=22))
    before = eng.kernel_families()
    lpm, gm = eng.log_prob_grad(xm)
    fam = {k: v - before[k] for k, v in eng.kernel_families().items()}
    assert fam["f32"] == 0 and fam["h3s_half"] >= 4, fam
    eng.set_precision(_lib.PREC_F32)
    lpm32, gm32 = eng.log_prob_grad(xm)
    np.testing.assert_allclose(lpm.cpu().numpy(), lpm32.cpu().numpy(), rtol=2e-6)
    dgm = (gm - gm32).abs() / gm32.abs().max()
    assert float(dgm.max()) < 2e-2 and float((dgm > 1e-3).float().mean()) < 5e-4       # (isolated ReLU flips tolerated, as in the fuzz)
    # plain forward: the last level (c = 32) on the 16x16x32 split kernel (four passes, three fused output groups), as
    # workgroups of their own (small grids) and inside one workgroup (4 x workgroups > CUs: more than 4096 of these tiles)
    eng.set_precision(_lib.PREC_F32)
    xs = dev(x)
    xb = dev(synthetic_mel_tiles(4200, cfg, seed=23))
    lp32s, lp32b = eng.log_prob(xs), eng.log_prob(xb)
    np.testing.assert_allclose(lp32s.cpu().numpy(), lp_ref, rtol=1e-6)
    for prec, tol in ((_lib.PREC_F16X3, 2e-6), (_lib.PREC_F16X2, 5e-5)):
        eng.set_precision(prec)
        np.testing.assert_allclose(eng.log_prob(xs).cpu().numpy(), lp_ref, rtol=tol)
        lpb, zb = eng.log_prob(xb, return_latent=True)
        np.testing.assert_allclose(lpb.cpu().numpy(), lp32b.cpu().numpy(), rtol=tol)
        assert float((eng.inverse(zb) - xb).abs().max()) < (5e-3 if prec == _lib.PREC_F16X3 else 0.5)
    xl = dev(synthetic_mel_tiles(150, cfg, seed=22))   # 600 level-3 pixels: five workgroups, the last one ragged
    for prec in (_lib.PREC_F32, _lib.PREC_F16X3):
        eng.set_precision(prec)
        lp0, g0 = eng.log_prob_grad(xl)
        assert torch.isfinite(g0).all()
        for _ in range(100):
            lp, g = eng.log_prob_grad(xl)
            assert torch.equal(lp, lp0) and torch.equal(g, g0)
