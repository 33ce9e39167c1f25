"""GPU tests of the range guard of the split arithmetics, the engine's memory accounting / device handling, the coupling-network
widths the reference's own trained flows used (n_filters 256) and the full-size parity checks of BASELINE.json's metric config.

Contracts checked here
  * a checkpoint whose hidden activations leave the fp16 range must never produce silent inf/NaN in f16x3 / f16x2:
    GLOWK_ERR_RANGE under the C ABI's default policy, an fp32 re-run (matching the oracle) under the mirror's default
    (the reference's callers assert on NaN: run_basis_sep.py:183-191, train_glow.py:115-118);
  * glowk_workspace_bytes is what glowk_reserve allocates, and after a reserve a compute call allocates nothing;
  * flow_builder.build_glow takes any n_filters (flow_builder.py:60); the reference's trained Glows used 256
    (trained_flow/generated_samples/glow_mnist_16_256_256_dist_ckpt-21.png).
"""
import ctypes
import warnings

import numpy as np
import pytest
import torch

from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig, CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
from oracle import glowref as R
from oracle import glowref_torch as RT

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def p64(params):
    return R.cast_params(params, np.float64)


def hot_engine(cfg, scale_log2=18):
    """A calibrated flow whose HIDDEN activations are ~2^scale_log2 times too large for the split kernels while its outputs
    stay ordinary: conv1 kernels and bias times 2^s, conv3 kernels times 2^-s (both exact).  |h1| ~ 0.2 * 2^18 = 5e4 > 16 376."""
    eng, params = calibrated_engine(cfg, device=0, init_tiles=8)
    s = float(2 ** scale_log2)
    for name in list(params):
        if name.endswith("nn/conv1/kernel") or name.endswith("nn/conv1/bias"):
            params[name] = params[name] * np.float32(s)
        elif name.endswith("nn/conv3/kernel"):
            params[name] = params[name] * np.float32(1.0 / s)
    eng.load_params(params)
    return eng, params


def test_range_guard_error_policy_and_fp32_fallback():
    cfg = GlowConfig(H=16, W=16, C=1, L=2, K=2, F=128)
    eng, params = hot_engine(cfg)
    x = synthetic_mel_tiles(5, cfg, seed=3)
    xd = dev(x)
    lp_ref, g_ref = RT.log_prob_and_grad(x.astype(np.float64), params, cfg.as_dict())
    assert np.all(np.isfinite(lp_ref))
    # exact fp32: fine, and the flag is never armed
    lp32 = eng.log_prob(xd)
    np.testing.assert_allclose(lp32.cpu().numpy(), lp_ref, rtol=1e-5)
    assert eng.range_status() == (False, 0)
    z32 = eng.forward(xd, with_logdet=False)
    for prec in (_lib.PREC_F16X3, _lib.PREC_F16X2):
        eng.set_precision(prec)
        # --- the C ABI's default policy: a loud, distinct error from every compute entry point ---
        eng.set_range_policy("error")
        with pytest.raises(_lib.GlowkRangeError):
            eng.log_prob(xd)
        with pytest.raises(_lib.GlowkRangeError):
            eng.forward(xd)
        with pytest.raises(_lib.GlowkRangeError):
            eng.log_prob_grad(xd)
        with pytest.raises(_lib.GlowkRangeError):
            eng.inverse(z32)
        with pytest.raises(_lib.GlowkRangeError):
            eng.sample_from_eps(torch.randn(3, *cfg.latent_shape(), device="cuda"))
        with pytest.raises(_lib.GlowkRangeError):
            eng.coupling_net(0, 0, torch.randn(2, 8, 8, 2, device="cuda"))
        # the raw return code is GLOWK_ERR_RANGE and the message names the cause
        lp = torch.empty(5, device="cuda")
        rc = eng.lib.glowk_log_prob(eng.h, ctypes.c_void_p(xd.data_ptr()), 5, ctypes.c_void_p(lp.data_ptr()), None, None)
        assert rc == _lib.ERR_RANGE and b"fp16 range" in eng.lib.glowk_last_error()
        assert eng.range_status() == (False, 0)      # the failing call cleared the flag; nothing was re-run
        # --- ignore: asynchronous, the caller polls; the flag is sticky until looked at ---
        eng.set_range_policy("ignore")
        bad = eng.log_prob(xd)
        # what round 1 handed back silently: overflowed activations become inf - inf = NaN, the next ReLU turns NaN into 0,
        # and the result is a FINITE number that is simply wrong (or inf/NaN when the last hidden layer overflows)
        assert not np.allclose(bad.cpu().numpy(), lp_ref, rtol=1e-3, equal_nan=False)
        assert eng.range_status()[0] is True
        assert eng.range_status()[0] is False
    # --- the mirror's default: re-run on the exact kernels inside the engine, one warning ---
    eng.set_precision(_lib.PREC_F16X3)
    eng.set_range_policy("fallback")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        lp = eng.log_prob(xd)
        lpg, g = eng.log_prob_grad(xd)
        z = eng.forward(xd, with_logdet=False)
    assert any("fp16 range" in str(m.message) for m in w)
    assert torch.equal(lp, lp32) and torch.equal(z, z32)          # bitwise the fp32 kernels' answer
    np.testing.assert_allclose(lpg.cpu().numpy(), lp_ref, rtol=1e-5)
    np.testing.assert_allclose(g.cpu().numpy(), g_ref, atol=1e-3 * np.abs(g_ref).max(), rtol=1e-2)
    assert eng.range_status(sync=False)[1] == 3
    assert eng.get_precision() == _lib.PREC_F16X3                  # the handle stays in the mode it was given


def test_range_guard_is_silent_on_a_normalised_flow():
    """Config-B-shaped calibrated flow in both split modes: no trip, no fp32 re-run, results as before."""
    cfg = GlowConfig(H=64, W=64, C=1, L=3, K=4, F=512)
    eng, params = calibrated_engine(cfg, device=0, init_tiles=16)
    x = dev(synthetic_mel_tiles(40, cfg, seed=4))
    lp32 = eng.log_prob(x)
    eng.set_range_policy("error")
    for prec, tol in ((_lib.PREC_F16X3, 2e-6), (_lib.PREC_F16X2, 5e-5)):
        eng.set_precision(prec)
        np.testing.assert_allclose(eng.log_prob(x).cpu().numpy(), lp32.cpu().numpy(), rtol=tol)
        eng.log_prob_grad(x)
    assert eng.range_status() == (False, 0)


def test_nonfinite_input_is_reported_not_hidden():
    """A NaN tile reaches the same guard (the reference's callers assert on it); under the fallback policy the fp32 answer --
    NaN for that tile, finite for the others -- comes back."""
    cfg = GlowConfig(H=16, W=16, C=1, L=2, K=2, F=128)
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=8)
    x = dev(synthetic_mel_tiles(4, cfg, seed=5))
    x[1, 3, 3, 0] = float("nan")
    eng.set_precision(_lib.PREC_F16X3)
    eng.set_range_policy("error")
    with pytest.raises(_lib.GlowkRangeError):
        eng.log_prob(x)
    eng.set_range_policy("fallback")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lp = eng.log_prob(x)
    assert not torch.isfinite(lp[1]) and torch.isfinite(lp[[0, 2, 3]]).all()


def test_workspace_bytes_is_what_reserve_allocates():
    cfg = GlowConfig(H=64, W=64, C=1, L=3, K=4, F=512)
    n = 96
    for prec in (_lib.PREC_F32, _lib.PREC_F16X3):
        # the HIP runtime loads a translation unit's code object (device memory) at its first launch: do that with a throw-away
        # engine, so that the measurement below sees the engine's own allocations only
        warm, _ = calibrated_engine(cfg, device=0, init_tiles=8)
        warm.set_precision(prec)
        warm.log_prob_grad(dev(synthetic_mel_tiles(n, cfg, seed=6)))
        warm.close()
        torch.cuda.empty_cache()
        eng, _ = calibrated_engine(cfg, device=0, init_tiles=8)
        eng.set_precision(prec)
        x = dev(synthetic_mel_tiles(n, cfg, seed=6))
        lp, dx = torch.empty(n, device="cuda"), torch.empty_like(x)
        torch.cuda.synchronize()
        # what the init left allocated (workspace for 8 tiles + the ActNorm scratch) is part of the handle's footprint
        before = eng.workspace_bytes(8, False) + 8 * 64 * 64 * 4
        free0 = torch.cuda.mem_get_info(0)[0]
        eng.reserve(n, with_grad=True)
        free1 = torch.cuda.mem_get_info(0)[0]
        want = eng.workspace_bytes(n, True)
        got = free0 - free1 + before
        # hipMalloc rounds every buffer up to its allocation granule (2 MiB): allow 12 buffers' worth
        assert want <= got + (1 << 20) and got <= want + 12 * (2 << 20), (want, got)
        # after the reserve a compute call of that size allocates nothing (required for hipGraph capture)
        rc = eng.lib.glowk_log_prob_grad(eng.h, ctypes.c_void_p(x.data_ptr()), n, ctypes.c_void_p(lp.data_ptr()), ctypes.c_void_p(dx.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(0)[0] == free1
        assert torch.isfinite(dx).all()
        # grad chunk: the largest batch whose footprint fits the budget
        m = eng.grad_max_tiles
        assert eng.workspace_bytes(m, True) <= 64 * 2 ** 30 and m >= 1024
        eng.close()


@pytest.mark.parametrize("big,small", [(40, 30), (12, 8), (160, 128)])
def test_smaller_batch_after_a_larger_reserve_keeps_its_masks_apart(big, small):
    """Round-3 advisor (high): the per-step ReLU-mask offsets are laid out once for the largest batch a handle has seen; the
    blocks a SMALLER batch needs must fit that layout (the half-wave launch form takes twice the blocks per pixel below its grid
    threshold: on 256 CUs a level-1 tensor of 40 tiles needed 320 blocks, of 30 tiles 480).  reserve(big) then the gradient /
    parameter-gradient of `small` tiles must be what a fresh engine returns -- bit for bit (same launches, same order)."""
    cfg = GlowConfig(H=64, W=64, C=1, L=3, K=2, F=512)
    x = dev(synthetic_mel_tiles(small, cfg, seed=9))
    for prec in (_lib.PREC_F16X3, _lib.PREC_F32):
        fresh, _ = calibrated_engine(cfg, device=0, init_tiles=8)
        fresh.set_precision(prec)
        lp0, dx0 = fresh.log_prob_grad(x)
        _, g0 = fresh.param_grad(x, -1.0 / small)
        lp0, dx0, g0 = lp0.clone(), dx0.clone(), g0.clone()
        fresh.close()
        eng, _ = calibrated_engine(cfg, device=0, init_tiles=8)
        eng.set_precision(prec)
        eng.reserve(big, with_grad=True)
        lp1, dx1 = eng.log_prob_grad(x)
        assert torch.equal(lp0, lp1) and torch.equal(dx0, dx1), (prec, float((dx0 - dx1).abs().max()))
        xb = dev(synthetic_mel_tiles(big, cfg, seed=10))
        eng.param_grad(xb, -1.0 / big)                      # the training buffers sized for the larger batch as well
        _, g1 = eng.param_grad(x, -1.0 / small)
        # (the split sweep's gradient scale is sized on the previous sweep: histories differ, so rounding-level agreement)
        tol = 0.0 if prec == _lib.PREC_F32 else 2e-6 * float(g0.abs().max())
        assert float((g0 - g1).abs().max()) <= tol, (prec, float((g0 - g1).abs().max()), tol)
        lp2, dx2 = eng.log_prob_grad(x)                     # and again after the larger sweep used the whole layout
        assert torch.equal(lp0, lp2) and torch.equal(dx0, dx2)
        assert eng.range_status() == (False, 0)
        eng.close()


def test_chunked_log_prob_sum_survives_a_range_fallback():
    """Round-3 advisor (medium): glowk_log_prob_sum accumulates chunk sums into *sum_dev; a chunk whose split-precision pass trips
    the range guard is re-run on the exact kernels (policy FALLBACK) -- the rejected pass must not have touched the sum."""
    cfg = GlowConfig(H=32, W=32, C=1, L=2, K=2, F=128)
    eng, params = hot_engine(cfg)
    eng.set_precision(_lib.PREC_F16X3)
    eng.set_range_policy("fallback")
    x = dev(synthetic_mel_tiles(10, cfg, seed=3))
    eng._max_tiles_cap = 4                                  # chunks of 4, 4, 2: the later ones accumulate
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        lp, total = eng.log_prob_sum(x)
    assert eng.range_status(sync=False)[1] >= 3             # every chunk fell back
    eng.set_precision(_lib.PREC_F32)
    lp32 = eng.log_prob(x)
    assert torch.equal(lp, lp32)
    want = float(lp32.double().sum())
    assert abs(float(total[0]) - want) <= 1e-9 * abs(want), (float(total[0]), want)
    # under ERROR the failing call leaves the caller's sum alone
    eng.set_precision(_lib.PREC_F16X3)
    eng.set_range_policy("error")
    total.fill_(123.0)
    with pytest.raises(_lib.GlowkRangeError):
        eng.log_prob_sum(x[:4], total=total)
    torch.cuda.synchronize()
    assert float(total[0]) == 123.0
    eng.close()


def test_device_is_restored_and_respected():
    """An engine on a non-current device must run there and leave the caller's current device alone (one-GPU boxes: the
    current device must simply be unchanged by every kind of call)."""
    ndev = torch.cuda.device_count()
    target = 1 if ndev > 1 else 0
    torch.cuda.set_device(0)
    cfg = GlowConfig(H=16, W=16, C=1, L=2, K=2, F=128)
    eng, _ = calibrated_engine(cfg, device=target, init_tiles=8)
    assert torch.cuda.current_device() == 0
    x = torch.from_numpy(synthetic_mel_tiles(3, cfg, seed=7)).to("cuda:%d" % target)
    lp, g = eng.log_prob_grad(x)
    assert lp.device.index == target and torch.cuda.current_device() == 0
    from audiosourcesep_amd.engine import squeeze, unsqueeze
    assert torch.equal(unsqueeze(squeeze(x)), x) and torch.cuda.current_device() == 0
    eng.profile_begin()
    eng.log_prob(x)
    eng.profile_end()
    eng.close()
    assert torch.cuda.current_device() == 0


@pytest.mark.parametrize("F", [256, 384])
def test_other_network_widths(F):
    """n_filters 256 (the reference's trained MNIST Glows) and 384: coupling network per level, log_prob, inverse and the input
    gradient against the fp64 oracle, all three arithmetics."""
    cfg = GlowConfig(H=16, W=32, C=1, L=3, K=2, F=F)
    eng, params = calibrated_engine(cfg, device=0, init_tiles=8)
    pr = p64(params)
    rng = np.random.default_rng(F)
    for level, (h, w, c) in enumerate(cfg.level_shapes()):
        xb = rng.standard_normal((3, h, w, c // 2)).astype(np.float32)
        ls_ref, t_ref = R.convnet(xb.astype(np.float64), pr, "b%d/s0/" % level, cfg.bn_eps)
        for prec in (_lib.PREC_F32, _lib.PREC_F16X3):
            eng.set_precision(prec)
            log_s, t = eng.coupling_net(level, 0, dev(xb))
            np.testing.assert_allclose(log_s.cpu().numpy(), ls_ref, atol=2e-5, rtol=1e-4)
            np.testing.assert_allclose(t.cpu().numpy(), t_ref, atol=2e-5, rtol=1e-4)
    x = synthetic_mel_tiles(5, cfg, seed=F)
    lp_ref, g_ref = RT.log_prob_and_grad(x.astype(np.float64), params, cfg.as_dict())
    scale = np.abs(g_ref).max()
    eng.set_range_policy("error")
    for prec, tol in ((_lib.PREC_F32, 1e-6), (_lib.PREC_F16X3, 2e-6), (_lib.PREC_F16X2, 5e-5)):
        eng.set_precision(prec)
        lp, z = eng.log_prob(dev(x), return_latent=True)
        np.testing.assert_allclose(lp.cpu().numpy(), lp_ref, rtol=tol)
        assert float((eng.inverse(z) - dev(x)).abs().max()) < (0.5 if prec == _lib.PREC_F16X2 else 5e-3)
        lpg, g = eng.log_prob_grad(dev(x))
        np.testing.assert_allclose(lpg.cpu().numpy(), lp_ref, rtol=2e-6)
        np.testing.assert_allclose(g.cpu().numpy(), g_ref, atol=3e-4 * scale, rtol=3e-3)
    # a batch large enough for the unsplit launch forms
    xl = dev(synthetic_mel_tiles(700, cfg, seed=F + 1))
    eng.set_precision(_lib.PREC_F32)
    lp32, g32 = eng.log_prob_grad(xl)
    eng.set_precision(_lib.PREC_F16X3)
    lp16, g16 = eng.log_prob_grad(xl)
    np.testing.assert_allclose(lp16.cpu().numpy(), lp32.cpu().numpy(), rtol=2e-6)
    # (isolated ReLU decisions may differ between the two arithmetics: a few entries of one tile by ~1e-3 of the maximum)
    np.testing.assert_allclose(g16.cpu().numpy(), g32.cpu().numpy(), atol=1e-3 * float(g32.abs().max()), rtol=5e-3)
    np.testing.assert_allclose(eng.log_prob(xl).cpu().numpy(), lp32.cpu().numpy(), rtol=2e-6)


def test_config_B_full_batch_parity():
    """BASELINE.json's metric config at the benchmark's own batch: the headline arithmetic (f16x3) against the exact-fp32 kernels
    on all 1024 tiles (bar 1e-4 relative; asserted 2e-6) and both against the fp64 oracle on 8 tiles (the torch restatement in
    double precision; the NumPy one is checked on 2 tiles in test_gpu_parity.py)."""
    cfg = CONFIG_B
    eng, params = calibrated_engine(cfg, device=0, init_tiles=64)
    x = synthetic_mel_tiles(1024, cfg, seed=1234)
    xd = dev(x)
    lp32 = eng.log_prob(xd)
    eng.set_range_policy("error")
    eng.set_precision(_lib.PREC_F16X3)
    lp16 = eng.log_prob(xd)
    eng.set_precision(_lib.PREC_F16X2)
    lp2 = eng.log_prob(xd)
    assert torch.isfinite(lp32).all()
    r16 = float(((lp16 - lp32).abs() / lp32.abs()).max())
    r2 = float(((lp2 - lp32).abs() / lp32.abs()).max())
    print("config B, 1024 tiles: max rel diff f16x3 vs f32 %.2e, f16x2 vs f32 %.2e" % (r16, r2))
    assert r16 < 2e-6 and r2 < 5e-5
    torch.set_num_threads(16)
    p = RT.to_torch(params, torch.float64)
    with torch.no_grad():
        ref = RT.log_prob(torch.from_numpy(x[:8].astype(np.float64)), p, cfg.as_dict())[0].numpy()
    e32 = np.max(np.abs(lp32[:8].cpu().numpy() - ref) / np.abs(ref))
    e16 = np.max(np.abs(lp16[:8].cpu().numpy() - ref) / np.abs(ref))
    print("config B, 8 tiles vs fp64 oracle: f32 %.2e, f16x3 %.2e" % (e32, e16))
    assert e32 < 1e-6 and e16 < 1e-6


def test_backward_kernels_normalise_any_gradient_magnitude():
    """Round 3: the split backward kernels scale every pixel's gradient vector by a power of two before the fp16
    split (the backward network is linear), so no gradient MAGNITUDE can leave the range -- the round-2 static backward bound fired on
    every call of a trained checkpoint.  A prior with tiny / huge scales makes the gradient wrt the latent 1e-8 ... 1e+8 times its
    usual size: the f16x3 input gradient must agree with the exact fp32 kernels' to fp32 rounding at every magnitude, with no range
    trip (policy "error"); a non-finite gradient is still reported."""
    cfg = GlowConfig(H=32, W=32, C=1, L=3, K=2, F=512)
    eng, params = calibrated_engine(cfg, device=0, init_tiles=16)
    x = dev(synthetic_mel_tiles(40, cfg, seed=5))
    base = np.asarray(params["prior/log_scale"], np.float32)
    eng.set_range_policy("error")
    for shift in (0.0, 9.0, -9.0, 4.0):        # prior sigma x e^shift: d log p / dz ~ e^(-2 shift) -> gradients 6e-9 ... 6e+7 of the usual
        eng.set_tensor("prior/log_scale", base + np.float32(shift))
        eng.finalize()
        eng.set_precision(_lib.PREC_F32)
        lp32, g32 = eng.log_prob_grad(x)
        eng.set_precision(_lib.PREC_F16X3)
        lp16, g16 = eng.log_prob_grad(x)              # raises GlowkRangeError if the guard fires
        scale = float(g32.abs().max())
        d = (g16 - g32).abs() / scale
        # (a ReLU decided differently by the two arithmetics moves a few dozen entries of ONE tile by a fixed absolute amount -- which
        #  is ~1e-3 of the maximum at the usual gradient scale and several times that where the prior's share of the gradient is tiny;
        #  tolerated as everywhere: bounded size, and the typical tile is untouched)
        assert float(d.max()) < 2e-2, (shift, scale, float(d.max()))
        per_tile = (g16 - g32).flatten(1).norm(dim=1) / g32.flatten(1).norm(dim=1)
        assert float(per_tile.median()) < 2e-6 and float((per_tile > 1e-4).float().mean()) < 0.3, (shift, per_tile)   # fp32-class, whatever the magnitude
        np.testing.assert_allclose(lp16.cpu().numpy(), lp32.cpu().numpy(), rtol=2e-6)
        print("prior sigma x e^%+.0f: max |g| %.2e, median per-tile |g16 - g32| / |g32| %.1e" % (shift, scale, float(per_tile.median())))
    assert eng.range_status() == (False, 0)
    # per-pixel: a batch whose tiles differ by 12 orders of magnitude in gradient size keeps every tile accurate
    eng.set_tensor("prior/log_scale", base)
    eng.finalize()
    xs = x.clone()
    xs[::2] = xs[::2] * 0.0 - 99.9               # constant tiles at the edge of the data range: very different gradient scale
    eng.set_precision(_lib.PREC_F32)
    _, g32 = eng.log_prob_grad(xs)
    eng.set_precision(_lib.PREC_F16X3)
    _, g16 = eng.log_prob_grad(xs)
    per_tile = (g16 - g32).flatten(1).norm(dim=1) / g32.flatten(1).norm(dim=1)
    assert float(per_tile.max()) < 5e-3 and float(per_tile.median()) < 2e-6, (float(per_tile.max()), float(per_tile.median()))
    # a NaN input still trips
    xb = x.clone()
    xb[3, 5, 5, 0] = float("nan")
    with pytest.raises(_lib.GlowkRangeError):
        eng.log_prob_grad(xb)


def test_training_sweep_recalibrates_its_gradient_scale():
    """Dynamic gradient scaling of the split training sweep: the scale of g_o is sized on the previous sweep with a
    256x headroom.  A jump of the gradient magnitude beyond that (here: the prior's sigma shrunk by e^-6 between two sweeps, gradients
    ~1.6e5 times larger) makes ONE sweep fall back to the exact kernels -- counted, never silent -- and the next one runs split again
    with the new scale; every gradient vector matches the exact sweep."""
    cfg = GlowConfig(H=32, W=32, C=1, L=2, K=2, F=256)
    eng, params = calibrated_engine(cfg, device=0, init_tiles=16)
    x = dev(synthetic_mel_tiles(24, cfg, seed=8))
    eng.set_precision(_lib.PREC_F16X3)
    eng.set_range_policy("fallback")

    def sweep():
        fam0, fb0 = eng.kernel_families(), eng.range_status(sync=False)[1]
        _, g = eng.param_grad(x, -1.0 / 24)
        fam = eng.kernel_families()
        return g.clone(), eng.range_status(sync=False)[1] - fb0, fam["f32"] - fam0["f32"]

    g1, fb1, f32_1 = sweep()
    g2, fb2, f32_2 = sweep()
    assert fb1 == 0 and fb2 == 0 and f32_1 == 0 and f32_2 == 0
    eng.set_precision(_lib.PREC_F32)
    _, gx = eng.param_grad(x, -1.0 / 24)
    assert float((g2 - gx).abs().max()) <= 5e-6 * float(gx.abs().max())
    eng.set_tensor("prior/log_scale", np.asarray(params["prior/log_scale"], np.float32) - np.float32(6.0))
    eng.finalize()
    eng.set_precision(_lib.PREC_F16X3)
    with pytest.warns(RuntimeWarning):
        g3, fb3, f32_3 = sweep()                    # gradients ~1.6e5 x larger than the scale was sized for: repeated on the exact kernels
    assert fb3 == 1 and f32_3 > 0
    g4, fb4, f32_4 = sweep()                        # recalibrated: split again
    assert fb4 == 0 and f32_4 == 0
    eng.set_precision(_lib.PREC_F32)
    _, gy = eng.param_grad(x, -1.0 / 24)
    for g in (g3, g4):
        assert float((g - gy).abs().max()) <= 5e-6 * float(gy.abs().max())
