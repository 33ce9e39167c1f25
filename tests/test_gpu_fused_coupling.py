"""The coupling fused into the network kernel (k_net_h3s<..., MODE | 16> + k_couple_edge): one flow step =
ONE heavy kernel at the 4-channel level.  Checked against the unfused launch form of the same arithmetic (k_net + k_couple, selected
with GLOWK_NO_FUSE=1: the only difference is the order in which the nine taps of the edge rows and the log-det are added up) to a few
units of fp32 rounding, against the fp64 oracle, and through inverse(forward(x)); over the geometries that exercise every branch:
workgroups inside a tile (edge rows, halo exchange), a tile per workgroup, several tiles per workgroup, a ragged last workgroup."""
import os

import numpy as np
import pytest
import torch

from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
from oracle import glowref as R

pytestmark = pytest.mark.gpu

CASES = {
    # name: (cfg, tiles) -- the batch is large enough for the one-workgroup-two-passes launch form (2 * workgroups > compute units)
    "64x64_edges_4_wgs_per_tile": (GlowConfig(H=64, W=64, C=1, L=3, K=2, F=512), 160),
    "96x64_edges_6_wgs_per_tile": (GlowConfig(H=96, W=64, C=1, L=3, K=2, F=512), 96),
    "32x32_one_tile_per_wg": (GlowConfig(H=32, W=32, C=1, L=2, K=3, F=512), 300),
    "16x16_four_tiles_per_wg_ragged": (GlowConfig(H=16, W=16, C=1, L=2, K=2, F=128), 1031),
    "16x32_two_tiles_per_wg": (GlowConfig(H=16, W=32, C=1, L=2, K=2, F=256), 771),
    "64x32_w16_edges": (GlowConfig(H=64, W=32, C=1, L=3, K=2, F=128), 301),
}


def _setenv(name, on):
    """The engine reads its diagnostic switches at load time: change one and have them read again (glowk_reload_env)."""
    if on:
        os.environ[name] = "1"
    else:
        os.environ.pop(name, None)
    _lib.load().glowk_reload_env()


def _run(eng, x, fuse):
    _setenv("GLOWK_NO_FUSE", not fuse)
    try:
        before = eng.fused_steps
        lp, z = eng.log_prob(x, return_latent=True)
        xr = eng.inverse(z)
        torch.cuda.synchronize()
        return lp, z, xr, eng.fused_steps - before
    finally:
        _setenv("GLOWK_NO_FUSE", False)


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("precision", ["f16x3", "f16x2"])
def test_fused_step_equals_network_plus_coupling(name, precision):
    cfg, n = CASES[name]
    eng, params = calibrated_engine(cfg, device=0, init_tiles=32)
    eng.set_precision({"f16x3": _lib.PREC_F16X3, "f16x2": _lib.PREC_F16X2}[precision])
    eng.set_range_policy("error")
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=5)).cuda()
    lp_f, z_f, xr_f, steps_f = _run(eng, x, True)
    lp_u, z_u, xr_u, steps_u = _run(eng, x, False)
    # the fused form ran: every step of the 4-channel level, in both directions; the unfused one did not
    assert steps_f == 2 * cfg.K and steps_u == 0, (steps_f, steps_u)
    assert torch.isfinite(lp_f).all()
    # same arithmetic, different summation order in the edge rows and the log-det: fp32 rounding apart
    rel = float(((lp_f - lp_u).abs() / lp_u.abs()).max())
    dz = float((z_f - z_u).abs().max())
    dx = float((xr_f - xr_u).abs().max())
    print(name, precision, "fused vs unfused: log_prob rel %.1e, |dz| %.1e, |dx| %.1e dB" % (rel, dz, dx))
    assert rel < 2e-6 and dz < (2e-4 if precision == "f16x3" else 1e-3) and dx < (5e-3 if precision == "f16x3" else 2e-2)
    # inverse(forward(x)) = x through the fused kernels (f16x2 rounds activations to fp16: DESIGN section 5)
    assert float((xr_f - x).abs().max()) < (0.02 if precision == "f16x3" else 0.3)
    # repeatable bit for bit
    lp2, z2, xr2, _ = _run(eng, x, True)
    assert torch.equal(lp2, lp_f) and torch.equal(z2, z_f) and torch.equal(xr2, xr_f)
    # batch independence: a window of the batch evaluated alone in the fused form gives the same tiles' results (workgroup
    # boundaries fall elsewhere for the small tiles; to rounding only, because the DEEPER levels pick their launch form -- passes
    # as workgroups or not -- by the grid size)
    m = max(n * 3 // 4, 1)
    lp_w, z_w, _, steps_w = _run(eng, x[:m].contiguous(), True)
    assert steps_w == 2 * cfg.K
    assert float(((lp_w - lp_f[:m]).abs() / lp_f[:m].abs()).max()) < 2e-6 and float((z_w - z_f[:m]).abs().max()) < 2e-4
    if precision == "f16x3":
        # against the fp64 oracle on a few tiles (the bar of the north star: 1e-4; fp32-class here)
        idx = [0, n // 2, n - 1]
        ref = R.log_prob(x[idx].cpu().numpy().astype(np.float64), R.cast_params(params, np.float64), cfg.as_dict())
        np.testing.assert_allclose(lp_f[idx].cpu().numpy(), ref, rtol=2e-6)
    assert eng.range_status() == (False, 0)


@pytest.mark.parametrize("name", ["64x64_edges_4_wgs_per_tile", "32x32_one_tile_per_wg", "16x16_four_tiles_per_wg_ragged"])
def test_fused_saving_pass_feeds_the_backward_sweep(name):
    """The saving forward pass of log_prob_grad takes the fused form too (MODE NET_FWD_SAVE | 16): it stores the ReLU masks and the
    coupling's pre-tanh inputs -- the only thing the backward pass needs of the network's output -- and no per-tap buffer at all.
    Input gradient and log_prob against the unfused form of the same build and against the exact fp32 kernels."""
    cfg, n = CASES[name]
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=32)
    eng.set_precision(_lib.PREC_F16X3)
    eng.set_range_policy("error")
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=6)).cuda()
    before = eng.fused_steps
    lp_f, g_f = eng.log_prob_grad(x)
    fused = eng.fused_steps - before
    _setenv("GLOWK_NO_FUSE", True)
    try:
        lp_u, g_u = eng.log_prob_grad(x)
    finally:
        _setenv("GLOWK_NO_FUSE", False)
    assert fused == cfg.K, fused                                   # every forward step of the 4-channel level
    assert float(((lp_f - lp_u).abs() / lp_u.abs()).max()) < 2e-6
    d = (g_f - g_u).abs() / g_u.abs().max()
    assert float(d.max()) < 2e-2 and float((d > 1e-3).float().mean()) < 5e-4, float(d.max())     # (isolated ReLU flips at most)
    per_tile = (g_f - g_u).flatten(1).norm(dim=1) / g_u.flatten(1).norm(dim=1)
    assert float(per_tile.median()) < 2e-6
    eng.set_precision(_lib.PREC_F32)
    lp32, g32 = eng.log_prob_grad(x)
    np.testing.assert_allclose(lp_f.cpu().numpy(), lp32.cpu().numpy(), rtol=2e-6)
    per_tile = (g_f - g32).flatten(1).norm(dim=1) / g32.flatten(1).norm(dim=1)
    assert float(per_tile.median()) < 3e-6
    lp2, g2 = None, None
    eng.set_precision(_lib.PREC_F16X3)
    lp2, g2 = eng.log_prob_grad(x)
    assert torch.equal(lp2, lp_f) and torch.equal(g2, g_f)          # repeatable bit for bit
    assert eng.range_status() == (False, 0)


CO_CASES = {
    # name: (cfg, tiles) -- grids of >= 4 x CUs 128-pixel workgroups at the 4-channel level, geometries with and without edge rows;
    # and mid-size grids (more workgroups than CUs, fewer than four per CU: one or two rounds of workgroups two to a CU)
    "64x64_8_wgs_per_tile": (GlowConfig(H=64, W=64, C=1, L=3, K=2, F=512), 160),
    "96x64_12_wgs_per_tile": (GlowConfig(H=96, W=64, C=1, L=3, K=2, F=512), 100),
    "128x128_w64_two_rows_per_wg": (GlowConfig(H=128, W=128, C=1, L=3, K=2, F=512), 40),
    "32x32_two_wgs_per_tile_ragged": (GlowConfig(H=32, W=32, C=1, L=2, K=3, F=512), 555),
    "16x16_two_tiles_per_wg_ragged": (GlowConfig(H=16, W=16, C=1, L=2, K=2, F=256), 2231),
    "64x32_w16_F384": (GlowConfig(H=64, W=32, C=1, L=3, K=2, F=384), 301),
    "96x64_mid_grid_basis_batch": (GlowConfig(H=96, W=64, C=1, L=3, K=2, F=512), 30),
    "64x64_mid_grid_two_rounds": (GlowConfig(H=64, W=64, C=1, L=3, K=2, F=512), 77),
}


@pytest.mark.parametrize("precision", ["f16x3", "f16x2"])
@pytest.mark.parametrize("name", list(CO_CASES))
def test_co_resident_form_equals_the_one_workgroup_per_cu_form(name, precision):
    """k_net_h3c (glowk_co.h: four-wave / 128-pixel workgroups, two to a CU, a ring of 16-KiB units) against k_net_h3s (eight waves /
    256 pixels, one per CU) on the same weights: the per-pixel arithmetic -- MFMA order, split, epilogues -- is the same, so the
    network's outputs are bit for bit equal; in the fused form only the set of edge rows k_couple_edge finishes and the order of the
    log-det partials differ.  Both the fused (GLOWK_NO_FUSE unset) and the P-to-HBM form (set), forward and inverse direction."""
    cfg, n = CO_CASES[name]
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=32)
    eng.set_precision({"f16x3": _lib.PREC_F16X3, "f16x2": _lib.PREC_F16X2}[precision])
    eng.set_range_policy("error")
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=8)).cuda()
    out = {}
    try:
        for co in (False, True):
            for fuse in (True, False):
                _setenv("GLOWK_CO_OFF", not co)
                _setenv("GLOWK_NO_FUSE", not fuse)
                before = eng.kernel_families()
                lp, z = eng.log_prob(x, return_latent=True)
                xr = eng.inverse(z)
                torch.cuda.synchronize()
                fam = {k: v - before[k] for k, v in eng.kernel_families().items()}
                out[(co, fuse)] = (lp, z, xr, fam)
    finally:
        _setenv("GLOWK_CO_OFF", False)
        _setenv("GLOWK_NO_FUSE", False)
    K = cfg.K
    assert out[(False, True)][3]["co_resident"] == 0 and out[(False, False)][3]["co_resident"] == 0
    for fuse in (True, False):
        fam = out[(True, fuse)][3]
        # every step of the 4-channel level, forward and inverse -- and of the 8-channel level (five conv3 units, two partial P buffers,
        # never fused) where its grid has more 128-pixel workgroups than CUs
        assert fam["co_resident"] in (2 * K, 4 * K), (fuse, fam)
        assert (fam["fused"] == 2 * K) == fuse, (fuse, fam)
    # P-to-HBM form: the network kernel alone differs, and its outputs are bit for bit equal
    for i in range(3):
        assert torch.equal(out[(True, False)][i], out[(False, False)][i]), (name, precision, i)
    # fused form: edge rows / log-det partials are summed in another order
    lp_a, z_a, x_a, _ = out[(True, True)]
    lp_b, z_b, x_b, _ = out[(False, True)]
    # (two-term mode: activations are ROUNDED to fp16 -- not smooth in their input, so rounding-level differences of one step's edge
    #  rows are amplified by the next; DESIGN section 5: inverse(forward(x)) returns x to 0.06 dB there, not 2e-4 dB)
    two = precision == "f16x2"
    assert float(((lp_a - lp_b).abs() / lp_b.abs()).max()) < (2e-5 if two else 2e-7)
    assert float((z_a - z_b).abs().max()) < (2e-3 if two else 4e-6)
    assert float((x_a - x_b).abs().max()) < (0.2 if two else 2e-3)        # (dB; inverse(forward(x)) itself is ~2e-4 dB from x)
    assert float((x_a - x).abs().max()) < (0.5 if two else 2e-2)
    lp2, z2 = eng.log_prob(x, return_latent=True)                            # (no switch set: the default form -- co-resident -- again,
    assert torch.equal(lp2, lp_a) and torch.equal(z2, z_a)                   #  bit for bit what it gave before)
    assert eng.range_status() == (False, 0)


@pytest.mark.parametrize("name,n,forms", [("64x64_8_wgs_per_tile", 160, "both passes per workgroup"), ("96x64_12_wgs_per_tile", 100, "both passes per workgroup"),
                                          ("64x64_8_wgs_per_tile", 30, "one pass per workgroup"), ("32x32_two_wgs_per_tile_ragged", 97, "one pass per workgroup"),
                                          ("64x64_8_wgs_per_tile", 20, "one pass per workgroup"),
                                          ("96x64_mid_grid_basis_batch", 30, "both passes per workgroup, mid-size grid"),
                                          ("64x64_mid_grid_two_rounds", 77, "both passes per workgroup, mid-size grid")])
def test_co_resident_gradient_path(name, n, forms):
    """The gradient path's level-0 launches in the co-resident form (k_net_h3c in NET_FWD_SAVE and NET_BWD mode): on grids that fill the chip
    both passes in one workgroup (saving pass fused with the coupling), on small grids one pass per workgroup (2 x Q/128 four-wave
    workgroups, two to a CU) -- against the eight-wave kernels (GLOWK_CO_OFF=1).  The ReLU-mask layout is shared, the arithmetic per
    accumulator too where both are 16x16x32 kernels: log_prob to rounding, gradients equal but for isolated ReLU flips, and both
    against the exact-fp32 kernels; bitwise repeatable."""
    cfg, _ = CO_CASES[name]
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=32)
    eng.set_precision(_lib.PREC_F16X3)
    eng.set_range_policy("error")
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=9)).cuda()
    out = {}
    try:
        for co in (False, True):
            _setenv("GLOWK_CO_OFF", not co)
            before = eng.kernel_families()
            lp, dx = eng.log_prob_grad(x)
            torch.cuda.synchronize()
            out[co] = (lp.clone(), dx.clone(), {k: v - before[k] for k, v in eng.kernel_families().items()})
    finally:
        _setenv("GLOWK_CO_OFF", False)
    K = cfg.K
    assert out[False][2]["co_resident"] == 0
    # level 0: K saving + K backward launches; the 8-channel level's saving launches too where its grid is large enough (its backward
    # network has no co-resident instance at 512 filters: 88 KB of LDS -- the mask layout is shared, the eight-wave kernel follows)
    assert out[True][2]["co_resident"] in (2 * K, 3 * K), (forms, out[True][2])
    assert out[True][2]["f32"] == 0
    lp_a, dx_a, _ = out[True]
    lp_b, dx_b, _ = out[False]
    assert torch.isfinite(dx_a).all()
    assert float(((lp_a - lp_b).abs() / lp_b.abs()).max()) < 2e-6
    d = (dx_a - dx_b).abs() / dx_b.abs().max()
    assert float(d.max()) < 2e-2 and float((d > 1e-3).float().mean()) < 5e-4, float(d.max())     # (isolated ReLU flips at most)
    per_tile = (dx_a - dx_b).flatten(1).norm(dim=1) / dx_b.flatten(1).norm(dim=1)
    assert float(per_tile.median()) < 3e-6, float(per_tile.median())
    eng.set_precision(_lib.PREC_F32)
    lp32, g32 = eng.log_prob_grad(x)
    np.testing.assert_allclose(lp_a.cpu().numpy(), lp32.cpu().numpy(), rtol=2e-6)
    per_tile = (dx_a - g32).flatten(1).norm(dim=1) / g32.flatten(1).norm(dim=1)
    assert float(per_tile.median()) < 3e-6
    eng.set_precision(_lib.PREC_F16X3)
    for _ in range(20):
        lp2, dx2 = eng.log_prob_grad(x)
        assert torch.equal(lp2, lp_a) and torch.equal(dx2, dx_a)                   # repeatable bit for bit
    assert eng.range_status() == (False, 0)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("shape", ["64x64_L3", "32x32_L2_notop", "24x24_L2_ragged"])
def test_one_lane_per_pixel_backward_merge_equals_the_four_lane_form(precision, shape):
    """k_bwd_light<C, 1> (grids of >= 2 x CUs 256-pixel workgroups: coalesced planar gathers, 16-byte row accesses) against the
    four-lanes-per-pixel form of the same launch (GLOWK_BWD_LIGHT_4=1): the nine taps are added in another order, nothing else
    differs -- input gradient and parameter gradient to fp32 rounding, and the input gradient against fp64 autograd on a few tiles."""
    from oracle import glowref_torch as RT
    cfg, n = {"64x64_L3": (GlowConfig(H=64, W=64, C=1, L=3, K=2, F=512), 600),            # level 0 AND level 1 take the one-lane form
              "32x32_L2_notop": (GlowConfig(H=32, W=32, C=1, L=2, K=3, F=256, learntop=False), 530),
              "24x24_L2_ragged": (GlowConfig(H=24, W=24, C=1, L=2, K=2, F=128), 950)}[shape]   # 136 800 pixels: a partial last workgroup
    eng, params = calibrated_engine(cfg, device=0, init_tiles=32)
    eng.set_precision({"f32": _lib.PREC_F32, "f16x3": _lib.PREC_F16X3}[precision])
    eng.set_range_policy("error")
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=11)).cuda()

    def run(four):
        _setenv("GLOWK_BWD_LIGHT_4", four)
        try:
            lp, dx = eng.log_prob_grad(x)
            lpp, g = eng.param_grad(x[:300].contiguous(), -1.0 / 300)
            torch.cuda.synchronize()
            return lp, dx, g
        finally:
            _setenv("GLOWK_BWD_LIGHT_4", False)

    lp1, dx1, g1 = run(False)
    lp4, dx4, g4 = run(True)
    assert torch.equal(lp1, lp4) and torch.isfinite(dx1).all() and torch.isfinite(g1).all()
    ddx = float((dx1 - dx4).abs().max() / dx4.abs().max())
    dg = float((g1 - g4).abs().max() / g4.abs().max())
    print(shape, precision, "one lane vs four: dx %.1e, param grad %.1e of the largest entry" % (ddx, dg))
    assert ddx < 2e-6 and dg < 2e-6
    lp1b, dx1b, g1b = run(False)
    assert torch.equal(dx1b, dx1)                                                # repeatable bit for bit
    # (the split training sweep sizes its gradient scale on the previous sweep: same products at another power of two)
    assert float((g1b - g1).abs().max() / g1.abs().max()) < 2e-6
    idx = [0, n // 2, n - 1]
    _, ref = RT.log_prob_and_grad(x[idx].cpu().numpy().astype(np.float64), params, cfg.as_dict())
    ref = torch.from_numpy(ref)
    err = float((dx1[idx].cpu().double() - ref).abs().max() / ref.abs().max())
    print("   vs fp64 autograd: %.1e" % err)
    assert err < (2e-4 if precision == "f32" else 1e-3)          # (the bar of the other gradient tests; isolated ReLU flips allowed)
