"""k_net_h3q (csrc/glowk_q.h) -- the small-grid form of the split coupling network with ALL conv1 blocks first, the split B fragments of
the whole hidden width in registers and conv2 as 16 back-to-back k-steps -- against the half-wave form it replaces on grids whose
pass-workgroups are alone on their CU (GLOWK_Q_OFF=1).  Same weight images, same arithmetic order inside conv1 / conv2 / conv3: every
result must be bit for bit equal -- log_prob and latent (plain forward), the input gradient (saving forward + backward network) and the
parameter gradients (the storing launches of the training sweep)."""
import os

import numpy as np
import pytest
import torch

from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles

pytestmark = pytest.mark.gpu


def _setenv(name, on):
    if on:
        os.environ[name] = "1"
    else:
        os.environ.pop(name, None)
    _lib.load().glowk_reload_env()


CASES = {
    # name: (cfg, tiles): small batches -- the deeper levels' pass-workgroups (4 per 128 pixels) fit the CUs in one round
    "config_B_shape_30_tiles": (GlowConfig(H=64, W=64, C=1, L=3, K=2, F=512), 30),
    "yaml_shape_96x64_30_tiles": (GlowConfig(H=96, W=64, C=1, L=3, K=2, F=512), 30),
    "32x32_F256_ragged_19": (GlowConfig(H=32, W=32, C=1, L=3, K=2, F=256), 19),
    "F384_L2_7_tiles": (GlowConfig(H=32, W=32, C=1, L=2, K=2, F=384), 7),
    "one_tile": (GlowConfig(H=64, W=64, C=1, L=3, K=1, F=512), 1),
}


@pytest.mark.parametrize("name", list(CASES))
def test_all_conv1_first_form_is_bitwise_the_half_wave_form(name):
    cfg, n = CASES[name]
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=16)
    eng.set_precision(_lib.PREC_F16X3)
    eng.set_range_policy("error")
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=21)).cuda()
    eng.param_grad(x, -1.0 / n)                         # (first sweep: sizes the training sweep's gradient scale; both forms then share it)
    res = {}
    try:
        for q in (False, True):
            _setenv("GLOWK_Q_OFF", not q)
            before = eng.kernel_families()
            lp, z = eng.log_prob(x, return_latent=True)
            lpg, dx = eng.log_prob_grad(x)
            _, g = eng.param_grad(x, -1.0 / n)
            torch.cuda.synchronize()
            fam = {k: v - before[k] for k, v in eng.kernel_families().items()}
            res[q] = (lp.clone(), z.clone(), lpg.clone(), dx.clone(), g.clone(), fam)
    finally:
        _setenv("GLOWK_Q_OFF", False)
    assert res[False][5]["small_grid_q"] == 0, res[False][5]
    assert res[True][5]["small_grid_q"] > 0 and res[True][5]["f32"] == 0, res[True][5]
    assert res[True][5]["small_grid_q"] <= res[True][5]["h3s_half"]
    for i, what in enumerate(("log_prob", "latent", "log_prob of the saving pass", "input gradient", "parameter gradient")):
        assert torch.isfinite(res[True][i]).all(), what
        assert torch.equal(res[True][i], res[False][i]), (name, what, float((res[True][i] - res[False][i]).abs().max()))
    assert eng.range_status() == (False, 0)


def test_small_grid_form_repeats_bit_for_bit():
    """The form's ring protocol (counted vmcnt + lgkmcnt(0) + raw barrier per op) under a stress of repeated calls: a hole would show as
    run-to-run differences (csrc/glowk_co.h has the story of one)."""
    cfg, n = CASES["config_B_shape_30_tiles"]
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=16)
    eng.set_precision(_lib.PREC_F16X3)
    eng.set_range_policy("error")
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=22)).cuda()
    lp0, dx0 = eng.log_prob_grad(x)
    lp0, dx0 = lp0.clone(), dx0.clone()
    for i in range(300):
        lp, dx = eng.log_prob_grad(x)
        assert torch.equal(lp, lp0) and torch.equal(dx, dx0), i
    assert eng.kernel_families()["small_grid_q"] > 0
