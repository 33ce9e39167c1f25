"""BASIS loop: host-side pieces on CPU against the oracle, the full inner loop on the GPU against the oracle loop."""
import numpy as np
import pytest
import torch

from audiosourcesep_amd import basis
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import synthetic_mel_tiles
from oracle import basis_ref


def test_get_sigmas_matches_reference_schedule():
    s = basis.get_sigmas(1.0, 0.01, 10)                      # run_basis_sep defaults: 10 levels from 1.0 to 0.01
    assert s.dtype == np.float32 and s.shape == (10,)
    np.testing.assert_allclose(s[0], 1.0)
    np.testing.assert_allclose(s[-1], 0.01, rtol=1e-6)
    np.testing.assert_allclose(s[1:] / s[:-1], (0.01) ** (1 / 9), rtol=1e-5)     # geometric
    np.testing.assert_allclose(basis.get_sigmas(1.0, 0.01, 10, "logarithmic"), s, rtol=1e-5)
    with pytest.raises(ValueError):
        basis.get_sigmas(1.0, 0.01, 10, "linear")


def test_db_mixing_and_its_gradient():
    rng = np.random.default_rng(0)
    a, b = rng.uniform(-80, 10, (2, 3, 4, 5, 1))
    mix = basis.mixing_db(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    np.testing.assert_allclose(mix, basis_ref.g_db(a, b), rtol=1e-12)
    # two equal sources mix to themselves; power sum of x and -inf dB is x - 10 log10(2)
    np.testing.assert_allclose(basis.mixing_db(torch.from_numpy(a), torch.from_numpy(a)).numpy(), a, rtol=1e-12)
    m1, m2 = basis.grad_mixing_db(torch.from_numpy(a), torch.from_numpy(b))
    r1, r2 = basis_ref.grad_g_db(a, b)
    np.testing.assert_allclose(m1.numpy(), r1, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose((m1 + m2).numpy(), 1.0, rtol=1e-12)
    # grad_g is the gradient of g up to the factor K... check by central differences: d g / d a = softmax weight
    eps = 1e-5
    num = (basis_ref.g_db(a + eps, b) - basis_ref.g_db(a - eps, b)) / (2 * eps)
    np.testing.assert_allclose(num, r1, rtol=1e-5, atol=1e-8)


@pytest.mark.gpu
def test_basis_inner_loop_matches_oracle():
    from audiosourcesep_amd.flow_models.flow_glow import GlowFlow
    from audiosourcesep_amd.synthetic import calibrated_engine
    cfg = GlowConfig(H=16, W=16, C=1, L=2, K=3, F=128)
    e1, p1 = calibrated_engine(cfg, device=0, init_tiles=16, seed=1)
    e2, p2 = calibrated_engine(cfg, device=0, init_tiles=16, seed=2)
    m1, m2 = GlowFlow(e1), GlowFlow(e2)
    n, T = 4, 3
    gt1, gt2 = synthetic_mel_tiles(n, cfg, seed=10), synthetic_mel_tiles(n, cfg, seed=11)
    mixed = basis_ref.g_db(gt1.astype(np.float64), gt2.astype(np.float64))
    rng = np.random.default_rng(3)
    x1 = synthetic_mel_tiles(n, cfg, seed=12).astype(np.float64)
    x2 = synthetic_mel_tiles(n, cfg, seed=13).astype(np.float64)
    noise = rng.standard_normal((T, 2) + x1.shape)
    sigmas = basis.get_sigmas(1.0, 0.01, 10)
    sidx = 9   # last level: eta = delta, lambda = 1/sigma_L^2 = 1e4
    r1, r2 = basis_ref.inner_loop(mixed, x1, x2, p1, p2, cfg.as_dict(), sidx, sigmas, noise, T=T)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()   # noqa: E731
    nf = lambda t, which, shape: dev(noise[t][which])                                     # noqa: E731
    y1, y2 = basis.basis_inner_loop(dev(mixed), dev(x1), dev(x2), m1, m2, sidx, sigmas, T=T, noise_fn=nf, debug=True)
    np.testing.assert_allclose(y1.cpu().numpy(), r1, atol=2e-3)    # dB units, range 120
    np.testing.assert_allclose(y2.cpu().numpy(), r2, atol=2e-3)
    # the update moved the estimates (not a no-op) and stays finite over a two-level outer loop
    assert np.abs(r1 - x1).max() > 1e-3
    o1, o2, arr = basis.basis_outer_loop(dev(mixed), dev(x1), dev(x2), m1, m2, sigmas[-2:], T=2,
                                         restore_1={float(s): m1.state_dict() for s in sigmas[-2:]})
    assert torch.isfinite(o1).all() and torch.isfinite(o2).all() and len(arr["x1"]) == 3
    assert basis.shard(dev(mixed), 3, 1).shape[0] == 1
