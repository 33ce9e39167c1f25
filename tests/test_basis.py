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


def test_db_schedule_and_psnr():
    """The reference's BASIS schedule (sigma 1.0 -> 0.01 on tiles normalised to [0, 1], delta 2e-5; run_basis_sep.py:152-161,352-356,
    465-468) carried to the dB units the melspec flows live in: sigma x 120, delta x 120^2; eta(sigma_L) = delta either way."""
    from audiosourcesep_amd.config import CONFIG_YAML
    from audiosourcesep_amd.noise_conditioned import db_schedule, psnr_db
    sig, delta = db_schedule(CONFIG_YAML)
    np.testing.assert_allclose(sig, basis.get_sigmas(1.0, 0.01, 10) * 120.0, rtol=1e-6)
    assert sig.dtype == np.float32 and abs(delta - 2e-5 * 120.0 ** 2) < 1e-12
    # the Langevin step sizes in the two unit systems correspond: eta_dB = 120^2 eta_normalised at every level
    eta_n = 2e-5 * (basis.get_sigmas(1.0, 0.01, 10) / 0.01) ** 2
    eta_db = delta * (sig / sig[-1]) ** 2
    np.testing.assert_allclose(eta_db, eta_n * 120.0 ** 2, rtol=1e-5)
    sig4, _ = db_schedule(CONFIG_YAML, sigma1=0.3, sigmaL=0.01, num_classes=4)
    np.testing.assert_allclose(sig4[[0, -1]], [36.0, 1.2], rtol=1e-6)
    # PSNR with the 120 dB range as peak: identical tiles -> inf, a constant offset of 12 dB -> 20 dB
    a = np.random.default_rng(0).uniform(-100, 20, (3, 8, 8, 1)).astype(np.float32)
    assert psnr_db(a, a) == float("inf")
    assert abs(psnr_db(a + 12.0, a) - 20.0) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["16x16_L2_K3_F128", "configB_geometry_K2", "configB_geometry_K2_f16x3"])
def test_basis_inner_loop_matches_oracle(shape):
    """Three Langevin steps of run_basis_sep.py:163-181 with injected noise against the fp64 oracle loop: a small flow, and the
    64x64 L=3 n_filters=512 geometry of BASELINE config 5 (K = 2 so that the oracle's fp64 autograd stays in seconds) in the exact
    and in the split arithmetic (the default of GlowFlow's callers)."""
    from audiosourcesep_amd import _lib
    from audiosourcesep_amd.flow_models.flow_glow import GlowFlow
    from audiosourcesep_amd.synthetic import calibrated_engine
    cfg = GlowConfig(H=16, W=16, C=1, L=2, K=3, F=128) if shape.startswith("16x16") else GlowConfig(H=64, W=64, C=1, L=3, K=2, F=512)
    e1, p1 = calibrated_engine(cfg, device=0, init_tiles=16, seed=1)
    e2, p2 = calibrated_engine(cfg, device=0, init_tiles=16, seed=2)
    if shape.endswith("f16x3"):
        for e in (e1, e2):
            e.set_precision(_lib.PREC_F16X3)
            e.set_range_policy("error")
    m1, m2 = GlowFlow(e1), GlowFlow(e2)
    n, T = (4, 3) if shape.startswith("16x16") else (3, 3)
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


@pytest.mark.gpu
def test_device_rng_moments_and_streams():
    """The engine's Philox4x32-10 / Box-Muller stream: standard-normal moments, U(0, 1) support, a pure function of
    (seed, step, which, element), and distinct streams uncorrelated."""
    n = 1 << 22
    a = basis.device_randn((n,), "cuda", seed=1234, step=0, which=0)
    assert torch.isfinite(a).all()
    m, v = float(a.mean()), float(a.var())
    k = float(((a - m) ** 4).mean() / v ** 2)
    assert abs(m) < 4.0 / n ** 0.5 * 1.0 and abs(v - 1.0) < 5e-3 and abs(k - 3.0) < 2e-2, (m, v, k)
    assert float((a.abs() > 4.0).float().mean()) < 2e-4 and float(a.abs().max()) < 7.0
    # a pure function of its counters: the same call twice, a shorter call, another launch geometry
    assert torch.equal(a, basis.device_randn((n,), "cuda", seed=1234, step=0, which=0))
    assert torch.equal(a[:1001], basis.device_randn((1001,), "cuda", seed=1234, step=0, which=0))
    others = [basis.device_randn((n,), "cuda", seed=1234, step=0, which=1), basis.device_randn((n,), "cuda", seed=1234, step=1, which=0),
              basis.device_randn((n,), "cuda", seed=1235, step=0, which=0)]
    for b in others:
        assert not torch.equal(a, b)
        assert abs(float((a * b).mean())) < 5.0 / n ** 0.5                   # uncorrelated streams
    assert abs(float((a[:-1] * a[1:]).mean())) < 5.0 / n ** 0.5              # and no lag-1 correlation within one
    u = basis.device_randn((n,), "cuda", seed=7, uniform=True)
    assert float(u.min()) > 0.0 and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 1e-3 and abs(float(u.var()) - 1.0 / 12) < 1e-3


@pytest.mark.gpu
def test_fused_update_kernel_against_the_formulas():
    """glowk_basis_update == run_basis_sep.py:163-181 written out in float64, with injected noise; with its own RNG it is
    bitwise the injected-noise result for the draws glowk_random reports; a non-finite gradient raises the flag."""
    rng = np.random.default_rng(5)
    shape = (7, 16, 12, 1)                                   # 1344 elements: not a multiple of the 1024-element workgroup
    x1, x2, mixed = rng.uniform(-80, 10, (3,) + shape)
    g1, g2 = rng.normal(0, 5, (2,) + shape)
    e1, e2 = rng.standard_normal((2,) + shape)
    eta, lam = 2e-5 * 37.0, 1.0 / 0.3 ** 2
    mix = basis_ref.g_db(x1, x2)
    m1, m2 = basis_ref.grad_g_db(x1, x2)
    r1 = x1 + eta * (g1 + lam * m1 * (mixed - mix)) + np.sqrt(2 * eta) * e1
    r2 = x2 + eta * (g2 + lam * m2 * (mixed - mix)) + np.sqrt(2 * eta) * e2
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()   # noqa: E731
    np.testing.assert_allclose(basis.mixing_db(dev(x1), dev(x2)).cpu().numpy(), mix, rtol=2e-6, atol=2e-5)
    y1, y2 = dev(x1), dev(x2)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    basis.langevin_update(dev(mixed), y1, y2, dev(g1), dev(g2), eta, lam, dev(e1), dev(e2), nonfinite=flag)
    np.testing.assert_allclose(y1.cpu().numpy(), r1, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(y2.cpu().numpy(), r2, rtol=1e-5, atol=1e-4)
    assert int(flag.item()) == 0
    # device RNG: same result as injecting the draws the RNG reports for (seed, step, source)
    z1, z2 = dev(x1), dev(x2)
    basis.langevin_update(dev(mixed), z1, z2, dev(g1), dev(g2), eta, lam, seed=99, step=5)
    w1, w2 = dev(x1), dev(x2)
    basis.langevin_update(dev(mixed), w1, w2, dev(g1), dev(g2), eta, lam, basis.device_randn(shape, "cuda", 99, 5, 0),
                          basis.device_randn(shape, "cuda", 99, 5, 1))
    assert torch.equal(z1, w1) and torch.equal(z2, w2) and not torch.equal(z1, y1)
    # the reference's NaN asserts (run_basis_sep.py:183-191)
    gbad = dev(g1)
    gbad[3, 2, 1, 0] = float("nan")
    basis.langevin_update(dev(mixed), dev(x1), dev(x2), gbad, dev(g2), eta, lam, nonfinite=flag)
    assert int(flag.item()) == 1


@pytest.mark.gpu
def test_inner_loop_with_device_rng_is_reproducible_and_leaves_inputs_alone():
    from audiosourcesep_amd.flow_models.flow_glow import GlowFlow
    from audiosourcesep_amd.synthetic import calibrated_engine
    cfg = GlowConfig(H=16, W=16, C=1, L=2, K=2, F=128)
    e1, _ = calibrated_engine(cfg, device=0, init_tiles=16, seed=1)
    e2, _ = calibrated_engine(cfg, device=0, init_tiles=16, seed=2)
    m1, m2 = GlowFlow(e1), GlowFlow(e2)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()   # noqa: E731
    x1, x2 = dev(synthetic_mel_tiles(5, cfg, seed=12)), dev(synthetic_mel_tiles(5, cfg, seed=13))
    mixed = basis.mixing_db(dev(synthetic_mel_tiles(5, cfg, seed=10)), dev(synthetic_mel_tiles(5, cfg, seed=11)))
    x1c, x2c = x1.clone(), x2.clone()
    sig = basis.get_sigmas(1.0, 0.01, 10)
    a1, a2 = basis.basis_inner_loop(mixed, x1, x2, m1, m2, 9, sig, T=3, seed=4, debug=True)
    b1, b2 = basis.basis_inner_loop(mixed, x1, x2, m1, m2, 9, sig, T=3, seed=4)
    c1, _ = basis.basis_inner_loop(mixed, x1, x2, m1, m2, 9, sig, T=3, seed=5)
    assert torch.equal(x1, x1c) and torch.equal(x2, x2c)
    assert torch.equal(a1, b1) and torch.equal(a2, b2) and not torch.equal(a1, c1)
    assert torch.isfinite(a1).all() and float((a1 - x1).abs().max()) > 1e-3


@pytest.mark.gpu
def test_noise_streams_do_not_depend_on_sharding_and_never_repeat():
    """Round-2 advisor finding: train_step(noise_std > 0) reused one noise tensor on every step and rank, and sharded BASIS ranks
    all drew the same Langevin noise.  Now the element offset of a shard is part of the Philox counter: a rank holding tiles
    [a, b) draws exactly what one process draws for those tiles (so results do not change with the world size), two ranks never
    share a draw, and the flow counts its own noisy steps (train_noisy_glow.py:31: fresh noise every step, every replica)."""
    from audiosourcesep_amd.flow_models.flow_glow import GlowFlow
    from audiosourcesep_amd.synthetic import calibrated_engine
    cfg = GlowConfig(H=16, W=16, C=1, L=2, K=2, F=128)
    E = cfg.H * cfg.W * cfg.C
    whole = basis.device_randn((12, cfg.H, cfg.W, cfg.C), "cuda", seed=5, step=9, which=2)
    for a, b in ((0, 5), (5, 9), (9, 12)):                                     # three ragged shards
        part = basis.device_randn((b - a, cfg.H, cfg.W, cfg.C), "cuda", seed=5, step=9, which=2, offset=a * E)
        assert torch.equal(part, whole[a:b])
    assert not torch.equal(whole[0:4], whole[4:8])                             # "rank 0" and "rank 1" of a 4-tile-per-rank job
    with pytest.raises(Exception):
        basis.device_randn((4,), "cuda", seed=5, offset=2)                     # offsets are multiples of 4 elements
    # the fused x + sigma * noise kernel draws from the same stream
    x = torch.from_numpy(synthetic_mel_tiles(12, cfg, seed=1)).cuda()
    y = basis.add_device_noise(x, 0.5, seed=5, step=9, which=2)
    assert torch.allclose(y, x + 0.5 * whole, rtol=0, atol=1e-5)               # (fma vs mul + add: one rounding apart)
    y1 = basis.add_device_noise(x[5:9], 0.5, seed=5, step=9, which=2, offset=5 * E)
    assert torch.equal(y1, y[5:9])
    # the Langevin update of a shard == the same tiles inside the whole batch (device RNG, no injected noise)
    rng = np.random.default_rng(2)
    mk = lambda: torch.from_numpy(rng.uniform(-80, 10, (12, cfg.H, cfg.W, cfg.C)).astype(np.float32)).cuda()   # noqa: E731
    mixed, x1, x2, g1, g2 = mk(), mk(), mk(), mk() * 0.01, mk() * 0.01
    w1, w2 = x1.clone(), x2.clone()
    basis.langevin_update(mixed, w1, w2, g1, g2, 1e-3, 4.0, seed=77, step=3)
    s1, s2 = x1[5:9].clone().contiguous(), x2[5:9].clone().contiguous()
    basis.langevin_update(mixed[5:9].contiguous(), s1, s2, g1[5:9].contiguous(), g2[5:9].contiguous(), 1e-3, 4.0, seed=77, step=3, offset=5 * E)
    assert torch.equal(s1, w1[5:9]) and torch.equal(s2, w2[5:9])
    # GlowFlow.train_step: consecutive noisy steps use consecutive RNG steps (engine-owned counter), never the same draw twice
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=8)
    flow = GlowFlow(eng)
    seen = []
    orig = basis.add_device_noise

    def spy(x, sigma, seed, step=0, which=2, offset=0):
        out = orig(x, sigma, seed, step, which, offset)
        seen.append((int(step), int(offset), (out - x).clone()))
        return out

    basis.add_device_noise = spy
    try:
        xs = x[:4].contiguous()
        flow.train_step(xs, lr=1e-5, noise_std=0.1)
        flow.train_step(xs, lr=1e-5, noise_std=0.1)
        flow.train_step(xs, lr=1e-5, noise_std=0.1, tile_offset=4)             # what rank 1 of a 4-tile-per-rank job passes
    finally:
        basis.add_device_noise = orig
    assert [s[0] for s in seen] == [0, 1, 2] and [s[1] for s in seen] == [0, 0, 4 * E]
    assert not torch.equal(seen[0][2], seen[1][2]) and not torch.equal(seen[1][2], seen[2][2])


@pytest.mark.gpu
def test_log_prob_sum_is_the_fp64_sum_in_a_fixed_order():
    """glowk_log_prob_sum: the fp64 summed log-likelihood leaves the engine itself (no tensor-library reduction between the
    engine's kernels and the all-reduce of train_glow.py:52-54); chunked batches accumulate; bitwise repeatable."""
    from audiosourcesep_amd.synthetic import calibrated_engine
    from audiosourcesep_amd.distributed import sharded_log_prob
    cfg = GlowConfig(H=16, W=16, C=1, L=2, K=2, F=128)
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=8)
    x = torch.from_numpy(synthetic_mel_tiles(37, cfg, seed=4)).cuda()
    lp, tot = eng.log_prob_sum(x)
    assert tot.dtype == torch.float64 and tot.shape == (1,)
    assert torch.equal(lp, eng.log_prob(x))
    ref = lp.double().sum()
    assert abs(float(tot[0] - ref)) <= 1e-12 * abs(float(ref))
    lp2, tot2 = eng.log_prob_sum(x)
    assert torch.equal(tot, tot2)
    eng._max_tiles_cap = 8                                                     # five chunks: the engine accumulates in order
    lp3, tot3 = eng.log_prob_sum(x)
    eng._max_tiles_cap = None
    assert torch.equal(lp3, lp) and abs(float(tot3[0] - ref)) <= 1e-12 * abs(float(ref))
    # the sharding helper takes the engine and hands back the same total (one rank: no process group)
    lp4, t4 = sharded_log_prob(eng, x)
    assert torch.equal(lp4, lp) and float(t4) == float(tot[0])
    _, t0 = eng.log_prob_sum(x[:0])
    assert float(t0[0]) == 0.0
