"""BASELINE config 5 for real (round-2 verdict, "Next" #1): two noise-conditioned Glow priors are MADE with the repo's own training
step on the reference's real tiles (fine_tune_ladder = train_noisy_glow.py:309-358 on GlowFlow.train_step(noise_std=sigma)), kept
resident per sigma, and the BASIS sigma ladder (run_basis_sep.py:217-260, T = 100 Langevin steps per level from the reference's
uniform start) runs on the 30 mixture tiles in the split arithmetic under GLOWK_RANGE_ERROR -- the first TRAINED checkpoints
through the f16x3 kernels.  Shortened: K = 8 steps per level instead of 32, four sigma levels instead of ten (the largest is where
the reference's uniform start is typical for the noised data), 600 training steps per prior."""
import os

import numpy as np
import pytest
import torch

from audiosourcesep_amd import basis, _lib
from audiosourcesep_amd.flow_models.flow_builder import build_glow
from audiosourcesep_amd.noise_conditioned import fine_tune_ladder, psnr_db, db_schedule

pytestmark = pytest.mark.gpu
TILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "basis_real_tiles.npz")
MEL = dict(data_type="melspec", minval=-100.0, maxval=20.0, use_logit=False)


@pytest.fixture(scope="module")
def trained():
    """Both priors trained once per module (the ladder of train_noisy_glow.py:309-358 on the repo's own training step)."""
    f = np.load(TILES)
    gt1, gt2, mixed = (torch.from_numpy(f[k].astype(np.float32))[..., None].cuda() for k in ("gt1", "gt2", "mixed"))
    assert gt1.shape == (30, 96, 64, 1)
    priors, ladders = [], []
    sig_db = delta_db = None
    for i, gt in enumerate((gt1, gt2)):
        flow = build_glow(gt, [96, 64, 1], L=3, K=8, n_filters=512, learntop=True, seed=100 + i, precision="f16x3", actnorm_init="runtime", **MEL)
        flow.engine.set_range_policy("fallback")       # training may repeat a sweep on the exact kernels (counted below); the chain may not
        sig_db, delta_db = db_schedule(flow.cfg, sigma1=0.3, sigmaL=0.01, num_classes=4)       # 36 -> 1.2 dB; delta 2e-5 -> 0.288 dB^2
        models, losses = fine_tune_ladder(flow, gt, sig_db, [300, 100, 100, 100], lr=1e-3, seed=7 + i)
        for s in sig_db:
            l = losses[float(s)]
            assert np.isfinite(l).all() and l[-1] < l[0], (i, float(s), l[0], l[-1])
        # the split sweep carried the training: only the very first sweep (gradient scale still unknown) may have been repeated
        tripped, fallbacks = flow.engine.range_status()
        assert not tripped and fallbacks <= 2, (i, fallbacks)
        bits = losses[float(sig_db[-1])][-1] / (96 * 64 * np.log(2.0))
        print("prior %d: %d fallback sweep(s) of 600; loss per level %s; %.2f bits/dim at sigma_L" %
              (i, fallbacks, ["%.0f->%.0f" % (losses[float(s)][0], losses[float(s)][-1]) for s in sig_db], bits))
        priors.append(flow)
        ladders.append(models)
    return dict(f=f, gt=(gt1, gt2), mixed=mixed, priors=priors, ladders=ladders, sig_db=sig_db, delta_db=delta_db)


def test_trained_priors_match_the_fp64_oracle_in_both_arithmetics(trained):
    """Round-3 verdict, missing #1: TRAINED weights (wider dynamic range than synthetic_params: real ActNorm scales, real conv
    kernels after 600 Adamax steps) against the oracle -- not properties.  For each prior, its noisiest level (sigma_1) on the
    noised tiles it was trained on and its cleanest level (sigma_L) on the real tiles: the variables the engine reports go into
    oracle/glowref.py (fp64, NumPy) for log_prob and oracle/glowref_torch.py (fp64 autograd; compute_grad_logprob of
    run_basis_sep.py:73-79) for the input gradient; the engine runs the exact-fp32 kernels and the f16x3 split under
    GLOWK_RANGE_ERROR (a trip raises: no silent fp32 re-run can stand in for the split result)."""
    from oracle import glowref as R, glowref_torch as RT
    sig_db = trained["sig_db"]
    worst = {"f32": [0.0, 0.0], "f16x3": [0.0, 0.0]}
    rows = []
    for i, (gt, ladder) in enumerate(zip(trained["gt"], trained["ladders"])):
        for s, noisy in ((float(sig_db[0]), True), (float(sig_db[-1]), False)):
            flow = ladder[s]
            eng = flow.engine
            x = gt[[1, 11, 23]].contiguous()
            if noisy:
                x = basis.add_device_noise(x, s, seed=41 + i, step=0, which=2)
            xn = x.cpu().numpy().astype(np.float64)
            sd = flow.state_dict()
            cfg = flow.cfg.as_dict()
            lp_ref = R.log_prob(xn, R.cast_params(sd, np.float64), cfg)
            lp_t, g_ref = RT.log_prob_and_grad(xn, sd, cfg)
            np.testing.assert_allclose(lp_t, lp_ref, rtol=1e-10)            # the two restatements agree on the trained weights
            gmax = np.abs(g_ref).max(axis=(1, 2, 3), keepdims=True)
            gnorm = np.sqrt((g_ref ** 2).sum(axis=(1, 2, 3)))
            # yardstick: the oracle's OWN reverse mode in float32 against its float64 one -- what fp32 arithmetic costs on this flow
            # whatever the implementation (a trained flow amplifies forward rounding in its gradient; a ReLU whose pre-activation
            # lies within rounding of zero falls either way)
            _, g32 = RT.log_prob_and_grad(xn.astype(np.float32), sd, cfg, dtype=torch.float32)
            y_max = float(np.max(np.abs(g32 - g_ref) / gmax))
            y_l2 = float(np.max(np.sqrt(((g32 - g_ref) ** 2).sum(axis=(1, 2, 3))) / gnorm))
            policy = int(eng.lib.glowk_get_range_policy(eng.h))
            prec = eng.get_precision()
            eng.set_range_policy("error")
            try:
                for name, mode, lp_tol in (("f32", _lib.PREC_F32, 1e-6), ("f16x3", _lib.PREC_F16X3, 5e-6)):
                    eng.set_precision(mode)
                    lp = eng.log_prob(x).cpu().numpy().astype(np.float64)
                    lp2, dx = eng.log_prob_grad(x)
                    lp2, dx = lp2.cpu().numpy().astype(np.float64), dx.cpu().numpy().astype(np.float64)
                    e_lp = float(np.max(np.abs(lp - lp_ref) / np.abs(lp_ref)))
                    e_lp2 = float(np.max(np.abs(lp2 - lp_ref) / np.abs(lp_ref)))
                    err = np.abs(dx - g_ref) / gmax
                    e_g = float(err.max())
                    e_l2 = float(np.max(np.sqrt(((dx - g_ref) ** 2).sum(axis=(1, 2, 3))) / gnorm))
                    n_big = int((err > 2e-4).sum())
                    print("prior %d sigma %.2f dB (%s tiles) %s: log_prob rel err %.2e (saving pass %.2e); input gradient: max %.2e of max |g|, "
                          "%d of %d entries above 2e-4, per-tile L2 %.2e  [oracle float32 vs float64: max %.2e, L2 %.2e]  (log_prob %.1f .. %.1f, "
                          "max |g| %.3g)" % (i, s, "noised" if noisy else "real", name, e_lp, e_lp2, e_g, n_big, err.size, e_l2, y_max, y_l2,
                                             lp_ref.min(), lp_ref.max(), float(gmax.max())))
                    rows.append((i, s, name, e_lp, e_lp2, e_g, e_l2, y_max, y_l2))
                    worst[name][0] = max(worst[name][0], e_lp, e_lp2)
                    worst[name][1] = max(worst[name][1], e_g)
            finally:
                eng.set_precision(prec)
                eng.set_range_policy(policy)
            assert eng.range_status() == (False, 0)
    for i, s, name, e_lp, e_lp2, e_g, e_l2, y_max, y_l2 in rows:
        lp_tol = 1e-6 if name == "f32" else 5e-6
        assert e_lp < lp_tol and e_lp2 < lp_tol, (i, s, name, e_lp, e_lp2)
        # input gradient: 2e-4 of the tile's largest entry (the bar of every other gradient test) -- or, where float32 itself cannot
        # hold that on this checkpoint, three times what the oracle's own float32 reverse mode loses against float64
        assert e_g < max(2e-4, 3.0 * y_max), (i, s, name, e_g, y_max)
        assert e_l2 < max(5e-5, 3.0 * y_l2), (i, s, name, e_l2, y_l2)
    print("trained checkpoints vs fp64 oracle, worst over 2 priors x 2 noise levels x 3 tiles: f32 log_prob %.2e grad %.2e; "
          "f16x3 log_prob %.2e grad %.2e" % (worst["f32"][0], worst["f32"][1], worst["f16x3"][0], worst["f16x3"][1]))


def test_trained_noise_conditioned_priors_run_the_basis_ladder_in_f16x3(trained):
    f, (gt1, gt2), mixed = trained["f"], trained["gt"], trained["mixed"]
    priors, ladders, sig_db, delta_db = trained["priors"], trained["ladders"], trained["sig_db"], trained["delta_db"]
    # the chain: reference start (uniform over the data range, run_basis_sep.py:360-361), every per-sigma engine in f16x3 under
    # GLOWK_RANGE_ERROR (a trip raises), the static bound's margin measured along the way
    engines = [m[float(s)].engine for m in ladders for s in sig_db]
    for e in engines:
        assert e.get_precision() == _lib.PREC_F16X3
        e.set_range_policy("error")
        e.range_probe_begin()
    x1 = -100.0 + 120.0 * basis.device_randn(tuple(mixed.shape), mixed.device, seed=11, which=0, uniform=True)
    x2 = -100.0 + 120.0 * basis.device_randn(tuple(mixed.shape), mixed.device, seed=11, which=1, uniform=True)
    start = (psnr_db(x1, gt1), psnr_db(x2, gt2))
    y1, y2, arr = basis.basis_outer_loop(mixed, x1, x2, priors[0], priors[1], sig_db, restore_1=ladders[0], restore_2=ladders[1], T=100,
                                         delta=delta_db, debug=True, seed=3)      # debug: the reference's per-step NaN asserts
    assert len(arr["x1"]) == len(sig_db) + 1
    for lvl in range(len(sig_db) + 1):
        assert np.isfinite(arr["x1"][lvl]).all() and np.isfinite(arr["x2"][lvl]).all(), lvl
    margins = [e.range_probe_end() for e in engines]
    for e, (mf, mb) in zip(engines, margins):
        assert e.range_status() == (False, 0)                       # zero trips, zero fp32 re-runs: every step ran the split kernels
        assert 0.0 < mf < 1.0 and 0.0 < mb < 1.0, (mf, mb)          # (forward: largest gathered input / limit; backward: static)
    end = (psnr_db(y1, gt1), psnr_db(y2, gt2))
    ref = (psnr_db(f["x1"].astype(np.float32), f["gt1"].astype(np.float32)), psnr_db(f["x2"].astype(np.float32), f["gt2"].astype(np.float32)))
    print("separation PSNR (dB, peak = 120 dB range): start %.2f / %.2f -> end %.2f / %.2f; the reference's own shipped result on these "
          "tiles %.2f / %.2f; largest forward input / limit %.3f" % (start + end + ref + (max(m[0] for m in margins),)))
    assert end[0] > start[0] + 3.0 and end[1] > start[1] + 3.0
    # the chain's own noise makes the state at the smallest sigma a sample, not a point estimate; it must still sit near the mixture
    mix = basis.mixing_db(y1, y2)
    assert float((mix - mixed).abs().mean()) < 6.0
    # Sharding (BASELINE config 5 on several GPUs: each rank holds a slice of the mixture tiles, no collective inside the loop): a rank
    # that runs tiles [a, b) with tile_offset = a walks the chain the one-process run walks for those tiles -- the same noise (the
    # offset is part of the Philox counter) and the same priors; the gradients differ only by the launch forms a smaller batch picks
    # (another summation order, ~1e-7), which a short chain does not amplify beyond a few thousandths of a dB.
    sig2 = sig_db[-2:]
    s1 = -100.0 + 120.0 * basis.device_randn(tuple(mixed.shape), mixed.device, seed=21, which=0, uniform=True)
    s2 = -100.0 + 120.0 * basis.device_randn(tuple(mixed.shape), mixed.device, seed=21, which=1, uniform=True)
    w1, w2, _ = basis.basis_outer_loop(mixed, s1.clone(), s2.clone(), priors[0], priors[1], sig2, restore_1=ladders[0], restore_2=ladders[1], T=8,
                                       delta=delta_db, seed=5)
    worst, mean = 0.0, 0.0
    for a, b in ((0, 8), (8, 19), (19, 30)):                       # three ragged "ranks"
        r1, r2, _ = basis.basis_outer_loop(mixed[a:b].contiguous(), s1[a:b].clone(), s2[a:b].clone(), priors[0], priors[1], sig2,
                                           restore_1=ladders[0], restore_2=ladders[1], T=8, delta=delta_db, seed=5, tile_offset=a)
        worst = max(worst, float((r1 - w1[a:b]).abs().max()), float((r2 - w2[a:b]).abs().max()))
        mean = max(mean, float((r1 - w1[a:b]).abs().mean()), float((r2 - w2[a:b]).abs().mean()))
    # a shard run WITHOUT its offset draws other noise: sigma_L alone moves the state by ~1 dB per step
    q1, _, _ = basis.basis_outer_loop(mixed[8:19].contiguous(), s1[8:19].clone(), s2[8:19].clone(), priors[0], priors[1], sig2,
                                      restore_1=ladders[0], restore_2=ladders[1], T=8, delta=delta_db, seed=5, tile_offset=0)
    wrong = float((q1 - w1[8:19]).abs().mean())
    print("sharded chain (3 shards, 2 levels x T = 8) vs the whole batch: |difference| mean %.2e dB, max %.2e dB (an isolated ReLU decision "
          "that falls differently moves one tile); without the tile offset: mean %.2e dB" % (mean, worst, wrong))
    assert mean < 2e-3 and worst < 0.5 and wrong > 100 * mean
    for e in engines:
        assert e.range_status() == (False, 0)
