#!/usr/bin/env python3
"""Config 5 for real, probe form: train two noise-conditioned Glow priors on the reference's real tiles with the repo's own
training step (fine_tune_ladder), then run the BASIS sigma ladder on the 30 mixture tiles in f16x3 under GLOWK_RANGE_ERROR.
Prints per-level losses, range status, separation PSNR.  Tunables on the command line."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiosourcesep_amd import basis, _lib  # noqa: E402
from audiosourcesep_amd.flow_models.flow_builder import build_glow  # noqa: E402
from audiosourcesep_amd.noise_conditioned import fine_tune_ladder, psnr_db, db_schedule  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--K", type=int, default=8)
ap.add_argument("--F", type=int, default=512)
ap.add_argument("--levels", type=int, default=3)
ap.add_argument("--sigma1", type=float, default=0.1, help="normalised units (reference: 1.0)")
ap.add_argument("--sigmaL", type=float, default=0.01)
ap.add_argument("--pre", type=int, default=200, help="clean pre-training steps")
ap.add_argument("--steps", type=int, default=100, help="fine-tune steps per level")
ap.add_argument("--top-steps", type=int, default=0, help="fine-tune steps of the first (largest) sigma; 0: --steps")
ap.add_argument("--T", type=int, default=100)
ap.add_argument("--lr", type=float, default=1e-3)
ap.add_argument("--delta", type=float, default=2e-5)
ap.add_argument("--precision", default="f16x3")
ap.add_argument("--policy", default="fallback")
ap.add_argument("--chain-policy", default="error")
ap.add_argument("--init", default="reference", choices=["reference", "runtime"], help="ActNorm data-dependent init order")
args = ap.parse_args()

f = np.load(os.path.join(ROOT, "tests", "golden", "basis_real_tiles.npz"))
gt1, gt2, mixed = (torch.from_numpy(f[k].astype(np.float32))[..., None].cuda() for k in ("gt1", "gt2", "mixed"))
MEL = dict(data_type="melspec", minval=-100.0, maxval=20.0, use_logit=False)
flows = []
t0 = time.time()
for i, gt in enumerate((gt1, gt2)):
    fl = build_glow(gt, [96, 64, 1], L=3, K=args.K, n_filters=args.F, learntop=True, seed=100 + i, precision=args.precision, **MEL)
    if args.init == "runtime":
        fl.engine.actnorm_data_init(gt, runtime_order=True, raw_minibatch_quirk=False)
    fl.engine.set_range_policy(args.policy)
    fl.engine.range_probe_begin()
    lp, _ = fl.engine.log_prob_grad(gt)
    print("prior %d at init: log_prob/dim %.3f  input/limit (fwd, bwd) = %s  fallbacks %s" % (i, float(lp.mean()) / (96 * 64), fl.engine.range_probe_end(), fl.engine.range_status()))
    flows.append(fl)
print("built in %.1f s" % (time.time() - t0))
sig_db, delta_db = db_schedule(flows[0].cfg, args.sigma1, args.sigmaL, args.levels, args.delta)
print("sigmas (dB):", sig_db, "delta (dB^2): %.4f" % delta_db)
models = []
for i, (fl, gt) in enumerate(zip(flows, (gt1, gt2))):
    t0 = time.time()
    pre = []
    for t in range(args.pre):
        pre.append(fl.train_step(gt, lr=args.lr))
    pre = [float(v) for v in pre]
    fl.engine.range_probe_begin()
    fl.engine.param_grad(gt, -1.0 / 30)
    print("prior %d after pre-training: margins (fwd, bwd) %s, fallbacks so far %s" % (i, fl.engine.range_probe_end(), fl.engine.range_status()))
    if pre:
        print("prior %d clean pre-training: loss %.1f -> %.1f (bits/dim %.3f)" % (i, pre[0], pre[-1], pre[-1] / (96 * 64 * np.log(2))))
    m, losses = fine_tune_ladder(fl, gt, sig_db, [args.top_steps or args.steps] + [args.steps] * (len(sig_db) - 1), lr=args.lr, seed=7 + i)
    for s in sig_db:
        l = losses[float(s)]
        print("  sigma %.2f dB: loss %.1f -> %.1f" % (s, l[0], l[-1]))
    fl.engine.range_probe_begin()
    fl.engine.log_prob_grad(gt)
    print("  margins (fwd, bwd) %s range status: %s trained in %.1f s" % (fl.engine.range_probe_end(), fl.engine.range_status(), time.time() - t0))
    models.append(m)

x1 = -100.0 + 120.0 * basis.device_randn(tuple(mixed.shape), mixed.device, seed=11, which=0, uniform=True)   # run_basis_sep.py:360-361
x2 = -100.0 + 120.0 * basis.device_randn(tuple(mixed.shape), mixed.device, seed=11, which=1, uniform=True)
print("start PSNR: %.2f %.2f" % (psnr_db(x1, gt1), psnr_db(x2, gt2)))
t0 = time.time()
for k in (0, 1):
    for s_ in sig_db:
        models[k][float(s_)].engine.set_range_policy(args.chain_policy)
        models[k][float(s_)].engine.range_probe_begin()
try:
    y1, y2, arr = basis.basis_outer_loop(mixed, x1, x2, flows[0], flows[1], sig_db, restore_1=models[0], restore_2=models[1], T=args.T,
                                         delta=delta_db, debug=True, seed=3)
except (AssertionError, _lib.GlowkRangeError) as e:
    print("CHAIN FAILED:", type(e).__name__, e)
    for s_ in sig_db:
        for k in (0, 1):
            e_ = models[k][float(s_)].engine
            print("  model %d sigma %.2f chain margins (fwd, bwd) %s range status: %s" % (k, s_, e_.range_probe_end(), e_.range_status()))
    sys.exit(0)
torch.cuda.synchronize()
dt = time.time() - t0
print("chain: %d levels x %d steps in %.2f s = %.1f tile-steps/s" % (len(sig_db), args.T, dt, 30 * len(sig_db) * args.T / dt))
for lvl in range(len(sig_db) + 1):
    print("  after level %d: PSNR %.2f %.2f   finite %s" % (lvl, psnr_db(arr["x1"][lvl], gt1.cpu()), psnr_db(arr["x2"][lvl], gt2.cpu()),
                                                          bool(np.isfinite(arr["x1"][lvl]).all() and np.isfinite(arr["x2"][lvl]).all())))
for s in sig_db:
    for k in (0, 1):
        e = models[k][float(s)].engine
        print("  model %d sigma %.2f chain margins (fwd, bwd) %s range status (tripped, fallbacks): %s" % (k, s, e.range_probe_end(), e.range_status()))
print("reference's own result on these tiles: PSNR %.2f %.2f" % (psnr_db(f["x1"].astype(np.float32), f["gt1"].astype(np.float32)),
                                                                  psnr_db(f["x2"].astype(np.float32), f["gt2"].astype(np.float32))))
