"""A/B of the co-resident form on mid-size grids (more 128-pixel workgroups than CUs, fewer than four per CU: default) against the
eight-wave kernels there (GLOWK_CO_MID_OFF=1), alternating in one process; config B on the reference's 96 x 64 tiles, f16x3:
log_prob / log_prob_grad at 30 tiles (BASIS), log_prob at 48 and 64 tiles, param_grad at 48 tiles.

    python scripts/ab_mid.py [rounds=3] [reps=20]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from audiosourcesep_amd import _lib  # noqa: E402
from audiosourcesep_amd.config import GlowConfig  # noqa: E402
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _lib.load()
cfg = GlowConfig(H=96, W=64, C=1, L=3, K=32, F=512)
eng, _ = calibrated_engine(cfg, device=0, init_tiles=32)
eng.set_precision(_lib.PREC_F16X3)
eng.set_range_policy("error")
xs = {n: torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=3 + n)).cuda() for n in (30, 48, 64)}
eng.reserve(64, with_grad=True)


def setmid(on):
    if on:
        os.environ.pop("GLOWK_CO_MID_OFF", None)
    else:
        os.environ["GLOWK_CO_MID_OFF"] = "1"
    lib.glowk_reload_env()


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


work = {"log_prob 30 tiles": lambda: eng.log_prob(xs[30]), "log_prob_grad 30 tiles": lambda: eng.log_prob_grad(xs[30]),
        "log_prob 48 tiles": lambda: eng.log_prob(xs[48]), "log_prob 64 tiles": lambda: eng.log_prob(xs[64]),
        "log_prob_grad 64 tiles": lambda: eng.log_prob_grad(xs[64]), "param_grad 48 tiles": lambda: eng.param_grad(xs[48], -1.0 / 48)}
res = {k: {True: [], False: []} for k in work}
out = {}
for r in range(rounds):
    for q in (False, True):
        setmid(q)
        for k, fn in work.items():
            res[k][q].append(timeit(fn))
        if r == 0:
            out[q] = (eng.log_prob(xs[30]).double().cpu(), eng.log_prob_grad(xs[30])[1].double().cpu(), eng.param_grad(xs[48], -1.0 / 48)[1].double().cpu())
setmid(True)
for k in work:
    a, b = sorted(res[k][False]), sorted(res[k][True])
    print("%-24s eight-wave kernels %.3f ms (min %.3f)   co-resident %.3f ms (min %.3f)   %+.1f %%" %
          (k, a[len(a) // 2], a[0], b[len(b) // 2], b[0], 100 * (b[len(b) // 2] / a[len(a) // 2] - 1)), flush=True)
lp0, g0, p0 = out[False]
lp1, g1, p1 = out[True]
print("log_prob max |diff| %.2e (|lp| ~ %.0f); input gradient rel l2 %.2e; parameter gradient rel l2 %.2e; families %s"
      % ((lp1 - lp0).abs().max().item(), lp0.abs().mean().item(), ((g1 - g0).norm() / g0.norm()).item(), ((p1 - p0).norm() / p0.norm()).item(),
         eng.kernel_families()))
