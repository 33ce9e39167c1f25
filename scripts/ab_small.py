"""A/B of the small-grid form k_net_h3q (default) against the half-wave form (GLOWK_Q_OFF=1) at the reference's batch sizes, alternating in
one process: log_prob (32 tiles), log_prob_grad (30 tiles), param_grad (32 tiles); config B, f16x3.

    python scripts/ab_small.py [rounds=3] [reps=30]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from audiosourcesep_amd import _lib  # noqa: E402
from audiosourcesep_amd.config import CONFIG_B  # noqa: E402
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lib = _lib.load()
eng, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=32)
eng.set_precision(_lib.PREC_F16X3)
eng.set_range_policy("error")
x30 = torch.from_numpy(synthetic_mel_tiles(30, CONFIG_B, seed=3)).cuda()
x32 = torch.from_numpy(synthetic_mel_tiles(32, CONFIG_B, seed=4)).cuda()
eng.reserve(32, with_grad=True)
eng.param_grad(x32, -1.0 / 32)
eng.param_grad(x32, -1.0 / 32)


def setq(on):
    if on:
        os.environ.pop("GLOWK_Q_OFF", None)
    else:
        os.environ["GLOWK_Q_OFF"] = "1"
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


work = {"log_prob 32 tiles": lambda: eng.log_prob(x32), "log_prob_grad 30 tiles": lambda: eng.log_prob_grad(x30),
        "param_grad 32 tiles": lambda: eng.param_grad(x32, -1.0 / 32)}
res = {k: {True: [], False: []} for k in work}
for r in range(rounds):
    for q in (False, True):
        setq(q)
        for k, fn in work.items():
            res[k][q].append(timeit(fn))
setq(True)
for k in work:
    a, b = sorted(res[k][False]), sorted(res[k][True])
    print("%-24s half-wave %.3f ms (min %.3f)   all-conv1-first %.3f ms (min %.3f)   %+.1f %%" %
          (k, a[len(a) // 2], a[0], b[len(b) // 2], b[0], 100.0 * (b[len(b) // 2] / a[len(a) // 2] - 1.0)))
