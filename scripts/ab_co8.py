"""A/B of the co-resident form at the 8-channel level (k_net_h3c<4, 72, ...>: five conv3 units, two partial P buffers; default) against the
eight-wave kernels there (GLOWK_CO8_OFF=1), alternating in one process; config B, 64 x 64 tiles, f16x3.
    python scripts/ab_co8.py [rounds=3]
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
lib = _lib.load()
eng, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=32)
eng.set_precision(_lib.PREC_F16X3)
eng.set_range_policy("error")
base = torch.from_numpy(synthetic_mel_tiles(64, CONFIG_B, seed=5)).cuda()
xs = {n: base.repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous() for n in (96, 128, 160, 256, 512, 1024)}


def setco8(on):
    if on:
        os.environ.pop("GLOWK_CO8_OFF", None)
    else:
        os.environ["GLOWK_CO8_OFF"] = "1"
    lib.glowk_reload_env()


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


work = {}
for n in xs:
    work["log_prob %4d tiles" % n] = (lambda n=n: eng.log_prob(xs[n]), max(3, 600 // n))
for n in (128, 256):
    work["log_prob_grad %4d tiles" % n] = (lambda n=n: eng.log_prob_grad(xs[n]), 4)
res = {k: {True: [], False: []} for k in work}
out = {}
for r in range(rounds):
    for q in (False, True):
        setco8(q)
        for k, (fn, reps) in work.items():
            res[k][q].append(timeit(fn, reps))
        if r == 0:
            b = eng.kernel_families()
            lp, z = eng.log_prob(xs[256], return_latent=True)
            out[q] = (lp.clone(), z.clone(), eng.log_prob_grad(xs[128])[1].double().cpu(), eng.kernel_families()["co_resident"] - b["co_resident"])
setco8(True)
for k in work:
    a, b = sorted(res[k][False]), sorted(res[k][True])
    print("%-26s eight-wave at level 1 %8.3f ms (min %8.3f)   co-resident %8.3f ms (min %8.3f)   %+.1f %%" %
          (k, a[len(a) // 2], a[0], b[len(b) // 2], b[0], 100 * (b[len(b) // 2] / a[len(a) // 2] - 1)), flush=True)
print("log_prob / latent bit for bit equal: %s / %s; gradient rel l2 %.2e; co-resident launches in the checked calls: %d (off) %d (on)"
      % (torch.equal(out[False][0], out[True][0]), torch.equal(out[False][1], out[True][1]),
         ((out[True][2] - out[False][2]).norm() / out[False][2].norm()).item(), out[False][3], out[True][3]))
