"""A/B of the co-resident form (default; GLOWK_CO_OFF=1 switches it off: k_net_h3c, two four-wave workgroups per CU) against the one-workgroup-per-CU kernel on the
headline workload, alternating in ONE process on one box: passes/s and the level-0 kernel's average duration (HIP events).

    python scripts/ab_co.py [tiles=1024] [rounds=4] [steps=20]
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lib = _lib.load()
eng, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=n)
eng.set_range_policy("error")
x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=1234)).cuda()
lp = torch.empty(n, device="cuda")


def setenv(name, on):
    if on:
        os.environ[name] = "1"
    else:
        os.environ.pop(name, None)
    lib.glowk_reload_env()


def run(prec, co, nofuse):
    eng.set_precision(prec)
    setenv("GLOWK_CO_OFF", not co)
    setenv("GLOWK_NO_FUSE", nofuse)
    for _ in range(3):
        eng.log_prob(x, out=lp)
    torch.cuda.synchronize()
    eng.profile_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.log_prob(x, out=lp)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    pr = eng.profile_end()
    return n * steps / el, pr[0][0] / max(pr[0][1], 1), lp.clone()


ref = {}
PRECS = (("f16x3", _lib.PREC_F16X3),) if os.environ.get("GLOWK_AB_F16X3_ONLY") else (("f16x3", _lib.PREC_F16X3), ("f16x2", _lib.PREC_F16X2))
for prec_name, prec in PRECS:
    for nofuse in ((False,) if os.environ.get("GLOWK_AB_FUSED_ONLY") else (False, True)):
        rows = {False: [], True: []}
        for r in range(rounds):
            for co in (False, True):
                v, ms0, out = run(prec, co, nofuse)
                rows[co].append((v, ms0))
                key = (prec_name, nofuse)
                if key not in ref:
                    ref[key] = out
                else:
                    d = float(((out - ref[key]).abs() / ref[key].abs()).max())
                    assert d < (3e-7 if prec_name == 'f16x3' else 2e-5) or os.environ.get('GLOWK_AB_NOCHECK'), d
        for co in (False, True):
            vs = sorted(v for v, _ in rows[co])
            ms = sorted(m for _, m in rows[co])
            print("%s %s %s: passes/s median %.0f (min %.0f max %.0f); level-0 kernel median %.4f ms" %
                  (prec_name, "P-to-HBM" if nofuse else "fused   ", "CO-RESIDENT" if co else "one-per-CU ", vs[len(vs) // 2], vs[0], vs[-1], ms[len(ms) // 2]), flush=True)
setenv("GLOWK_CO_OFF", False)
setenv("GLOWK_NO_FUSE", False)
