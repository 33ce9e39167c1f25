"""Bitwise-repeat stress of the co-resident form (the default since round 4; GLOWK_CO_OFF=1 switches it off) on the headline workload: N log_prob passes over the same resident batch
(each pass = 32 fused level-0 launches of k_net_h3c, two workgroups per CU), every result compared bit for bit with the first, and the
first compared with the one-workgroup-per-CU kernel.  A synchronisation hole in the ring protocol shows as tiles that differ from run to
run (glowk_co.h: the first build had one -- LDS reads still in flight across the barrier that frees their slot).

    python scripts/stress_co.py [passes=300] [tiles=1024] [precision=f16x3] [grad]      (grad: log_prob_grad -- the saving and backward modes; at
                                                                                          30 tiles the one-pass-per-workgroup form)
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

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
prec = {"f16x3": _lib.PREC_F16X3, "f16x2": _lib.PREC_F16X2}[sys.argv[3] if len(sys.argv) > 3 else "f16x3"]
grad = len(sys.argv) > 4 and sys.argv[4] == "grad"
lib = _lib.load()
eng, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=min(n, 256))
eng.set_range_policy("error")
eng.set_precision(prec)
x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=1234)).cuda()


def setenv(name, on):
    if on:
        os.environ[name] = "1"
    else:
        os.environ.pop(name, None)
    lib.glowk_reload_env()


setenv("GLOWK_CO_OFF", True)
call = (lambda: eng.log_prob_grad(x)) if grad else (lambda: eng.log_prob(x, return_latent=True))
ref_lp, ref_z = call()
ref_lp, ref_z = ref_lp.clone(), ref_z.clone()
setenv("GLOWK_CO_OFF", False)
before = eng.kernel_families()
first_lp, first_z = call()
first_lp, first_z = first_lp.clone(), first_z.clone()
fam = {k: v - before[k] for k, v in eng.kernel_families().items()}
assert fam["co_resident"] >= (2 if grad else 1) * CONFIG_B.K, fam      # (level 0; the 8-channel level too where its grid is large enough)
d = float(((first_lp - ref_lp).abs() / ref_lp.abs()).max())
print("co-resident vs one-per-CU: max rel diff of log_prob %.2e, max |d second output| / max %.2e" % (d, float((first_z - ref_z).abs().max() / ref_z.abs().max())), flush=True)
assert d < (2e-7 if prec == _lib.PREC_F16X3 else 2e-5), d
bad = 0
t0 = time.time()
for i in range(passes):
    lp, z = call()
    if not (torch.equal(lp, first_lp) and torch.equal(z, first_z)):
        bad += 1
        off = torch.nonzero(lp != first_lp).flatten().tolist()
        print("pass %d differs: tiles %s" % (i, off[:20]), flush=True)
    if (i + 1) % 100 == 0:
        print("%d passes (%d level-0 launches), %d differing, %.0f s" % (i + 1, (i + 1) * CONFIG_B.K, bad, time.time() - t0), flush=True)
setenv("GLOWK_CO_OFF", False)
print("RESULT passes %d tiles %d differing %d" % (passes, n, bad))
sys.exit(1 if bad else 0)
