"""Diagnostic: range margins of consecutive split training sweeps (forward: largest gathered input / limit; backward: the same for
the uniformly pre-scaled g_o of the sweep, GLOWK_PROBE_RAW_BWD=1) and which sweeps fell back."""
import os, sys
os.environ["GLOWK_PROBE_RAW_BWD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=n)
eng.set_precision(_lib.PREC_F16X3)
eng.set_range_policy("fallback")
x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=1234)).cuda()
for it in range(10):
    fb0 = eng.range_status(sync=False)[1]
    eng.range_probe_begin()
    lp, g = eng.param_grad(x, -1.0 / n)
    m = eng.range_probe_end()
    fb = eng.range_status(sync=False)[1] - fb0
    print("step %d: loss %.2f |g|max %.3e  margins fwd %.4f bwd(raw) %.4f  fallback %d" % (it, float(-lp.mean()), float(g.abs().max()), m[0], m[1], fb), flush=True)
    eng.apply_gradients(g, "adamax", 1e-4)
