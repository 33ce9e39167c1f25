"""Probe for the activation scale of the fp16x3 kernels: accuracy vs the fp64 oracle, and whether the BASIS state stays finite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audiosourcesep_amd import basis, _lib
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.flow_models.flow_glow import GlowFlow
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
from oracle import glowref as R
e1, params = calibrated_engine(CONFIG_B, device=0, init_tiles=64)
e2, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=64, seed=4048)
x = synthetic_mel_tiles(2, CONFIG_B, seed=9)
ref = R.log_prob(x.astype(np.float64), R.cast_params(params, np.float64), CONFIG_B.as_dict())
e1.set_precision(_lib.PREC_F16X3); e2.set_precision(_lib.PREC_F16X3)
lp = e1.log_prob(torch.from_numpy(x).cuda()).cpu().numpy()
print("log_prob rel err vs fp64: %.2e" % np.max(np.abs(lp - ref) / np.abs(ref)))
xl = torch.from_numpy(synthetic_mel_tiles(256, CONFIG_B, seed=10)).cuda()
l16 = e1.log_prob(xl); e1.set_precision(_lib.PREC_F32); l32 = e1.log_prob(xl); e1.set_precision(_lib.PREC_F16X3)
print("max rel diff vs fp32 kernels over 256 tiles: %.2e" % float(((l16 - l32).abs() / l32.abs()).max()))
m1, m2 = GlowFlow(e1), GlowFlow(e2)
if os.environ.get("PROBE_F32"):
    m1.set_precision("f32"); m2.set_precision("f32")
n = 30
a = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=1)).cuda(); b = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=2)).cuda()
mixed = basis.mixing_db(a, b)
sig = basis.get_sigmas(1.0, 0.01, 10)
x1 = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=3)).cuda(); x2 = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=4)).cuda()
for t in range(20):
    x1, x2 = basis.basis_inner_loop(mixed, x1, x2, m1, m2, 9, sig, T=1)
    g1 = basis.compute_grad_logprob(x1, m1)
    print("   step %d: x range [%.1f, %.1f], max |grad log p| %.3e" % (t + 1, float(x1.min()), float(x1.max()), float(g1.abs().max())), flush=True)
    if not (torch.isfinite(x1).all() and torch.isfinite(x2).all()):
        print("BASIS state non-finite after %d steps" % (t + 1)); break
else:
    print("BASIS state finite after 20 steps; range [%.1f, %.1f] dB" % (float(torch.minimum(x1.min(), x2.min())), float(torch.maximum(x1.max(), x2.max()))))
