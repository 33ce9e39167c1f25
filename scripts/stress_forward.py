"""Determinism stress of log_prob / forward / inverse on config B (all three arithmetics, three batch sizes; ONLY=f16x2 restricts): every repeat bitwise
equal to the first."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
eng, _ = calibrated_engine(CONFIG_B, device=0)
for n in (1024, 64, 7):
    x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=n)).cuda()
    for name, prec, reps in (("fp32", _lib.PREC_F32, 60), ("f16x3", _lib.PREC_F16X3, 200), ("f16x2", _lib.PREC_F16X2, 200)):
        if os.environ.get("ONLY") and name not in os.environ["ONLY"].split(","):
            continue
        eng.set_precision(prec)
        lp0 = eng.log_prob(x); z0, ld0 = eng.forward(x); x0 = eng.inverse(z0)
        bad = 0
        for _ in range(reps if n == 1024 else 400):
            lp = eng.log_prob(x); z, ld = eng.forward(x); xr = eng.inverse(z)
            bad += 0 if (torch.equal(lp, lp0) and torch.equal(z, z0) and torch.equal(xr, x0)) else 1
        print("config B N=%d %-5s: %d mismatching repeats (log_prob, forward, inverse)" % (n, name, bad), flush=True)
