"""Determinism stress of the c = 32 forward instance of k_net_h3s (18 output row blocks in three groups, nine conv3 chunks per
pass through the three LDS slots): L = 4, n_filters = 512, 32x64 tiles, three batch sizes (split launches, one ragged
workgroup, the unsplit form), f16x3 and two-term; every repeat of log_prob / forward / inverse bitwise equal to the first."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
cfg = GlowConfig(H=32, W=64, C=1, L=4, K=3, F=512)
eng, _ = calibrated_engine(cfg, device=0, init_tiles=16)
total = 0
for n in (2500, 150, 3):
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=n)).cuda()
    for name, prec in (("f16x3", _lib.PREC_F16X3), ("f16x2", _lib.PREC_F16X2)):
        eng.set_precision(prec)
        lp0 = eng.log_prob(x); z0, ld0 = eng.forward(x); x0 = eng.inverse(z0)
        bad = 0
        for _ in range(100 if n == 2500 else 500):
            lp = eng.log_prob(x); z, ld = eng.forward(x); xr = eng.inverse(z)
            bad += 0 if (torch.equal(lp, lp0) and torch.equal(z, z0) and torch.equal(xr, x0)) else 1
        total += bad
        print("L=4 F=512 N=%d %-5s: %d mismatching repeats (log_prob, forward, inverse)" % (n, name, bad), flush=True)
sys.exit(1 if total else 0)
