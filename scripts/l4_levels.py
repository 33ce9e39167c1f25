"""Diagnostic: per-level k_net time of a 4-level model, forward vs log_prob_grad (the last level's backward is the
32-channel shape that still loads its small-conv operands from global memory)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
cfg = GlowConfig(H=64, W=64, C=1, L=4, K=int(os.environ.get("K", "8")), F=512)
for prec in (1, 0):
    eng, _ = calibrated_engine(cfg, device=0)
    eng.set_precision(prec)
    for n in [int(v) for v in os.environ.get("NS", "30,1024").split(",")]:
        x = torch.from_numpy(synthetic_mel_tiles(n, cfg)).cuda(); eng.reserve(n)
        for name, f in (("log_prob", lambda: eng.log_prob(x)), ("log_prob_grad", lambda: eng.log_prob_grad(x))):
            f(); f(); torch.cuda.synchronize()
            eng.profile_begin(); f(); torch.cuda.synchronize(); pr = eng.profile_end()
            print("prec=%d N=%4d %-14s" % (prec, n, name), "  ".join("L%d %.3f ms/%d" % (i, ms, k) for i, (ms, k) in enumerate(pr)), flush=True)
