"""Diagnostic: repeated log_prob / log_prob_grad at a small batch (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
n = int(os.environ.get("GLOWK_AB_N", "32")); grad = os.environ.get("GLOWK_GRAD") == "1"
eng, _ = calibrated_engine(CONFIG_B, device=0)
eng.set_precision(int(os.environ.get("GLOWK_PREC", "1")))
if os.environ.get("GLOWK_IGNORE_RANGE") == "1":   # diagnostic builds compute wrong numbers on purpose: time them, do not re-run them in fp32
    eng.set_range_policy("ignore")
x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B)).cuda(); eng.reserve(n)
f = (lambda: eng.log_prob_grad(x)) if grad else (lambda: eng.log_prob(x))
for _ in range(3): f()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20): f()
torch.cuda.synchronize(); print("N=%d grad=%s: %.3f ms per call" % (n, grad, (time.time() - t0) / 20 * 1e3))
