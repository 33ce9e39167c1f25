"""A/B in one process: glowk_param_grad's host share (ActNorm / 1x1 chain rule) beside the last level's weight-gradient GEMMs (default)
against a host join of the caller's stream first (GLOWK_PG_JOIN=1: the flow before the side stream); whole training steps, config B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
for n in (32, 256):
    eng, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=max(n, 64))
    x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B)).cuda()
    eng.set_precision(_lib.PREC_F16X3)
    g = torch.empty(eng.param_vector_size, device="cuda")
    def step():
        eng.param_grad(x, -1.0 / n, g)
        eng.apply_gradients(g, "adamax", 1e-6)
    res = {0: [], 1: []}
    for r in range(3):
        for join in (1, 0):
            if join: os.environ["GLOWK_PG_JOIN"] = "1"
            else: os.environ.pop("GLOWK_PG_JOIN", None)
            for _ in range(2): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10 if n == 32 else 4): step()
            torch.cuda.synchronize()
            res[join].append((time.perf_counter() - t0) / (10 if n == 32 else 4) * 1e3)
    os.environ.pop("GLOWK_PG_JOIN", None)
    a, b = sorted(res[1]), sorted(res[0])
    print("tiles %4d: step with the host join %.3f ms (min %.3f)   chain rule beside the GEMMs %.3f ms (min %.3f)   %+.1f %%" % (n, a[1], a[0], b[1], b[0], 100 * (b[1] / a[1] - 1)), flush=True)
    eng.close()
