"""A/B in one process: the 16-wave / 256 x 256 form of the split weight-gradient GEMM (conv2; default) against the 8-wave / 256 x 128 form
(GLOWK_WGRAD_16_OFF=1); glowk_param_grad of config B at 32 and 256 tiles, gradient vectors compared."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
lib = _lib.load()
for n in (32, 256):
    eng, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=max(n, 64))
    x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B)).cuda()
    eng.set_precision(_lib.PREC_F16X3)
    g = torch.empty(eng.param_vector_size, device="cuda")
    res = {0: [], 1: []}
    grads = {}
    for r in range(3):
        for off in (1, 0):
            if off: os.environ["GLOWK_WGRAD_16_OFF"] = "1"
            else: os.environ.pop("GLOWK_WGRAD_16_OFF", None)
            lib.glowk_reload_env()
            for _ in range(2): eng.param_grad(x, -1.0 / n, g)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            reps = 10 if n == 32 else 4
            for _ in range(reps): eng.param_grad(x, -1.0 / n, g)
            torch.cuda.synchronize()
            res[off].append((time.perf_counter() - t0) / reps * 1e3)
            grads[off] = g.double().cpu()
    os.environ.pop("GLOWK_WGRAD_16_OFF", None); lib.glowk_reload_env()
    a, b = sorted(res[1]), sorted(res[0])
    rel = ((grads[0] - grads[1]).norm() / grads[1].norm()).item()
    print("tiles %4d: param_grad with 256 x 128 tiles %.3f ms (min %.3f)   256 x 256 / 16 waves %.3f ms (min %.3f)   %+.1f %%   gradient rel l2 difference %.1e"
          % (n, a[1], a[0], b[1], b[0], 100 * (b[1] / a[1] - 1), rel), flush=True)
    eng.close()
