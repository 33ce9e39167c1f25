"""Time glowk_param_grad alone (no optimizer step) in the exact and the split arithmetic: what the split training sweep buys."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
for n in (32, 256):
    eng, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=max(n, 64))
    x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B)).cuda()
    g = torch.empty(eng.param_vector_size, device="cuda")
    for prec, name in ((_lib.PREC_F32, "f32"), (_lib.PREC_F16X3, "f16x3")):
        eng.set_precision(prec)
        for _ in range(2):
            eng.param_grad(x, -1.0 / n, g)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5):
            eng.param_grad(x, -1.0 / n, g)
        torch.cuda.synchronize()
        print("batch %d %s: %.1f ms per param_grad" % (n, name, (time.time() - t0) / 5 * 1e3), flush=True)
    eng.close()
