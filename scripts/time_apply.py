"""Time glowk_apply_gradients alone (optimizer step + device-side refresh of the kernel images) after one parameter-gradient sweep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
n = 32
eng, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=64)
x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B)).cuda()
for prec, name in ((_lib.PREC_F16X3, "f16x3"), (_lib.PREC_F32, "f32")):
    eng.set_precision(prec)
    _, g = eng.param_grad(x, -1.0 / n)
    g = g.clone()
    for _ in range(2):
        eng.apply_gradients(g, "adamax", 1e-6)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10):
        eng.apply_gradients(g, "adamax", 1e-6)
    torch.cuda.synchronize()
    print("%s: %.2f ms per apply_gradients" % (name, (time.time() - t0) / 10 * 1e3), flush=True)
