import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
from audiosourcesep_amd import _lib
eng, _ = calibrated_engine(CONFIG_B, device=0)
for prec in (_lib.PREC_F32, _lib.PREC_F16X3):
    eng.set_precision(prec)
    x = torch.from_numpy(synthetic_mel_tiles(512, CONFIG_B)).cuda()
    z, ld = eng.forward(x)
    torch.cuda.synchronize(); t0 = time.time(); xr = eng.inverse(z); torch.cuda.synchronize(); dt = time.time() - t0
    print("prec", prec, "inverse(forward(x)) max abs err dB %.3e   inverse 512 tiles %.1f ms -> %.0f tiles/s" % ((xr - x).abs().max().item(), dt * 1e3, 512 / dt))
    eps = torch.randn(512, *CONFIG_B.latent_shape(), device="cuda")
    torch.cuda.synchronize(); t0 = time.time(); xs = eng.sample_from_eps(eps); torch.cuda.synchronize(); dt = time.time() - t0
    print("   sample 512: %.1f ms, finite %s, range [%.1f, %.1f]" % (dt * 1e3, bool(torch.isfinite(xs).all()), xs.min().item(), xs.max().item()))
