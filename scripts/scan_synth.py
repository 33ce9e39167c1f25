"""Diagnostic: how robust is the synthetic flow to tiles outside its ActNorm calibration set?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.engine import GlowEngine
from audiosourcesep_amd.synthetic import synthetic_params, synthetic_mel_tiles
cfg = CONFIG_B
xt = torch.from_numpy(synthetic_mel_tiles(1024, cfg, seed=99)).cuda()
for conv3_std in (0.01, 0.005, 0.0025):
    params = synthetic_params(cfg, conv3_std=conv3_std)
    eng = GlowEngine(cfg, device=0); eng.load_params(params)
    for init_tiles in (8, 64, 512):
        eng.actnorm_data_init(synthetic_mel_tiles(init_tiles, cfg, seed=77), runtime_order=True, raw_minibatch_quirk=False)
        lp, z = eng.log_prob(xt, return_latent=True)
        zmax = z.abs().amax(dim=(1, 2, 3))
        xr = eng.inverse(z)
        rt = (xr - xt).abs().amax(dim=(1, 2, 3))
        print("conv3_std %.4f init_tiles %4d: tiles |z|max>20: %4d  >100: %4d   lp median %.1f min %.3g | round-trip err median %.2e, >0.05: %d, max %.3g"
              % (conv3_std, init_tiles, int((zmax > 20).sum()), int((zmax > 100).sum()), lp.median().item(), lp.min().item(),
                 rt.median().item(), int((rt > 0.05).sum()), rt.max().item()), flush=True)
    eng.close()
