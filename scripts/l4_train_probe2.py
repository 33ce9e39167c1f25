"""Who is right at the flagged case?  Both sweeps against the fp64 autograd of the oracle (tiles seed 2 of scripts/l4_train_probe.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
from test_gpu_training import oracle_param_grads
cfg = GlowConfig(H=64, W=48, C=1, L=4, K=3, F=512)
eng, params = calibrated_engine(cfg, device=0, init_tiles=16, seed=500021)
x = synthetic_mel_tiles(3, cfg, seed=2)
torch.set_num_threads(16)
_, ref = oracle_param_grads(x, params, cfg, -1.0 / 3)
xd = torch.from_numpy(x).cuda()
res = {}
for prec, tag in ((_lib.PREC_F32, "f32"), (_lib.PREC_F16X3, "f16x3")):
    eng.set_precision(prec)
    _, g = eng.param_grad(xd, -1.0 / 3)
    g = g.cpu().numpy()
    for lvl in range(cfg.L):
        for name in ("nn/conv1/kernel", "nn/conv2/kernel"):
            e = []
            for k in range(cfg.K):
                key = "b%d/s%d/%s" % (lvl, k, name)
                off, cnt = eng.param_slice(key)
                r = ref[key].ravel()
                e.append("%.1e" % (np.linalg.norm(g[off:off + cnt] - r) / np.linalg.norm(r)))
            print(tag, "level", lvl, name, "norm-rel vs fp64:", e, flush=True)
