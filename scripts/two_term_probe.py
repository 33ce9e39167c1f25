"""Accuracy and throughput of the two-term forward mode (GLOWK_PREC_F16X2) beside f16x3 and exact fp32: config B, log_prob of
1024 tiles; fp64 oracle on 2 tiles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
from oracle import glowref as R
n = int(os.environ.get("N", "1024"))
eng, params = calibrated_engine(CONFIG_B, device=0, init_tiles=n)
x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=1234)).cuda(); eng.reserve(n)
ref2 = R.log_prob(synthetic_mel_tiles(n, CONFIG_B, seed=1234)[:2].astype(np.float64), R.cast_params(params, np.float64), CONFIG_B.as_dict())
res = {}
for name, prec in (("f32", _lib.PREC_F32), ("f16x3", _lib.PREC_F16X3), ("f16x2", _lib.PREC_F16X2)):
    eng.set_precision(prec)
    lp = eng.log_prob(x); z, ld = eng.forward(x); xr = eng.inverse(z)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3): eng.log_prob(x)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    res[name] = lp.double().cpu().numpy()
    e_or = np.abs(res[name][:2] - ref2).max() / np.abs(ref2).max()
    e32 = np.abs((res[name] - res["f32"]) / res["f32"]).max()
    print("%-6s %8.0f passes/s   max rel err vs fp64 oracle (2 tiles) %.2e   vs fp32 kernels (%d tiles) max %.2e mean %.2e   round trip %.2e dB"
          % (name, n / dt, e_or, n, e32, np.abs((res[name] - res["f32"]) / res["f32"]).mean(), float((xr - x).abs().max())), flush=True)
