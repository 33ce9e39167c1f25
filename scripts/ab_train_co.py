"""Training sweep: the co-resident form of level 0 (k_net_h3c<..., MODE | 8>) against the 32x32x16 family (GLOWK_CO_TRAIN_OFF=1) --
gradient vectors compared (both carry the split arithmetic's error, in different summation orders), log-probs, and time per sweep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine

sizes = [int(v) for v in sys.argv[1:]] or [32, 128, 256]
lib = _lib.load()
for n in sizes:
    eng, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=max(n, 64))
    x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B)).cuda()
    eng.set_precision(_lib.PREC_F16X3)
    res = {}
    for off in ("1", "0"):
        if off == "1": os.environ["GLOWK_CO_TRAIN_OFF"] = "1"      # (the switches are set / unset)
        else: os.environ.pop("GLOWK_CO_TRAIN_OFF", None)
        lib.glowk_reload_env()
        g = torch.zeros(eng.param_vector_size, device="cuda")
        before = eng.kernel_families()
        for _ in range(2):
            lp, _ = eng.param_grad(x, -1.0 / n, g)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5):
            lp, _ = eng.param_grad(x, -1.0 / n, g)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / 5 * 1e3
        after = eng.kernel_families()
        res[off] = (g.double().cpu(), lp.double().cpu(), ms, {k: after[k] - before[k] for k in after})
    g0, l0, t0_, f0 = res["1"]
    g1, l1, t1_, f1 = res["0"]
    rel = ((g1 - g0).norm() / g0.norm()).item()
    mx = ((g1 - g0).abs().max() / g0.abs().max()).item()
    print("tiles %4d: 32x32x16 family %.2f ms, co-resident %.2f ms (%.1f %%)  grad rel l2 %.2e  max/max %.2e  logp max diff %.2e  fallbacks %d"
          % (n, t0_, t1_, 100 * (t1_ / t0_ - 1), rel, mx, (l1 - l0).abs().max().item(), eng._fallbacks_seen), flush=True)
    print("   families off:", f0, "\n   families on: ", f1, flush=True)
    assert f1["co_resident"] > 0, "the co-resident training form did not run"
    eng.close()
