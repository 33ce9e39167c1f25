"""Time per tile of log_prob / log_prob_grad / param_grad against the number of tiles of one call, config B in f16x3 -- where do the
launch rules (glowk_launch.h) leave CUs idle?  Prints per-level k_net time too.   python scripts/grid_sweep.py [H W] [tile counts ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
args = [int(v) for v in sys.argv[1:]]
H, W = (args[0], args[1]) if len(args) >= 2 else (64, 64)
ns = args[2:] or [4, 8, 12, 16, 24, 30, 32, 40, 48, 64, 80, 96, 112, 128, 160, 192, 256, 384, 512]
cfg = GlowConfig(H=H, W=W, C=1, L=3, K=32, F=512)
eng, _ = calibrated_engine(cfg, device=0, init_tiles=32)
eng.set_precision(_lib.PREC_F16X3)
eng.set_range_policy("error")
base = torch.from_numpy(synthetic_mel_tiles(64, cfg, seed=11)).cuda()
def timed(f, reps):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for n in ns:
    x = base.repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
    reps = max(3, min(20, 2000 // n))
    lp = timed(lambda: eng.log_prob(x), reps)
    gr = timed(lambda: eng.log_prob_grad(x), reps) if n <= eng.grad_max_tiles else float("nan")
    pg = timed(lambda: eng.param_grad(x, -1.0 / n), max(2, reps // 2)) if n <= min(256, eng.grad_max_tiles) else float("nan")
    eng.profile_begin(); eng.log_prob(x); torch.cuda.synchronize(); pr = eng.profile_end()
    print("tiles %4d: log_prob %7.3f ms (%6.1f us/tile)  log_prob_grad %7.3f ms (%6.1f us/tile)  param_grad %7.3f ms (%6.1f us/tile)   k_net per level (forward): %s"
          % (n, lp, lp / n * 1e3, gr, gr / n * 1e3, pg, pg / n * 1e3, " ".join("%.2f" % ms for ms, _ in pr)), flush=True)
