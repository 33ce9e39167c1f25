"""Diagnostic: throughput of log_prob / log_prob_grad against the batch size of one call (config B)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
eng, _ = calibrated_engine(CONFIG_B, device=0)
base = torch.from_numpy(synthetic_mel_tiles(256, CONFIG_B, seed=11)).cuda()
for n in [int(v) for v in os.environ.get("NS", "1024,2048,4096,8192,16384").split(",")]:
    x = base.repeat((n + 255) // 256, 1, 1, 1)[:n].contiguous()
    for name, f in (("log_prob", lambda: eng.log_prob(x)), ("log_prob_grad", lambda: eng.log_prob_grad(x))):
        if name == "log_prob_grad" and n > eng.grad_max_tiles:
            continue
        f(); torch.cuda.synchronize()
        eng.profile_begin(); t0 = time.time(); f(); torch.cuda.synchronize(); dt = time.time() - t0
        pr = eng.profile_end()
        print("N=%6d %-14s %8.1f ms  %7.0f tiles/s   k_net %s" % (n, name, dt * 1e3, n / dt, " ".join("%.1f" % ms for ms, _ in pr)), flush=True)
