"""One-off check of the batch bound: log_prob at glowk_max_tiles() + 5 tiles (config B: 65 541 tiles, two chunks, 13 GB of
workspace) and log_prob_grad at grad_max_tiles + 3; tiles at the far end of the batch must agree with the same tiles
evaluated in 1024-tile windows (an index overflow would show up there first)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
eng, _ = calibrated_engine(CONFIG_B, device=0)
base = torch.from_numpy(synthetic_mel_tiles(64, CONFIG_B, seed=11)).cuda()
for name, n, f in (("log_prob", eng.max_tiles + 5, lambda x: eng.log_prob(x)),
                   ("log_prob_grad", eng.grad_max_tiles + 3, lambda x: eng.log_prob_grad(x)[1])):
    reps = (n + 63) // 64
    x = base.repeat(reps, 1, 1, 1)[:n].contiguous()
    x += 0.01 * torch.randn(n, 1, 1, 1, device="cuda")      # every tile different, cheaply
    torch.cuda.synchronize(); t0 = time.time()
    out = f(x); torch.cuda.synchronize(); dt0 = time.time() - t0   # (first call: allocates the workspace)
    t0 = time.time(); out = f(x); torch.cuda.synchronize(); dt = time.time() - t0
    assert torch.isfinite(out).all()
    worst = 0.0
    w = 1024   # windows large enough to run the same launch forms as the big batch
    edge = eng.max_tiles if name == "log_prob" else eng.grad_max_tiles
    tail = 0.0
    for lo in (0, n // 2 - 3, edge - w, n - w):
        ref = f(x[lo:lo + w].contiguous())
        k = min(w, edge - lo)   # tiles past `edge` belong to the short last chunk, which runs the small-batch launch forms
        d = (out[lo:lo + k] - ref[:k]).abs().max().item() / ref.abs().max().item()
        worst = max(worst, d)
        if k < w:
            tail = max(tail, (out[lo + k:lo + w] - ref[k:]).abs().max().item() / ref.abs().max().item())
    print("%s: %d tiles in %.2f s (%.0f tiles/s), peak HBM %.1f GB, worst rel. diff vs 1024-tile windows %.2e "
          "(short last chunk, other launch forms: %.2e); first call %.2f s"
          % (name, n, dt, n / dt, torch.cuda.max_memory_allocated() / 2**30, worst, tail, dt0), flush=True)
    assert worst < 1e-5 and tail < 3e-3
    del x, out
print("ok")
