"""Ad-hoc timing of the benchmark's pieces (diagnostic; not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_params, synthetic_mel_tiles, calibrated_engine

def T(msg, t0):
    print("%-40s %.3f s" % (msg, time.time() - t0), flush=True)

cfg = CONFIG_B
t0 = time.time(); eng, params = calibrated_engine(cfg, device=0); torch.cuda.synchronize(); T("calibrated_engine", t0)
if os.environ.get("GLOWK_PREC"):
    eng.set_precision(int(os.environ["GLOWK_PREC"]))
for n in (32, 128, 512, 1024):
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg)).cuda()
    eng.reserve(n)
    lp = eng.log_prob(x); torch.cuda.synchronize()
    t0 = time.time(); eng.profile_begin(); lp = eng.log_prob(x); torch.cuda.synchronize(); dt = time.time() - t0
    prof = eng.profile_end()
    print("N=%d  log_prob %.4f s  -> %.1f passes/s   k_net ms per level %s  lp[0]=%.2f" % (n, dt, n / dt, [(round(m, 3), c) for m, c in prof], lp[0].item()), flush=True)
if len(sys.argv) > 1:
    from oracle import glowref_torch as RT
    from oracle import glowref as R
    nthr = max(1, min(16, len(os.sched_getaffinity(0))))
    print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), "threads", nthr, flush=True)
    torch.set_num_threads(nthr)
    p = RT.to_torch(params, torch.float32)
    xc = torch.from_numpy(synthetic_mel_tiles(4, cfg))
    with torch.no_grad():
        t0 = time.time(); RT.log_prob(xc[:1], p, cfg.as_dict()); T("torch cpu 1 tile (first)", t0)
        t0 = time.time(); RT.log_prob(xc, p, cfg.as_dict()); T("torch cpu 4 tiles", t0)
    p32 = R.cast_params(params, np.float32)
    t0 = time.time(); R.log_prob(xc.numpy()[:1], p32, cfg.as_dict()); T("numpy fp32 1 tile", t0)
    t0 = time.time(); R.log_prob(xc.numpy(), p32, cfg.as_dict()); T("numpy fp32 4 tiles", t0)

if os.environ.get("GLOWK_TIME_GRAD"):
    for n in (30, 256, 1024):
        x = torch.from_numpy(synthetic_mel_tiles(n, cfg)).cuda()
        lp, dx = eng.log_prob_grad(x); torch.cuda.synchronize()
        t0 = time.time(); lp, dx = eng.log_prob_grad(x); torch.cuda.synchronize(); dt = time.time() - t0
        t0 = time.time(); lp2 = eng.log_prob(x); torch.cuda.synchronize(); dt2 = time.time() - t0
        print("N=%d  log_prob_grad %.4f s -> %.1f tiles/s   (log_prob alone %.4f s; ratio %.2f)  |dx| max %.3g finite %s"
              % (n, dt, n / dt, dt2, dt / dt2, dx.abs().max().item(), bool(torch.isfinite(dx).all())), flush=True)
