"""One-off measurement for BASELINE.md: the torch-CPU fp32 restatement of the reference graph on the GPU
node's host cores, config B, 32 tiles (the reference's batch): faithful (the coupling network evaluated twice per step, as
TFP's forward + forward_log_det_jacobian do) and deduplicated, at all available threads and at 1 thread."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_params, synthetic_mel_tiles
from oracle import glowref_torch as RT

cfg = CONFIG_B
params = synthetic_params(cfg)
p = RT.to_torch(params, torch.float32)
aff = len(os.sched_getaffinity(0))
model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
print("cpu:", model[0] if model else "?", "| logical cpus in /proc/cpuinfo:", len(model), "| affinity:", aff, flush=True)
for threads, ntiles, reps in ((min(aff, 16), 32, 3), (1, 2, 1)):
    torch.set_num_threads(threads)
    x = torch.from_numpy(synthetic_mel_tiles(ntiles, cfg, seed=4321))
    with torch.no_grad():
        RT.log_prob(x[:1], p, cfg.as_dict())
        for evals, name in ((1, "deduplicated"), (2, "faithful (2 network evaluations per step)")):
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); RT.log_prob(x, p, cfg.as_dict(), evals_per_step=evals); ts.append(time.perf_counter() - t0)
            print("threads %2d  %-45s %d tiles  median %.2f s  -> %.2f passes/s" % (threads, name, ntiles, statistics.median(ts), ntiles / statistics.median(ts)), flush=True)
