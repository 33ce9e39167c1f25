"""Experiment: does hipGraph replay shorten the small-batch launch chain?  log_prob / log_prob_grad of N tiles, eager vs a
torch.cuda.CUDAGraph capture of the same engine calls (the engine launches on torch's current stream, makes no host
synchronisation and allocates nothing once reserve(N) has run, so its launch sequence is capturable as is)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine

reps = int(os.environ.get("REPS", "30"))
eng, _ = calibrated_engine(CONFIG_B, device=0)
eng.set_precision(int(os.environ.get("GLOWK_PREC", "1")))
eng.set_range_policy("ignore")     # (the other policies wait for the stream after every call: not capturable)
for n in [int(v) for v in os.environ.get("NS", "1,8,30,128").split(",")]:
    x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B)).cuda()
    eng.reserve(n)
    eng.reserve(n, with_grad=True)
    for name, f in (("log_prob", lambda: eng.log_prob(x)), ("log_prob_grad", lambda: eng.log_prob_grad(x))):
        for _ in range(3):
            ref = f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / reps * 1e3
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            f()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            out = f()
        g.replay(); torch.cuda.synchronize()
        a = ref if torch.is_tensor(ref) else ref[-1]
        b = out if torch.is_tensor(out) else out[-1]
        same = bool(torch.equal(a, b))
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / reps * 1e3
        print("N=%4d %-14s eager %.3f ms   graph replay %.3f ms   (%.1f %%)  bitwise equal: %s"
              % (n, name, eager, graph, (graph / eager - 1) * 100, same), flush=True)
