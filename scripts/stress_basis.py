"""Determinism stress of the two-stream BASIS inner loop: the same Langevin steps (replayed noise) twice, bitwise equal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd import basis, _lib
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.flow_models.flow_glow import GlowFlow
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
e1, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=64)
e2, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=64, seed=4048)
m1, m2 = GlowFlow(e1), GlowFlow(e2)
n = 30
a = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=1)).cuda(); b = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=2)).cuda()
mixed = basis.mixing_db(a, b)
sig = basis.get_sigmas(1.0, 0.01, 10)
g = torch.Generator(device="cuda"); g.manual_seed(0)
noise = [[torch.randn(a.shape, device="cuda", generator=g) for _ in range(2)] for _ in range(5)]
x10 = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=3)).cuda(); x20 = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=4)).cuda()   # (uniform noise is far outside the synthetic priors' domain: inf/NaN)
for streams in ("auto", None):
  for prec in ("f32", "f16x3"):
    m1.set_precision(prec); m2.set_precision(prec)
    ref = None; prev = None; bad = badprev = 0; worst = 0.0
    for rep in range(int(os.environ.get("REPS", "60"))):
        r = basis.basis_inner_loop(mixed, x10.clone(), x20.clone(), m1, m2, 9, sig, T=5, noise_fn=lambda t, w, s: noise[t][w], streams=streams)
        torch.cuda.synchronize()
        if ref is None: ref = (r[0].clone(), r[1].clone())
        else:
            if not (torch.equal(r[0], ref[0]) and torch.equal(r[1], ref[1])): bad += 1; worst = max(worst, float((r[0] - ref[0]).abs().max()), float((r[1] - ref[1]).abs().max()))
            if not (torch.equal(r[0], prev[0]) and torch.equal(r[1], prev[1])): badprev += 1
        prev = (r[0].clone(), r[1].clone())
    print("BASIS inner loop, 5 steps, 30 tiles, %s, streams=%s: %d repeats differ from the first (max |diff| %.3e dB), %d from the previous" % (prec, streams, bad, worst, badprev), flush=True)
