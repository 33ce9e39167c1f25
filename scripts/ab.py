"""A/B timing of two builds of libglowk.so on the SAME box (boxes differ by up to ~10 % in sustained clock):
   python scripts/ab.py libA.so libB.so [rounds]   -- alternates subprocesses, prints median log_prob time at N=1024."""
import os, subprocess, sys
child = r'''
import sys, time, os, statistics
sys.path.insert(0, os.getcwd())
import torch
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
eng, _ = calibrated_engine(CONFIG_B, device=0)
eng.set_precision(int(os.environ.get("GLOWK_PREC", "1")))
eng.set_range_policy("ignore")   # (diagnostic builds compute wrong numbers on purpose: time them, do not re-run them in fp32)
N = int(os.environ.get("GLOWK_AB_N", "1024"))
x = torch.from_numpy(synthetic_mel_tiles(N, CONFIG_B)).cuda(); eng.reserve(N)
for _ in range(2): lp = eng.log_prob(x)
torch.cuda.synchronize(); ts = []
for _ in range(8):
    t0 = time.time(); lp = eng.log_prob(x); torch.cuda.synchronize(); ts.append(time.time() - t0)
print("N=%d  %.2f ms median  %.2f ms best  lp0=%.2f" % (N, 1e3 * statistics.median(ts), 1e3 * min(ts), lp[0].item()))
'''
libs = [a for a in sys.argv[1:] if a.endswith('.so')]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, GLOWK_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=300)
        print(os.path.basename(lib), (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
