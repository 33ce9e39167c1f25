import os, subprocess, sys
child = r'''
import sys, time, os, statistics
sys.path.insert(0, os.getcwd())
import torch
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_mel_tiles, calibrated_engine
eng, _ = calibrated_engine(CONFIG_B, device=0)
eng.set_precision(1)
N = int(os.environ.get("GLOWK_AB_N", "1024"))
x = torch.from_numpy(synthetic_mel_tiles(N, CONFIG_B)).cuda(); eng.reserve(N)
for _ in range(2): eng.log_prob_grad(x)
torch.cuda.synchronize(); ts = []
for _ in range(5):
    t0 = time.time(); eng.log_prob_grad(x); torch.cuda.synchronize(); ts.append(time.time() - t0)
print("grad N=%d  %.2f ms median" % (N, 1e3 * statistics.median(ts)))
'''
for r in range(2):
    for lib in sys.argv[1:3]:
        env = dict(os.environ, GLOWK_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=300)
        print(os.path.basename(lib), (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
