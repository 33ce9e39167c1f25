import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiosourcesep_amd.config import CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_params
from audiosourcesep_amd.engine import GlowEngine
cfg = CONFIG_B
p = synthetic_params(cfg)
eng = GlowEngine(cfg, device=0)
t0 = time.time(); eng.load_params(p); t1 = time.time(); eng.finalize(); torch.cuda.synchronize(); t2 = time.time()
print("load_params %.2f s   finalize %.2f s" % (t1 - t0, t2 - t1))
t0 = time.time(); eng.load_params(p); eng.finalize(); torch.cuda.synchronize(); print("again: %.2f s" % (time.time() - t0))
