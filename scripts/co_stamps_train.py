"""In-kernel time stamps of workgroup 0 of the training sweep's level-0 BACKWARD launch in the co-resident form (k_net_h3c<4, 18, 16,
NET_BWD | 8>; the last stamped launch of a parameter-gradient sweep), 100 MHz counter.   python scripts/co_stamps_train.py [tiles=256]"""
# needs -DGLOWK_STAMPS (see co_stamps.py); with -DGLOWK_EXP_NOHST as well the same launch without its hidden stores
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
buf = (ctypes.c_uint64 * 16)()
cfg = GlowConfig(H=64, W=64, C=1, L=2, K=1, F=512)
eng, _ = calibrated_engine(cfg, device=0, init_tiles=64)
eng.set_precision(_lib.PREC_F16X3); eng.set_range_policy("error")
x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=3)).cuda()
lib.glowk_debug_stamps(buf, 0)
for rep in range(4):
    for _ in range(3): eng.param_grad(x, -1.0 / n)
    torch.cuda.synchronize()
    lib.glowk_debug_stamps(buf, 16)
    t = [int(v) for v in buf]
    us = lambda a_, b_: (t[b_] - t[a_]) / 100.0
    print("prologue %.2f us | block 2 of pass 0: X %.2f (+wait/barrier %.2f) Ya %.2f (+%.2f) Yb %.2f (+%.2f) = %.2f per block | pass 0 blocks %.2f | rest of pass 0 + pass 1 %.2f | total %.2f"
          % (us(0, 1), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(6, 7), us(7, 8), us(2, 8), us(1, 9), us(9, 12), us(0, 12)))
