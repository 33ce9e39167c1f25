"""In-kernel time stamps of one workgroup of the co-resident headline kernel k_net_h3c at 1024 tiles (workgroup 0, first lane; 100 MHz
counter): prologue, one block's X / Ya / Yb with their barrier waits, the passes, the fused tail.   python scripts/co_stamps.py [tiles=1024]"""
# needs a build with the stamps compiled in:  python -c "import __graft_entry__ as g; g.build(tag='stamps', extra_flags=['-DGLOWK_STAMPS'])"
#                                            GLOWK_LIB=$PWD/audiosourcesep_amd/libglowk_stamps.so python scripts/...
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = _lib.load()
buf = (ctypes.c_uint64 * 16)()
cfg = GlowConfig(H=64, W=64, C=1, L=2, K=1, F=512)      # one step at level 0, then one at level 1 (no stamps there)
eng, _ = calibrated_engine(cfg, device=0, init_tiles=64)
eng.set_precision(_lib.PREC_F16X3); eng.set_range_policy("error")
x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=3)).cuda()
lib.glowk_debug_stamps(buf, 0)
for rep in range(4):
    for _ in range(3): eng.log_prob(x)
    torch.cuda.synchronize()
    lib.glowk_debug_stamps(buf, 16)
    t = [int(v) for v in buf]
    us = lambda a_, b_: (t[b_] - t[a_]) / 100.0
    print("prologue %.2f us | block 2 of pass 0: X %.2f (+wait/barrier %.2f) Ya %.2f (+%.2f) Yb %.2f (+%.2f) = %.2f per block | pass 0 blocks %.2f, Z %.2f | pass 1 %.2f | tail %.2f | total %.2f"
          % (us(0, 1), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(6, 7), us(7, 8), us(2, 8) , us(1, 9), 0.0, us(9, 12) , us(12, 13), us(0, 13)))
