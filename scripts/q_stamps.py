"""In-kernel time stamps of one launch of the small-grid form k_net_h3q per level and direction (workgroup (0, 0), first lane; 100 MHz
counter): prologue / X-all / Y-all / Z, at the reference's 30 tiles.   python scripts/q_stamps.py [tiles=30]"""
# needs a build with the stamps compiled in:  python -c "import __graft_entry__ as g; g.build(tag='stamps', extra_flags=['-DGLOWK_STAMPS'])"
#                                            GLOWK_LIB=$PWD/audiosourcesep_amd/libglowk_stamps.so python scripts/...
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lib = _lib.load()
buf = (ctypes.c_uint64 * 8)()
# one level at a time: K = 1 and the level of interest LAST in launch order is what the buffer holds afterwards
for name, L in (("level 1 (c = 8)", 2), ("level 2 (c = 16)", 3)):
    cfg = GlowConfig(H=64, W=64, C=1, L=L, K=1, F=512)
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=32)
    eng.set_precision(_lib.PREC_F16X3); eng.set_range_policy("error")
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=3)).cuda()
    lib.glowk_debug_stamps(buf, 0)                      # arm
    for what, fn in (("plain forward", lambda: eng.log_prob(x)), ("gradient call (the last instrumented launch of it: the backward network of level 1)", lambda: eng.log_prob_grad(x))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        lib.glowk_debug_stamps(buf, 8)
        t = [int(v) for v in buf]
        us = lambda a_, b_: (t[b_] - t[a_]) / 100.0
        print("%s, %s: prologue %.2f us | X first half %.2f | Y first half %.2f | X second half %.2f | Y second half %.2f | Z %.2f | total %.2f" %
              (name, what, us(0, 1), us(2, 3), us(3, 6), us(6, 7), us(7, 4), us(4, 5), us(0, 5)))
