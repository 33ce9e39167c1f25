"""Localise differences between the co-resident fused kernel and the one-workgroup-per-CU fused kernel (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
lib = _lib.load()
def setenv(name, on):
    if on: os.environ[name] = "1"
    else: os.environ.pop(name, None)
    lib.glowk_reload_env()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 160
cfg = GlowConfig(H=64, W=64, C=1, L=2, K=K, F=512)
eng, _ = calibrated_engine(cfg, device=0, init_tiles=32)
eng.set_precision(_lib.PREC_F16X3); eng.set_range_policy("ignore")
x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=8)).cuda()
setenv("GLOWK_NO_FUSE", len(sys.argv) > 3 and sys.argv[3] == "nofuse")
res = {}
for rep in range(3):
    for co in (False, True):
        setenv("GLOWK_CO_OFF", not co)
        z, ld = eng.forward(x)
        torch.cuda.synchronize()
        print("co", co, "range flag (diagnostic builds: LDS copy of P changed under the kernel):", eng.range_status()[0])
        res.setdefault(co, []).append((z.clone(), ld.clone()))
setenv("GLOWK_CO_OFF", False)
setenv("GLOWK_NO_FUSE", False)
za, lda = res[False][0]
for rep in range(3):
    zb, ldb = res[True][rep]
    dz = (zb - za).abs().flatten(1).max(dim=1).values
    dl = (ldb - lda).abs()
    bad = torch.nonzero((dz > 1e-4) | (dl > 1e-2)).flatten().tolist()
    print("rep", rep, "tiles off:", bad, "max dz %.3e max dld %.3e" % (float(dz.max()), float(dl.max())), "repeat-equal:", torch.equal(zb, res[True][0][0]), torch.equal(ldb, res[True][0][1]))
    for t in bad[:4]:
        d = (zb[t] - za[t]).abs()
        idx = torch.nonzero(d > 1e-4)
        # latent [H/4, W/4, 16]: level-0 factored half = channels 0..7 (row-major reshape of [32,32,2]), rest level 1
        print("  tile", t, "dld %.4f" % float(dl[t]), "n diff", len(idx), "first", idx[:4].tolist())
        # level-0 factored half: latent[..., :8] row-major = [32, 32, 2] (pixel-major, 2 channels)
        d0 = d[..., :8].reshape(32, 32, 2)
        rows = [(r, float(d0[r].max()), int((d0[r] > 1e-4).sum())) for r in range(32) if float(d0[r].max()) > 1e-4]
        print("    level-0 rows (row, max, count):", [(r, "%.1e" % m, c) for r, m, c in rows])
        cols = sorted(set(torch.nonzero(d0 > 1e-4)[:, 1].tolist()))
        print("    columns:", cols[:8], "...", cols[-4:], "n", len(cols))
