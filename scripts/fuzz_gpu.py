"""Randomised cross-check on the GPU (not part of the test suite; training checks per shape at the end of the loop): for random shapes and batch sizes -- i.e. random mixes of
launch forms (2/4 passes, split or not, 16x16x32 or 32x32x16 tiling, one lane or four lanes per pixel) -- the fp16x3 kernels
against the exact-fp32 kernels: log_prob, latent, inverse round trip, input gradient; plus batch independence (a tile's result
does not depend on what else is in the batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles

def tile_by_tile(cfg, eseed, x):
    eng, _ = calibrated_engine(cfg, device=0, init_tiles=16, seed=eseed)
    out = []
    for j in range(x.shape[0]):
        eng.set_precision(_lib.PREC_F32)
        a = eng.param_grad(x[j:j + 1], -1.0)[1].clone()
        eng.set_precision(_lib.PREC_F16X3)
        b = eng.param_grad(x[j:j + 1], -1.0)[1]
        out.append(float((a - b).norm() / a.norm()))
    eng.close()
    return out


rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "0")))
budget = float(os.environ.get("FUZZ_SECONDS", "240"))
t_end = time.time() + budget
case = 0
worst = {"lp": 0.0, "g": 0.0, "inv": 0.0, "batch": 0.0, "lp_two_term": 0.0}
while time.time() < t_end:
    L = int(rng.choice([2, 3, 3, 4]))
    unit = 2 ** L
    H, W = unit * int(rng.integers(1, 5)), unit * int(rng.integers(1, 5))
    F = int(rng.choice([128, 128, 256, 384, 512]))
    K = int(rng.integers(1, 4))
    if os.environ.get("FUZZ_ONLY"):   # e.g. FUZZ_ONLY=64,64,4,3,512: hammer one shape (random weights, batch sizes, call order)
        H, W, L, K, F = [int(v) for v in os.environ["FUZZ_ONLY"].split(",")]
    cfg = GlowConfig(H=H, W=W, C=1, L=L, K=K, F=F)
    eseed = int(rng.integers(1, 10 ** 6))
    eng, params = calibrated_engine(cfg, device=0, init_tiles=16, seed=eseed)
    for _ in range(3):
        n = int(rng.choice([1, 2, 3, 7, 30, 64, 129, 300, 700])) if F == 128 else int(rng.choice([1, 3, 30, 65, 200, 520]))
        xseed = int(rng.integers(1, 10 ** 6))
        x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=xseed)).cuda()
        eng.set_precision(_lib.PREC_F32)
        lp32, g32 = eng.log_prob_grad(x)
        z32, _ = eng.forward(x)
        eng.set_precision(_lib.PREC_F16X3)
        lp16, g16 = eng.log_prob_grad(x)
        lp16b = eng.log_prob(x)
        z16, _ = eng.forward(x)
        xr = eng.inverse(z16)
        eng.set_precision(_lib.PREC_F16X2)   # throughput mode: inside the 1e-4 bar (never fp32-class), gradient path = f16x3's
        lp2 = eng.log_prob(x)
        e_two = float(((lp2 - lp32).abs() / lp32.abs()).max())
        worst["lp_two_term"] = max(worst["lp_two_term"], e_two)
        if not (e_two < 1e-4 and bool(torch.isfinite(lp2).all())):
            print("FAIL two-term log_prob: H%d W%d L%d K%d F%d N%d  %.2e (engine seed %d, tiles seed %d)" % (H, W, L, K, F, n, e_two, eseed, xseed), flush=True)
            sys.exit(1)
        eng.set_precision(_lib.PREC_F16X3)
        e_lp = float(((lp16 - lp32).abs() / lp32.abs()).max())
        e_lpb = float(((lp16b - lp32).abs() / lp32.abs()).max())
        dg = (g16 - g32).abs() / g32.abs().max()
        e_g = float(dg.max())
        frac_g = float((dg > 1e-3).float().mean())      # isolated outliers = ReLU decisions that flipped between the arithmetics
        e_inv = float((xr - x).abs().max())
        j = int(rng.integers(0, n))
        e_b = float(((eng.log_prob(x[j:j + 1]) - lp16b[j:j + 1]).abs() / lp16b[j:j + 1].abs()).max())
        worst["lp"] = max(worst["lp"], e_lp, e_lpb); worst["g"] = max(worst["g"], e_g); worst["inv"] = max(worst["inv"], e_inv); worst["batch"] = max(worst["batch"], e_b)
        ok = e_lp < 5e-6 and e_lpb < 5e-6 and (e_g < 2e-3 or (frac_g < 5e-4 and e_g < 2e-2)) and e_inv < 5e-2 and e_b < 5e-6 and bool(torch.isfinite(g16).all())
        case += 1
        print("%s case %d: H%d W%d L%d K%d F%d N%d  lp %.1e/%.1e grad %.1e (>1e-3: %.1e of the entries) inv %.1e batch %.1e" % ("ok  " if ok else "FAIL", case, H, W, L, K, F, n, e_lp, e_lpb, e_g, frac_g, e_inv, e_b), flush=True)
        if not ok:
            print("   replay: H=%d W=%d L=%d K=%d F=%d engine seed %d, N=%d tiles seed %d" % (H, W, L, K, F, eseed, n, xseed), flush=True)
        if not ok:   # who is right?  both arithmetics against the fp64 autograd of the oracle.  A ReLU whose pre-activation is within
            # rounding of zero may be decided differently by ANY arithmetic; one such flip moves a few dozen gradient entries by
            # ~1e-3 of the maximum while log_prob does not notice -- tolerated; anything systematic is not.
            from oracle import glowref_torch as RT
            lp_ref, g_ref = RT.log_prob_and_grad(x.cpu().numpy().astype(np.float64), params, cfg.as_dict())
            sc = np.abs(g_ref).max()
            verdict = True
            for name, gg in (("fp32", g32), ("f16x3", g16)):
                d = np.abs(gg.cpu().numpy() - g_ref) / sc
                tiles = sorted(set(np.argwhere(d > 1e-3)[:, 0].tolist()))
                print("   %-5s vs fp64 autograd: max %.2e, entries > 1e-3: %d of %d, in tiles %s" % (name, d.max(), int((d > 1e-3).sum()), d.size, tiles[:6]), flush=True)
                if d.max() > 2e-2 or len(tiles) > 3:
                    verdict = False
            lp_ok = e_lp < 5e-6 and e_lpb < 5e-6 and e_inv < 5e-2 and e_b < 5e-6 and bool(torch.isfinite(g16).all())
            if not (verdict and lp_ok):
                sys.exit(1)
            print("   -> isolated ReLU flips, tolerated", flush=True)
    # ---- training (round 2): the parameter-gradient sweep in both arithmetics, one optimizer step in f16x3, and the kernel images
    #      refreshed on the device against the host packer's (a fresh engine loading the trained variables): bitwise
    from audiosourcesep_amd.flow_models.flow_glow import GlowFlow
    n = int(rng.choice([3, 17, 40]))
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=int(rng.integers(1, 10 ** 6)))).cuda()
    eng.set_precision(_lib.PREC_F32)
    _, ga = eng.param_grad(x, -1.0 / n)
    ga = ga.clone()
    eng.set_precision(_lib.PREC_F16X3)
    lpb, gb = eng.param_grad(x, -1.0 / n)
    e_pg = float((ga - gb).norm() / ga.norm())
    worst["param_grad"] = max(worst.get("param_grad", 0.0), e_pg)
    eng.apply_gradients(gb, "adamax", 1e-4)
    lp_dev = eng.log_prob(x)
    other, _ = calibrated_engine(cfg, device=0, init_tiles=16, seed=eseed)
    GlowFlow(other).load_state_dict(GlowFlow(eng).state_dict())
    other.set_precision(_lib.PREC_F16X3)
    same = bool(torch.equal(other.log_prob(x), lp_dev))
    fb = eng.range_status(sync=False)[1] + other.range_status(sync=False)[1]
    ok = e_pg < 2e-5 and same and bool(torch.isfinite(lp_dev).all())
    note = ""
    if same and not e_pg < 2e-5:   # a systematic difference shows in every tile; a ReLU decided differently by the two arithmetics (see
        # above) shows in the one tile it happened in: tile by tile, before the optimizer step (a fresh engine, same variables)
        per = tile_by_tile(cfg, eseed, x)
        bad = [j for j, e in enumerate(per) if e > 2e-5]
        note = "; per tile: %d of %d above 2e-5 (max %.1e, median %.1e)" % (len(bad), n, max(per), float(np.median(per)))
        ok = len(bad) <= 3 and max(per) < 5e-2 and float(np.median(per)) < 2e-6
    if same and not ok:
        # who is right?  Both sweeps against the fp64 autograd of the oracle on (at most 6 of) the same tiles, before the optimizer step:
        # a ReLU decided differently from fp64 by EITHER arithmetic moves that step's gradients by ~1e-4 and everything downstream of it by
        # ~1e-5 of the vector (round 3: seen at an L = 4 shape once its sweep ran split -- fp32 had flipped at level 1, f16x3 at level 2)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from test_gpu_training import oracle_param_grads
        chk, _ = calibrated_engine(cfg, device=0, init_tiles=16, seed=eseed)
        xs = x[:6]
        _, ref = oracle_param_grads(xs.cpu().numpy(), params, cfg, -1.0 / n)
        errs = {}
        for prec, tag in ((_lib.PREC_F32, "f32"), (_lib.PREC_F16X3, "f16x3")):
            chk.set_precision(prec)
            gg = chk.param_grad(xs, -1.0 / n)[1].cpu().numpy()
            num = den = 0.0
            per = []
            for key, r in ref.items():
                off, cnt = chk.param_slice(key)
                d2, r2 = float(np.sum((gg[off:off + cnt] - r.ravel()) ** 2)), float(np.sum(r ** 2))
                num += d2; den += r2
                if r2 > 0.0:
                    per.append((d2 / r2) ** 0.5)
            errs[tag] = ((num / den) ** 0.5, float(np.median(per)), float(np.max(per)))
        chk.close()
        note += "; vs fp64 autograd (whole vector, median tensor, worst tensor): fp32 %.1e %.1e %.1e, f16x3 %.1e %.1e %.1e" % (errs["f32"] + errs["f16x3"])
        # a flip: one step's tensors off by ~1e-4 ... 1e-3, everything downstream of it by ~1e-5, the typical tensor fp32-class; a broken
        # kernel would lift the MEDIAN tensor (every step of a level) -- that is what fails
        ok = errs["f16x3"][1] < 3e-5 and errs["f16x3"][2] < 5e-2 and errs["f16x3"][0] < 5e-3
    print("%s train: H%d W%d L%d K%d F%d N%d  |g16 - g32| / |g32| %.1e, device-refreshed images == host-packed: %s, fp32 fallbacks %d%s"
          % ("ok  " if ok else "FAIL", H, W, L, K, F, n, e_pg, same, fb, note), flush=True)
    if not ok:
        print("   replay: H=%d W=%d L=%d K=%d F=%d engine seed %d" % (H, W, L, K, F, eseed), flush=True)
        sys.exit(1)
    eng.close(); other.close()
print("fuzz: %d cases, worst %s" % (case, worst))
