#!/usr/bin/env python3
"""Headline benchmark: Glow forward + log-det (= log_prob) passes/s on 64x64x1 mel tiles (K=32, L=3, F=512).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one log_prob pass of the hot path over one resident batch of synthetic tiles (``--batch`` per
GPU, default 1024).  Tiles are independent, so N ranks each own a shard (weak scaling); the only collective is
the all-reduce (RCCL) of the summed log-likelihood, once per step.  Rank 0 prints ONE JSON line.

Extra objects on the line
  roofline      dominant kernel = k_net_f32 at level 0 (32x32x4 tensors, 77 % of the FLOPs): algorithmic FLOP per
                launch (conv1+conv2+conv3 of the coupling network at 2 FLOP/MAC) divided by
                its average duration from HIP events recorded on the launch stream inside the timed region,
                against the fp32-input MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md:42).
  cpu_baseline  the CPU oracle (torch-CPU restatement, oracle/glowref_torch.py -- a port, not TensorFlow) timed
                on this host's cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from audiosourcesep_amd.config import GlowConfig, CONFIG_A, CONFIG_B, CONFIG_YAML  # noqa: E402
from audiosourcesep_amd.synthetic import synthetic_params, synthetic_mel_tiles  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md:42
PEAK_F16_MFMA_TFLOPS = 2500.0  # :43 (dense); the split path issues 3 fp16 MFMAs per fp32-equivalent product
SUSTAINED_F16_MFMA_TFLOPS = 1697.0   # scripts/mfma_shape.hip on this pool's MI355X: v_mfma_f32_16x16x32_f16, A from LDS, random data
SUSTAINED_F32_MFMA_TFLOPS = 150.8    # same program: v_mfma_f32_32x32x2_f32
PEAK_HBM_GBS = 8000.0          # :36


def baseline_metric():
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "Glow fwd+logdet passes/sec on 64x64x1 mel tiles (K=32,L=3), 1->8 MI355X"


def net_flop_per_pixel(c, F):
    """conv1 3x3 (c/2 -> F) + conv2 1x1 (F -> F) + conv3 3x3 (F -> c), 2 FLOP per MAC."""
    return 2 * (9 * (c // 2) * F + F * F + 9 * F * c)


def host_threads():
    """Threads the CPU baseline may use: the cgroup/affinity share, capped at 16 (one GPU's share of the host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def cpu_info():
    model, phys = "?", set()
    try:
        pid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "?":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                phys.add((pid, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    return {"model": model, "physical_cores_visible": len(phys) or None, "logical_cpus_visible": os.cpu_count(),
            "affinity": len(os.sched_getaffinity(0))}


def cpu_baseline(cfg, params, budget_s=24.0):
    """Oracle port on the host cores: torch-CPU fp32 restatement of the reference graph (oracle/glowref_torch.py; not
    TensorFlow).  Bounded sample of the same workload: median of ten passes each
    of (a) the deduplicated graph (one network evaluation per step) at the host share of threads -- the headline `value`,
    (b) the faithful graph (two evaluations per step, as TFP's forward + forward_log_det_jacobian do), (c) one thread."""
    from oracle import glowref_torch as RT
    threads = host_threads()
    # pin the process to `threads` CPUs for the duration, so that "cores" is what the number was measured on whatever thread
    # pools the CPU backends keep (the box shows all 256 logical CPUs of the host; one GPU's share is 16)
    allowed = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, set(allowed[:threads]))
    try:
        return _cpu_baseline_pinned(RT, cfg, params, budget_s, threads)
    finally:
        os.sched_setaffinity(0, set(allowed))


def _cpu_baseline_pinned(RT, cfg, params, budget_s, threads):
    p = RT.to_torch(params, torch.float32)
    x = torch.from_numpy(synthetic_mel_tiles(128, cfg, seed=4321))
    d = cfg.as_dict()

    def timed(tiles, evals, reps=10):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            RT.log_prob(x[:tiles], p, d, evals_per_step=evals)
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), sum(ts)

    out, spent = {}, 0.0
    with torch.no_grad():
        torch.set_num_threads(threads)
        RT.log_prob(x[:1], p, d)                      # warms up primitive creation (not timed)
        t0 = time.perf_counter()
        RT.log_prob(x[:2], p, d)
        t1 = (time.perf_counter() - t0) / 2.0
        share = budget_s / 3.0 / 10.0                 # three variants x ten passes (>= 10 timed passes, median)
        # (the same tile count for the deduplicated and the faithful graph: per-tile CPU time depends on the batch --
        #  hidden activations of 64 tiles no longer fit the caches -- and the two variants are meant to be compared)
        tiles = int(max(2, min(32, share / max(1.5 * t1, 1e-3))))
        for _ in range(3):                            # (per-tile CPU time depends on the batch -- caches -- so the tile count that makes ten
            t0 = time.perf_counter()                  #  passes fit the share is found by measuring: at most three calibration passes)
            RT.log_prob(x[:tiles], p, d)
            tp = time.perf_counter() - t0
            spent += tp
            new = int(max(2, min(32, tiles * share / max(tp, 1e-3) * 0.85)))
            if new == tiles or (tp <= 1.1 * share and new >= tiles and tiles == 32):
                break
            if tp <= 1.1 * share and new <= tiles:    # already inside the share: keep it
                break
            tiles = new
        med, tot = timed(tiles, 1)
        spent += tot
        out = {"value": tiles / med, "unit": "passes/s", "cores": threads, "kind": "port",
               "sample": "median of 10 passes over %d tiles of the same config (%.1f s of CPU work in all three variants), torch-CPU "
                         "fp32 restatement of the reference graph (oracle/glowref_torch.py; NOT TensorFlow), deduplicated graph "
                         "(one network evaluation per step)" % (tiles, 0.0)}
        tf = tiles
        med, tot = timed(tf, 2)
        spent += tot
        out["faithful"] = {"value": tf / med, "unit": "passes/s", "cores": threads, "tiles": tf,
                           "note": "coupling network evaluated twice per step, as TFP's forward + forward_log_det_jacobian do"}
        torch.set_num_threads(1)
        RT.log_prob(x[:1], p, d)
        t0 = time.perf_counter()
        RT.log_prob(x[:1], p, d)
        t1 = time.perf_counter() - t0
        t1n = int(max(1, min(4, share / max(t1, 1e-3))))
        med, tot = timed(t1n, 1, reps=3)
        spent += tot
        out["threads_1"] = {"value": t1n / med, "unit": "passes/s", "cores": 1, "tiles": t1n, "note": "deduplicated graph, one thread, median of 3 passes"}
        torch.set_num_threads(threads)
    out["sample"] = out["sample"].replace("(0.0 s", "(%.1f s" % spent)
    out["host"] = cpu_info()
    return out


def _timed_steps(step, steps, warmup, dist, rehearsal):
    """W untimed + K timed calls of `step`, bracketed by barrier + synchronize on both sides; max over ranks."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if rehearsal else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def logprob_workload(cfg, n, precision, steps, warmup, rank, local_rank, dist, rehearsal):
    """log_prob over a resident batch of n tiles (n = 32: the reference's batch, configs/melspec_glow.yml:15), under GLOWK_RANGE_ERROR."""
    from audiosourcesep_amd.synthetic import calibrated_engine
    eng, _ = calibrated_engine(cfg, device=local_rank, init_tiles=n)
    eng.set_precision(precision)
    eng.set_range_policy("error")
    eng.reserve(n)
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=1234 + rank)).cuda()
    lp = torch.empty(n, device="cuda", dtype=torch.float32)
    total = torch.zeros(1, device="cuda", dtype=torch.float64)
    el = _timed_steps(lambda: eng.log_prob_sum(x, out=lp, total=total), steps, warmup, dist, rehearsal)
    assert eng.range_status() == (False, 0), "the range guard fired"
    assert torch.isfinite(total).all()
    eng.close()
    return el


def grad_workload(cfg, n, precision, steps, warmup, rank, local_rank, dist, rehearsal):
    """log_prob + d/dx (compute_grad_logprob, run_basis_sep.py:73-79) over a resident batch, under GLOWK_RANGE_ERROR."""
    from audiosourcesep_amd.synthetic import calibrated_engine
    eng, _ = calibrated_engine(cfg, device=local_rank, init_tiles=n)
    eng.set_precision(precision)
    eng.set_range_policy("error")
    eng.reserve(n, with_grad=True)
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=1234 + rank)).cuda()
    el = _timed_steps(lambda: eng.log_prob_grad(x), steps, warmup, dist, rehearsal)
    assert eng.range_status() == (False, 0), "the range guard fired"
    eng.close()
    return el


def train_workload(cfg, n, precision, steps, warmup, rank, world, local_rank, dist, rehearsal):
    """One step of train_glow.py:37-54: loss + all parameter gradients (in `precision`: the split kernels with hidden stores, or the
    exact ones), ONE all-reduce of the flat gradient vector (RCCL), Adamax, device-side refresh of the kernel images.  The range
    policy is "fallback" -- a sweep whose gradient scale is not yet known is repeated on the exact kernels -- and the number of such
    repeats inside the run is part of the result."""
    from audiosourcesep_amd.synthetic import calibrated_engine
    from audiosourcesep_amd.distributed import distributed_train_step
    eng, _ = calibrated_engine(cfg, device=local_rank, init_tiles=n)
    eng.set_precision(precision)
    eng.set_range_policy("fallback")
    eng.reserve(n, with_grad=True)
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=1234 + rank)).cuda()
    state = {}

    def step():
        state["loss"] = distributed_train_step(eng.param_grad, lambda g: eng.apply_gradients(g, "adamax", 1e-4), x, n * world)
    for _ in range(2):      # (the first sweep sizes the dynamic gradient scale of the split sweep; not part of the measurement)
        step()
    before = eng.range_status(sync=False)[1]
    el = _timed_steps(step, steps, warmup, dist, rehearsal)
    fallbacks = eng.range_status(sync=False)[1] - before
    assert torch.isfinite(state["loss"]).all(), "training diverged"
    pv = eng.param_vector_size
    eng.close()
    return el, fallbacks, pv


def basis_workload(args, K, levels, train_steps, T, steps, warmup, rank, world, local_rank, dist, rehearsal, precision, sigma1=0.3, parallel="replicas"):
    """BASELINE config 5 as it is meant: two noise-conditioned Glow priors (L = 3, n_filters = 512, K steps per level) are trained
    here with the repo's own training step on the reference's 30 real tiles per stem (tests/golden/basis_real_tiles.npz; the ladder
    of train_noisy_glow.py:309-358), kept resident per sigma, and the BASIS chain (run_basis_sep.py:217-260) runs on the 30 mixture
    tiles from the reference's uniform start in `precision` under GLOWK_RANGE_ERROR.  Timed: `steps` CONSECUTIVE Langevin steps of
    that chain at the last (smallest) sigma, after the chain has run T steps at every level -- no restart, the state carries over.
    parallel = "replicas": every rank separates the same 30 tiles with its own noise stream (weak scaling of independent chains).
    parallel = "prior": the ONE 30-tile problem over the job (strong scaling): rank r trains and holds only prior r % 2 and the tiles of
    shard r // 2; per Langevin step it evaluates its own prior's gradient, the pair all-gathers the two gradients (RCCL), both run the
    identical update (basis.prior_parallel_layout)."""
    from audiosourcesep_amd import basis
    from audiosourcesep_amd.flow_models.flow_builder import build_glow
    from audiosourcesep_amd.noise_conditioned import fine_tune_ladder, psnr_db, db_schedule
    f = np.load(os.path.join(ROOT, "tests", "golden", "basis_real_tiles.npz"))
    crop = getattr(args, "basis_crop", 0) or 96
    gt1, gt2, mixed = (torch.from_numpy(np.ascontiguousarray(f[k].astype(np.float32)[:, :crop]))[..., None].cuda() for k in ("gt1", "gt2", "mixed"))
    pp = parallel == "prior"
    lay = pair_group = None
    if pp:
        if dist is None or world % 2:
            raise SystemExit("--basis-parallel prior needs an even number of ranks (--gpus 2, 4, ...)")
        lay = basis.prior_parallel_layout(mixed.shape[0], world, rank)
        pair_group = basis.make_pair_group(world, rank)
        a, b = lay["bounds"]
        gt1, gt2, mixed = gt1[a:b].contiguous(), gt2[a:b].contiguous(), mixed[a:b].contiguous()
    n = mixed.shape[0]
    t0 = time.perf_counter()
    priors, ladders, fb_train, loss_curves = [None, None], [None, None], [], []
    for i, gt in enumerate((gt1, gt2)):
        if pp and i != lay["prior"]:
            continue                    # the pair partner owns this prior
        flow = build_glow(gt, [crop, 64, 1], L=3, K=K, n_filters=512, learntop=True, seed=100 + i, precision=precision, actnorm_init="runtime",
                          device=local_rank, data_type="melspec", minval=-100.0, maxval=20.0, use_logit=False)
        flow.engine.set_range_policy("fallback")
        sig_db, delta_db = db_schedule(flow.cfg, sigma1=sigma1, sigmaL=0.01, num_classes=levels)
        models, losses = fine_tune_ladder(flow, gt, sig_db, [3 * train_steps] + [train_steps] * (levels - 1), lr=1e-3, seed=7 + i)
        assert all(np.isfinite(losses[float(s)]).all() for s in sig_db), "prior training diverged"
        loss_curves.append({"%.3g" % float(s): [losses[float(s)][0], losses[float(s)][-1]] for s in sig_db})
        fb_train.append(flow.engine.range_status()[1])
        priors[i] = flow
        ladders[i] = models
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    cfg = [p for p in priors if p is not None][0].cfg
    engines = [m[float(s)].engine for m in ladders if m is not None for s in sig_db]
    seed_rank = 0 if pp else rank           # (prior-parallel: the pair shares ONE chain -- same start, same noise stream, tile offsets)
    t_off = lay["bounds"][0] if pp else 0
    e_off = t_off * int(np.prod(mixed.shape[1:]))
    full_shape = (30,) + tuple(mixed.shape[1:])
    for e in engines:
        e.set_range_policy("error")
        e.range_probe_begin()
    x1 = -100.0 + 120.0 * basis.device_randn(tuple(mixed.shape), mixed.device, seed=11 + seed_rank, which=0, uniform=True, offset=e_off)
    x2 = -100.0 + 120.0 * basis.device_randn(tuple(mixed.shape), mixed.device, seed=11 + seed_rank, which=1, uniform=True, offset=e_off)
    start = (psnr_db(x1, gt1), psnr_db(x2, gt2))
    t0 = time.perf_counter()
    y1, y2, arr = basis.basis_outer_loop(mixed, x1, x2, priors[0], priors[1], sig_db, restore_1=ladders[0], restore_2=ladders[1], T=T,
                                         delta=delta_db, seed=3 + seed_rank, tile_offset=t_off, prior_group=pair_group,
                                         prior_index=lay["prior"] if pp else None)
    torch.cuda.synchronize()
    t_chain = time.perf_counter() - t0
    psnr_levels = [(round(psnr_db(a, gt1), 2), round(psnr_db(b, gt2), 2)) for a, b in zip(arr["x1"], arr["x2"])]
    assert torch.isfinite(y1).all() and torch.isfinite(y2).all(), "the chain left the finite range"
    # timed region: consecutive steps at the last level, the state carried from call to call
    last = len(sig_db) - 1
    m1 = ladders[0][float(sig_db[last])] if ladders[0] is not None else None
    m2 = ladders[1][float(sig_db[last])] if ladders[1] is not None else None
    state = {"x1": y1, "x2": y2, "t": len(sig_db) * T}

    def step():
        state["x1"], state["x2"] = basis.basis_inner_loop(mixed, state["x1"], state["x2"], m1, m2, last, sig_db, delta=delta_db, T=1,
                                                          seed=3 + seed_rank, step0=state["t"], offset=e_off, prior_group=pair_group,
                                                          prior_index=lay["prior"] if pp else None)
        state["t"] += 1
    el = _timed_steps(step, steps, warmup, dist, rehearsal)
    assert torch.isfinite(state["x1"]).all() and torch.isfinite(state["x2"]).all(), "the chain left the finite range"
    margins = [e.range_probe_end() for e in engines]
    trips = sum(int(e.range_status()[0]) + e.range_status(sync=False)[1] for e in engines)
    end = (psnr_db(state["x1"], gt1), psnr_db(state["x2"], gt2))
    flop_step = (1 if pp else 2) * 2 * cfg.flop_per_tile()     # priors on this GPU x (forward + data gradient = 2x forward FLOPs)
    out = {
        "value": (30 if pp else n * world) * steps / el, "unit": "tile-steps/s", "ms_per_step": el / steps * 1e3, "dtype": precision, "steps": steps,
        "parallelism": ("prior-parallel: rank r holds prior r %% 2 of tile shard r // 2 (%d shard(s)); one all-gather of the two gradients per step; "
                        "STRONG scaling of the one 30-tile problem" % (world // 2)) if pp else "replicas: every rank runs its own 30-tile chain (weak scaling)",
        "config": {"workload": "BASIS Langevin steps, 2 trained noise-conditioned Glow priors, %dx64x1 real mel tiles (30 per GPU), L=3 K=%d "
                               "n_filters=512, %d sigma levels x T=%d then %d timed consecutive steps at sigma_L" % (crop, K, levels, T, steps)},
        "roofline": {"bound": "mfma", "achieved": n * steps / el * flop_step / 1e12, "peak": PEAK_F16_MFMA_TFLOPS / 3.0 if precision != "f32" else PEAK_F32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "note": "per GPU: %d prior(s) x (forward + data gradient = 2x forward FLOP) per tile-step" % (1 if pp else 2)},
        "range_guard": {"policy": "GLOWK_RANGE_ERROR on every chain step", "trips_or_fallbacks_in_chain": trips,
                        "largest_forward_input_over_limit": max(m[0] for m in margins), "backward_static_ratio": max(m[1] for m in margins),
                        "fallback_sweeps_while_training": fb_train},
        "priors": {"trained_here_s": t_train, "sigmas_db": [float(s) for s in sig_db], "delta_db2": delta_db,
                   "train_steps_per_level": [3 * train_steps] + [train_steps] * (levels - 1), "loss_first_last_per_sigma_db": loss_curves},
        "chain": {"levels_x_T_s": t_chain, "langevin_steps": len(sig_db) * T, "tile_steps_per_s_whole_ladder": n * len(sig_db) * T / t_chain,
                  "psnr_db_start": start, "psnr_db_end": end, "psnr_db_after_each_level": psnr_levels,
                  "reference_wall_clock_s": {"value": 1411.5, "what": "the reference's own log of this separation (30 tiles, 10 sigma x T=100, NCSN priors, "
                                                                       "its GPU): basis_sep_results/.../out.log:124 -- other priors, other hardware: context only"},
                  "psnr_db_reference_shipped_result": (psnr_db(f["x1"].astype(np.float32)[:, :crop], f["gt1"].astype(np.float32)[:, :crop]),
                                                       psnr_db(f["x2"].astype(np.float32)[:, :crop], f["gt2"].astype(np.float32)[:, :crop]))},
    }
    out["roofline"]["frac"] = out["roofline"]["achieved"] / out["roofline"]["peak"]
    for e in engines:
        e.close()
    return out


def secondary_workload(args, cfg, rank, world, local_rank, dist, rehearsal, extra):
    """--workload log_prob_grad | train | basis as the ONE line of the run."""
    n = args.batch
    prec = {"f32": 0, "f16x3": 1, "f16x2": 2}[args.precision]
    base = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic"}
    if args.workload == "log_prob_grad":
        el = grad_workload(cfg, n, prec, args.steps, args.warmup, rank, local_rank, dist, rehearsal)
        v = n * world * args.steps / el
        out = dict(base, metric="Glow log_prob + input-gradient tiles/sec", value=v, unit="tiles/s", ms_per_step=el / args.steps * 1e3,
                   config={"workload": "log_prob_grad, %dx%dx%d tiles, L=%d K=%d n_filters=%d, %d tiles/GPU" % (cfg.H, cfg.W, cfg.C, cfg.L, cfg.K, cfg.F, n)},
                   roofline=grad_roofline(cfg, v / world, args.precision, 2))
    elif args.workload == "train":
        el, fb, pv = train_workload(cfg, n, prec, args.steps, args.warmup, rank, world, local_rank, dist, rehearsal)
        v = n * world * args.steps / el
        out = dict(base, metric="Glow training step tiles/sec (loss + all gradients + Adamax)", value=v, unit="tiles/s",
                   ms_per_step=el / args.steps * 1e3, param_vector_floats=pv, fallback_sweeps_in_timed_region=fb,
                   config={"workload": "train, %dx%dx%d tiles, L=%d K=%d n_filters=%d, %d tiles/GPU" % (cfg.H, cfg.W, cfg.C, cfg.L, cfg.K, cfg.F, n)},
                   roofline=grad_roofline(cfg, v / world, args.precision, 3))
    else:
        r = basis_workload(args, args.basis_K, args.basis_levels, args.basis_train_steps, args.basis_T, args.steps, args.warmup, rank, world,
                           local_rank, dist, rehearsal, args.precision, sigma1=args.basis_sigma1, parallel=args.basis_parallel)
        if args.basis_parallel == "prior":
            base = dict(base, scaling="strong")
        out = dict(base, metric="BASIS Langevin tile-steps/sec (2 trained noise-conditioned Glow priors)", data="real mel tiles shipped with the reference (30 per stem); priors trained in this run", **r)
    out.update(extra)
    if rank == 0:
        print(json.dumps(out), flush=True)


def grad_roofline(cfg, tiles_per_s_per_gpu, precision, fwd_multiples):
    """Whole-path roofline of the gradient workloads: algorithmic FLOP = `fwd_multiples` x the forward pass (forward +
    data gradient = 2x; the training step adds the weight gradients: 3x) against the MFMA peak of the arithmetic."""
    peak = PEAK_F32_MFMA_TFLOPS if precision == "f32" else PEAK_F16_MFMA_TFLOPS / 3.0
    ach = tiles_per_s_per_gpu * fwd_multiples * cfg.flop_per_tile() / 1e12
    return {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "note": "%dx the forward pass's algorithmic FLOP per tile (%.2f GFLOP), whole step incl. light kernels and optimizer" % (fwd_multiples, cfg.flop_per_tile() / 1e9)}


def summary_of(out):
    """<= 1 500 characters: the values a reader of the line's last 2 000 characters needs (headline, the other arithmetics and the
    secondary workloads with their roofline fractions)."""
    def r4(v):
        return None if v is None else float("%.4g" % v)

    def sub(name):
        o = out.get(name)
        if not isinstance(o, dict):
            return None
        if "error" in o:
            return {"error": o["error"][:60]}
        d = {"v": r4(o.get("value")), "ms": r4(o.get("ms_per_step"))}
        if isinstance(o.get("roofline"), dict):
            d["frac"] = r4(o["roofline"].get("frac"))
        return d
    s = {"value": r4(out["value"]), "unit": out["unit"], "dtype": out["dtype"], "n_gpus": out["n_gpus"], "ms_per_step": r4(out["ms_per_step"]),
         "roofline_frac": r4(out["roofline"].get("frac")), "avg_launch_ms": r4(out["roofline"].get("avg_launch_ms")),
         "mfma_busy_pmc": out["roofline"].get("mfma_busy_pmc"), "traffic": out["roofline"].get("traffic")}
    for k in ("exact_fp32", "split_fp16", "two_term_split_fp16", "log_prob_32", "log_prob_grad_1024", "log_prob_grad_30", "train_32", "basis_30"):
        v = sub(k)
        if v is not None:
            s[k] = v
    if isinstance(out.get("config_A_32x32_K16_L2"), dict):
        s["config_A"] = r4(out["config_A_32x32_K16_L2"].get("value"))
    if isinstance(out.get("basis_30"), dict) and "chain" in out["basis_30"]:
        s["basis_30"]["psnr_end"] = [r4(v) for v in out["basis_30"]["chain"]["psnr_db_end"]]
    if isinstance(out.get("accuracy"), dict):
        s["acc"] = {("vs_fp64" if "fp64" in k else "vs_other"): r4(v) for k, v in out["accuracy"].items() if k != "north_star_bar"}
    if isinstance(out.get("cpu_baseline"), dict):
        s["cpu"] = {"v": r4(out["cpu_baseline"]["value"]), "cores": out["cpu_baseline"]["cores"]}
    s["git_head"] = out.get("git_head")
    return s


def self_launch(args):
    """``python bench.py --gpus N`` outside torch.distributed.run: start the N ranks ourselves -- BEFORE anything in this
    process touches the GPU (nothing has: importing torch does not) -- relay their output and exit code.  One process per
    GPU, rendezvous on 127.0.0.1 (train_glow.py:48-60 is the reference's MirroredStrategy counterpart)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in r.stdout.splitlines():     # rank 0's ONE JSON line goes to stdout; anything else the ranks' libraries printed, to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
    return r.returncode


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def file_commit(rel):
    """Commit that last touched a committed profile file (None outside a git checkout, e.g. on the GPU box's snapshot)."""
    try:
        return subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", rel], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="tiles per GPU per step")
    ap.add_argument("--config", default="B", choices=["A", "B", "YAML"])
    ap.add_argument("--workload", default="log_prob", choices=["log_prob", "log_prob_grad", "basis", "train"],
                    help="log_prob = BASELINE.json's headline metric; the others are secondary lines (SURVEY section 8f-1, 8f-3)")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3", "f16x2"],
                    help="f16x3: error-compensated fp16 split on the fp16 MFMA (fp32-class accuracy, demonstrated in the line); "
                         "f32: exact fp32-input MFMA")
    ap.add_argument("--basis-K", type=int, default=32, help="--workload basis: flow steps per level of the two priors")
    ap.add_argument("--basis-levels", type=int, default=4, help="--workload basis: sigma levels of the ladder (reference: 10)")
    ap.add_argument("--basis-sigma1", type=float, default=0.3, help="--workload basis: largest sigma of the ladder in the reference's normalised units "
                                                                     "(reference: 1.0 = 120 dB; run_basis_sep.py:492)")
    ap.add_argument("--basis-parallel", default="replicas", choices=["replicas", "prior"],
                    help="--workload basis with --gpus N > 1: independent 30-tile chains per rank (weak scaling), or the ONE 30-tile problem "
                         "prior-parallel over rank pairs composed with tile shards (strong scaling; N even)")
    ap.add_argument("--basis-train-steps", type=int, default=100, help="--workload basis: training steps per sigma level (3x at the first)")
    ap.add_argument("--basis-T", type=int, default=100, help="--workload basis: Langevin steps per sigma level before the timed region")
    ap.add_argument("--basis-crop", type=int, default=0, help="--workload basis: use only the first N mel bins of the 96x64 tiles (64: the 64x64 geometry of config B)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the log_prob_grad / train / basis sub-objects of the N=1 default run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-shapes", action="store_true", help="skip the config-A line inside the N=1 default run")
    ap.add_argument("--cpu-budget", type=float, default=24.0, help="seconds of CPU work for the cpu_baseline sample")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(self_launch(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%s: launch with --nproc-per-node %d" % (args.gpus, os.environ["WORLD_SIZE"], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GLOWK_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend -- exercises the multi-process control flow (sharding,
    # barriers, max-over-ranks timing, rank-0 JSON) on a one-GPU box; the numbers it prints mean nothing
    rehearsal = os.environ.get("GLOWK_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # GLOWK_BENCH_FORCE_DIST=1: take the RCCL path (init, barriers, all-reduces) even with one rank -- a one-GPU check that
    # the process-group calls this file makes work on the box's RCCL build
    if world > 1 or os.environ.get("GLOWK_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # what the collective actually saw: an all-reduce of ones over the job's process group (= the number of ranks RCCL connected),
    # and the device every rank computes on -- so that a line with n_gpus = N validates itself
    dist_info = {}
    if dist is not None:
        ones = torch.ones(1, dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        props = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "local_rank": local_rank, "cuda_device": torch.cuda.current_device(), "name": props.name,
                "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        dist_info = {"rccl_ranks_seen": int(ones.item()), "backend": dist.get_backend(), "rank_devices": gathered}

    cfg = {"A": CONFIG_A, "B": CONFIG_B, "YAML": CONFIG_YAML}[args.config]
    from audiosourcesep_amd import _lib
    from audiosourcesep_amd.synthetic import calibrated_engine
    from audiosourcesep_amd.distributed import sharded_log_prob
    PREC = {"f32": _lib.PREC_F32, "f16x3": _lib.PREC_F16X3, "f16x2": _lib.PREC_F16X2}
    n = args.batch

    if args.workload != "log_prob":
        secondary_workload(args, cfg, rank, world, local_rank, dist, rehearsal, dict(dist_info, git_head=git_head()))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    def measure(cfg, precision, steps, warmup, with_others):
        """One engine, one resident batch; K timed steps of the hot path in `precision`, then (with_others) the other two
        arithmetics on the same batch.  Every call runs under the C ABI's default range policy (GLOWK_RANGE_ERROR): a hidden
        activation outside the fp16 range of the split kernels would abort the benchmark instead of timing NaNs."""
        # synthetic weights + ActNorm data-dependent init on a minibatch of the benchmark's own batch size, so that
        # every k_net launch of the process has the same grid (rocprof's per-kernel average == the timed one)
        eng, params = calibrated_engine(cfg, device=local_rank, init_tiles=n)
        eng.set_range_policy("error")
        eng.set_precision(PREC[precision])
        eng.reserve(n)
        x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=1234 + rank)).cuda()   # resident in HBM before timing
        lp = torch.empty(n, device="cuda", dtype=torch.float32)
        total = torch.zeros(1, device="cuda", dtype=torch.float64)

        def step():
            # local shard on this GPU -- log_prob [n] and its fp64 sum both leave the engine (glowk_log_prob_sum) --, then ONE
            # all-reduce of that one element (RCCL over xGMI).  Nothing but the engine's kernels and the collective runs here.
            sharded_log_prob(eng, x, total=total, out=lp)

        def timed(k_steps):
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            eng.profile_begin()
            t0 = time.perf_counter()
            for _ in range(k_steps):
                step()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            pr = eng.profile_end()
            if dist is not None:
                t = torch.tensor([el], device="cpu" if rehearsal else "cuda", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            return el, pr

        for _ in range(warmup):
            step()
        res = {"eng": eng, "params": params, "x": x}
        fam0 = eng.kernel_families()
        res["main"] = timed(steps)
        res["co"] = eng.kernel_families()["co_resident"] > fam0["co_resident"]     # the level-0 launches took the co-resident form
        assert torch.isfinite(total).all(), "non-finite log-likelihood"
        assert eng.range_status() == (False, 0), "the range guard fired"
        res["lp_main"] = lp.clone()
        if with_others:
            # the other arithmetic on the same batch: exact fp32 beside the split path (or vice versa), same run
            other = "f32" if precision != "f32" else "f16x3"
            steps_o = max(2, steps // 2)
            eng.set_precision(PREC[other])
            step()
            res["other"] = (other, steps_o) + timed(steps_o)
            lp_other = lp.clone()
            res["rel_diff"] = float(((res["lp_main"] - lp_other).abs() / lp_other.abs()).max().item())
            # the throughput mode (two split terms per product: inside the 1e-4 bar, not fp32-class), same batch, same run
            if precision == "f16x3":
                eng.set_precision(PREC["f16x2"])
                step()
                el2, pr2 = timed(steps_o)
                res["two"] = (el2, pr2, float(((lp - lp_other).abs() / lp_other.abs()).max().item()))
            eng.set_precision(PREC[precision])
        return res

    def roofline_obj(cfg, precision, pr, co=False):
        h0, w0, c0 = cfg.level_shapes()[0]
        h3name = "k_net_h3c (co-resident: 4 waves / 128 pixels, two workgroups per CU)" if co else "k_net_h3s"
        flop_launch = net_flop_per_pixel(c0, cfg.F) * n * h0 * w0
        ms0, launches0 = pr[0]
        avg_ms = ms0 / max(launches0, 1)
        achieved = flop_launch / (avg_ms * 1e-3) / 1e12 if launches0 else None
        if precision == "f32":
            kernel, peak = "k_net_f32<CI=%d,NF=%d> (level 0)" % (c0 // 2, cfg.F // 32), PEAK_F32_MFMA_TFLOPS
            note = "fp32-input MFMA peak"
        elif precision == "f16x2":
            kernel, peak = "%s<CI=%d,NF=%d,two-term%s> (level 0)" % (h3name, c0 // 2, cfg.F // 32, ", coupling fused in" if c0 == 4 else ""), PEAK_F16_MFMA_TFLOPS / 2.0
            note = "fp16 dense MFMA peak / 2 (two fp16 MFMAs per product)"
        else:
            kernel, peak = "%s<CI=%d,NF=%d%s> (level 0)" % (h3name, c0 // 2, cfg.F // 32, ", coupling fused in" if c0 == 4 else ""), PEAK_F16_MFMA_TFLOPS / 3.0
            note = "fp16 dense MFMA peak / 3 (three fp16 MFMAs per fp32-equivalent product)"
        # HBM bytes per launch and MFMA-pipe busy fraction come from separate rocprofv3 --pmc passes of this same command
        # (scripts/final_run.sh, summarised by scripts/pmc_summary.py), committed under profiles/: NOT measured by this process -- the
        # `counters_source` object says which file they were read from, the commit that file belongs to (where there is no git
        # history -- the GPU box -- the commit of the build the counters were collected on, which the file records) and that build
        traffic = mfma_busy = None
        src = {}
        if cfg is CONFIG_B and n == 1024:
            tpath, upath = "profiles/roofline_traffic.json", "profiles/mfma_utilisation.json"
            try:
                tj = json.load(open(os.path.join(ROOT, tpath)))
                key = {"f32": "k_net_f32", "f16x3": "k_net_h3s", "f16x2": "k_net_h3s_two_term"}[precision]
                traffic = tj["hbm_bytes_per_level0_launch"][key]
                src["traffic"] = {"file": tpath, "commit": file_commit(tpath) or tj.get("build"), "build": tj.get("build"), "measured_in_this_run": False}
            except Exception:
                traffic = None
            try:
                uj = json.load(open(os.path.join(ROOT, upath)))
                mfma_busy = uj["level0"][{"f32": "k_net_f32", "f16x3": "k_net_h3s", "f16x2": "k_net_h3s_two_term"}[precision]]["mfma_utilisation"]
                src["mfma_busy_pmc"] = {"file": upath, "commit": file_commit(upath) or uj.get("build"), "build": uj.get("build"), "measured_in_this_run": False}
            except Exception:
                mfma_busy = None
        # what a bare MFMA loop with this kernel's operand pattern sustains on this chip under its power management
        # (scripts/mfma_shape.hip, random data): 1 697 TFLOP/s for 16x16x32 f16 (/3), 150.8 TFLOP/s for 32x32x2 f32
        sustained = SUSTAINED_F32_MFMA_TFLOPS if precision == "f32" else SUSTAINED_F16_MFMA_TFLOPS / (2.0 if precision == "f16x2" else 3.0)
        return {"kernel": kernel, "bound": "mfma", "achieved": achieved, "peak": peak, "peak_note": note, "unit": "TFLOP/s",
                "frac": (achieved / peak) if achieved else None, "traffic": traffic, "mfma_busy_pmc": mfma_busy, "counters_source": src,
                "avg_launch_ms": avg_ms, "launches": launches0, "flop_per_launch": flop_launch, "sustained_mfma_rate_measured": sustained,
                "frac_of_sustained": (achieved / sustained) if achieved else None}

    dtype_note = {"f32": "fp32 operands on v_mfma_f32_32x32x2_f32, fp32 accumulate",
                  "f16x3": "fp32 operands split into fp16 hi + lo, 3 fp16 MFMAs per product, fp32 accumulate",
                  "f16x2": "weights split into fp16 hi + lo, activations rounded to fp16, 2 fp16 MFMAs per product, fp32 accumulate"}

    def line(cfg, precision, res, steps, warmup, full):
        elapsed, prof = res["main"]
        value = n * world * steps / elapsed
        out = {
            "metric": baseline_metric() if cfg is CONFIG_B else "Glow fwd+logdet passes/sec on %dx%dx%d mel tiles (K=%d,L=%d)" % (cfg.H, cfg.W, cfg.C, cfg.K, cfg.L),
            "value": value, "unit": "passes/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": precision, "dtype_note": dtype_note[precision],
            "range_guard": "GLOWK_RANGE_ERROR on every call (a hidden activation beyond the fp16 range aborts the run); not tripped",
            "data": "synthetic",
            "config": {"workload": "Glow log_prob, %dx%dx%d mel tiles, L=%d K=%d n_filters=%d, %d tiles/GPU/step"
                                   % (cfg.H, cfg.W, cfg.C, cfg.L, cfg.K, cfg.F, n),
                       "tiles_per_gpu": n, "sharding": "batch shards, 1 all-reduce(sum log-lik)/step"},
            "gflop_per_pass": cfg.flop_per_tile() / 1e9,
            "whole_path_tflops_fp32_equivalent": value / world * cfg.flop_per_tile() / 1e12,
            "hbm_frac_activations": value / world * cfg.act_bytes_per_tile() / 1e9 / PEAK_HBM_GBS,
            "k_net_share_of_step_time": sum(m for m, _ in prof) * 1e-3 / elapsed,
            "roofline": roofline_obj(cfg, precision, prof, res.get("co", False)),
        }
        if "other" in res:
            other, steps_o, elapsed_o, prof_o = res["other"]
            out["accuracy"] = {"max_rel_diff_log_prob_%s_vs_%s_%d_tiles" % (precision, other, n): res["rel_diff"], "north_star_bar": 1e-4}
            out["exact_fp32" if other == "f32" else "split_fp16"] = {
                "value": n * world * steps_o / elapsed_o, "unit": "passes/s", "steps": steps_o, "ms_per_step": elapsed_o / steps_o * 1e3,
                "dtype": other, "roofline": roofline_obj(cfg, other, prof_o) if full else None}
        if "two" in res:
            el2, pr2, d2 = res["two"]
            steps_o = res["other"][1]
            out["two_term_split_fp16"] = {
                "value": n * world * steps_o / el2, "unit": "passes/s", "steps": steps_o, "ms_per_step": el2 / steps_o * 1e3,
                "dtype": "f16x2", "dtype_note": dtype_note["f16x2"], "max_rel_diff_log_prob_vs_f32_%d_tiles" % n: d2,
                "roofline": roofline_obj(cfg, "f16x2", pr2, res.get("co", False)) if full else None}
        return out

    res = measure(cfg, args.precision, args.steps, args.warmup, True)
    out = line(cfg, args.precision, res, args.steps, args.warmup, True) if rank == 0 else None
    # accuracy of the headline arithmetic against the fp64 CPU oracle on two tiles of the same batch (rank 0)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # (N > 1: the other ranks would idle at the barrier meanwhile)
        from oracle import glowref as R
        xs = res["x"][:2].cpu().numpy().astype(np.float64)
        ref = R.log_prob(xs, R.cast_params(res["params"], np.float64), cfg.as_dict())
        out["accuracy"]["max_rel_err_log_prob_vs_fp64_oracle_2_tiles"] = float(np.max(np.abs(res["lp_main"][:2].cpu().numpy() - ref) / np.abs(ref)))
    params_main = res["params"]
    res["eng"].close()
    del res
    torch.cuda.empty_cache()
    # the other north-star shape (32x32x1, K=16, L=2: BASELINE.json configs[1]) in the same run, as a sub-object of the ONE line
    if world == 1 and cfg is CONFIG_B and not args.no_other_shapes:
        ra = measure(CONFIG_A, args.precision, args.steps, args.warmup, True)
        la = line(CONFIG_A, args.precision, ra, args.steps, args.warmup, False)
        out["config_A_32x32_K16_L2"] = {k: la[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "gflop_per_pass",
                                                         "whole_path_tflops_fp32_equivalent", "accuracy", "exact_fp32", "two_term_split_fp16")
                                        if k in la}
        for k in ("exact_fp32", "two_term_split_fp16"):
            if k in out["config_A_32x32_K16_L2"]:
                out["config_A_32x32_K16_L2"][k].pop("roofline", None)
        out["config_A_32x32_K16_L2"]["roofline"] = {k: la["roofline"][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "flop_per_launch")}
        ra["eng"].close()
        del ra
        torch.cuda.empty_cache()
    # the secondary workloads of the path (BASELINE config 5) as compact sub-objects of the ONE line, each a few
    # seconds: the input-gradient path at the headline batch, the training step at the reference's batch of 32, and the BASIS chain
    # on the reference's 30 real mixture tiles with two priors trained here (K = 32, a shortened three-level ladder)
    if world == 1 and cfg is CONFIG_B and not args.no_secondary:
        def guarded_sub(name, fn):     # a secondary workload that fails reports its error in its sub-object; the headline line still prints
            try:
                out[name] = fn()
            except Exception as e:      # noqa: BLE001
                out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            torch.cuda.empty_cache()

        def sub_grad():
            el = grad_workload(CONFIG_B, 1024, PREC[args.precision], 3, 1, rank, local_rank, None, False)
            v = 1024 * 3 / el
            return {"value": v, "unit": "tiles/s", "ms_per_step": el / 3 * 1e3, "dtype": args.precision,
                    "config": {"workload": "log_prob + input gradient, 64x64x1, L=3 K=32 n_filters=512, 1024 tiles"},
                    "range_guard": "GLOWK_RANGE_ERROR, not tripped", "roofline": grad_roofline(CONFIG_B, v, args.precision, 2)}

        def sub_grad30():
            el = grad_workload(CONFIG_B, 30, PREC[args.precision], 20, 3, rank, local_rank, None, False)
            v = 30 * 20 / el
            return {"value": v, "unit": "tiles/s", "ms_per_step": el / 20 * 1e3, "dtype": args.precision,
                    "config": {"workload": "log_prob + input gradient, 64x64x1, L=3 K=32 n_filters=512, 30 tiles (the reference's BASIS batch)"},
                    "range_guard": "GLOWK_RANGE_ERROR, not tripped", "roofline": grad_roofline(CONFIG_B, v, args.precision, 2)}

        def sub_train():
            el, fb, pv = train_workload(CONFIG_B, 32, PREC[args.precision], 8, 2, rank, 1, local_rank, None, False)
            v = 32 * 8 / el
            return {"value": v, "unit": "tiles/s", "ms_per_step": el / 8 * 1e3, "dtype": args.precision, "fallback_sweeps_in_timed_region": fb,
                    "config": {"workload": "training step (loss + all gradients + Adamax + image refresh), 64x64x1, L=3 K=32 n_filters=512, 32 tiles"},
                    "param_vector_floats": pv, "roofline": grad_roofline(CONFIG_B, v, args.precision, 3)}

        def sub_lp32():
            el = logprob_workload(CONFIG_B, 32, PREC[args.precision], 30, 5, rank, local_rank, None, False)
            v = 32 * 30 / el
            return {"value": v, "unit": "passes/s", "ms_per_step": el / 30 * 1e3, "dtype": args.precision,
                    "config": {"workload": "log_prob, 64x64x1, L=3 K=32 n_filters=512, 32 tiles (the reference's batch, configs/melspec_glow.yml:15)"},
                    "range_guard": "GLOWK_RANGE_ERROR, not tripped", "roofline": grad_roofline(CONFIG_B, v, args.precision, 1)}

        guarded_sub("log_prob_32", sub_lp32)
        guarded_sub("log_prob_grad_1024", sub_grad)
        guarded_sub("log_prob_grad_30", sub_grad30)
        guarded_sub("train_32", sub_train)
        guarded_sub("basis_30", lambda: basis_workload(args, 32, 3, 60, 30, 20, 2, rank, 1, local_rank, None, False, args.precision))
    if rank == 0:
        out["git_head"] = git_head()
        out.update(dist_info)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, params_main, args.cpu_budget)
        out["summary"] = summary_of(out)      # LAST and short: the driver's record keeps only the tail of the line
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
