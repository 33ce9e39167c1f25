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


def cpu_baseline(cfg, params, budget_s=20.0):
    """Oracle port on the host cores: torch-CPU fp32, one network evaluation per step (deduplicated graph).
    Bounded sample: one tile is timed first, then as many tiles as fit the time budget (1..16)."""
    from oracle import glowref_torch as RT
    threads = host_threads()
    torch.set_num_threads(threads)
    p = RT.to_torch(params, torch.float32)
    x = torch.from_numpy(synthetic_mel_tiles(128, cfg, seed=4321))
    with torch.no_grad():
        t0 = time.perf_counter()
        RT.log_prob(x[:1], p, cfg.as_dict())          # also warms up primitive creation
        t1 = time.perf_counter() - t0
        tiles = int(max(1, min(128, budget_s / max(t1, 1e-3))))
        t0 = time.perf_counter()
        RT.log_prob(x[:tiles], p, cfg.as_dict())
        dt = time.perf_counter() - t0
    return {
        "value": tiles / dt, "unit": "passes/s", "cores": threads, "kind": "port",
        "sample": "%d tiles x 1 pass of the same config, torch-CPU fp32 restatement of the reference graph "
                  "(oracle/glowref_torch.py; not TensorFlow), %.1f s after a %.1f s one-tile warm-up" % (tiles, dt, t1),
    }


def secondary_workload(args, cfg, eng, params, rank, world, local_rank, dist):
    """log_prob_grad: one step = log_prob + d/dx over the resident batch.  basis: one step = one Langevin update of the
    BASIS loop (two flow priors, run_basis_sep.py:163-181) over ``--batch`` mixture tiles per GPU (reference: 30)."""
    from audiosourcesep_amd import basis
    from audiosourcesep_amd.flow_models.flow_glow import GlowFlow
    from audiosourcesep_amd.synthetic import calibrated_engine
    n = args.batch
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=1234 + rank)).cuda()
    if args.workload == "log_prob_grad":
        def step():
            eng.log_prob_grad(x)
        unit, metric, per_step = "tiles/s", "Glow log_prob + input-gradient tiles/sec", n
    else:
        eng2, _ = calibrated_engine(cfg, device=local_rank, init_tiles=max(n, 64), seed=4048)
        eng2.set_precision(eng.get_precision())
        m1, m2 = GlowFlow(eng), GlowFlow(eng2)
        x2 = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=4321 + rank)).cuda()
        mixed = basis.mixing_db(x, x2)
        # state: mel-like tiles (other seeds), not the reference's uniform draw -- the synthetic priors are far out of their
        # domain on uniform noise (log_prob ~ -1e33), and timing a loop that carries inf/NaN would not be a measurement
        state = {"x1": torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=777 + rank)).cuda(),
                 "x2": torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=888 + rank)).cuda()}
        sigmas = basis.get_sigmas(1.0, 0.01, 10)

        def step():   # every step starts from the same state: with these untrained priors the chain itself diverges within ~8 steps
            state["y1"], state["y2"] = basis.basis_inner_loop(mixed, state["x1"], state["x2"], m1, m2, 9, sigmas, T=1)
        unit, metric, per_step = "tile-steps/s", "BASIS Langevin tile-steps/sec (2 Glow priors)", n
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if args.workload == "basis":
        assert torch.isfinite(state["y1"]).all() and torch.isfinite(state["y2"]).all(), "BASIS update left the finite range"
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if dist.get_backend() == "gloo" else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": metric, "value": per_step * world * args.steps / elapsed, "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "%s, %dx%dx%d tiles, L=%d K=%d n_filters=%d, %d tiles/GPU" % (args.workload, cfg.H, cfg.W, cfg.C, cfg.L, cfg.K, cfg.F, n)},
        }), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="tiles per GPU per step")
    ap.add_argument("--config", default="B", choices=["A", "B", "YAML"])
    ap.add_argument("--workload", default="log_prob", choices=["log_prob", "log_prob_grad", "basis"],
                    help="log_prob = BASELINE.json's headline metric; the other two are secondary lines (SURVEY section 8f-1)")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3", "f16x2"],
                    help="f16x3: error-compensated fp16 split on the fp16 MFMA (fp32-class accuracy, demonstrated in the line); "
                         "f32: exact fp32-input MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work for the cpu_baseline sample")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
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

    cfg = {"A": CONFIG_A, "B": CONFIG_B, "YAML": CONFIG_YAML}[args.config]
    from audiosourcesep_amd import _lib
    from audiosourcesep_amd.synthetic import calibrated_engine
    # synthetic weights + ActNorm data-dependent init on a minibatch of the benchmark's own batch size, so that
    # every k_net launch of the process has the same grid (rocprof's per-kernel average == the timed one)
    eng, params = calibrated_engine(cfg, device=local_rank, init_tiles=args.batch)
    PREC = {"f32": _lib.PREC_F32, "f16x3": _lib.PREC_F16X3, "f16x2": _lib.PREC_F16X2}
    eng.set_precision(PREC[args.precision])
    n = args.batch
    eng.reserve(n)
    if args.workload != "log_prob":
        secondary_workload(args, cfg, eng, params, rank, world, local_rank, dist)
        return
    x = torch.from_numpy(synthetic_mel_tiles(n, cfg, seed=1234 + rank)).cuda()   # resident in HBM before timing
    lp = torch.empty(n, device="cuda", dtype=torch.float32)
    total = torch.zeros(1, device="cuda", dtype=torch.float64)

    from audiosourcesep_amd.distributed import sharded_log_prob

    def step():
        # local shard on this GPU, then ONE all-reduce of the fp64 summed log-likelihood (RCCL over xGMI)
        _, tot = sharded_log_prob(lambda xx: eng.log_prob(xx, out=lp), x)
        total.copy_(tot)

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

    for _ in range(args.warmup):
        step()
    elapsed, prof = timed(args.steps)
    assert torch.isfinite(total).all(), "non-finite log-likelihood"
    lp_main = lp.clone()
    # the other arithmetic on the same batch: exact fp32 beside the split path (or vice versa), same run
    other = "f32" if args.precision != "f32" else "f16x3"
    eng.set_precision(PREC[other])
    step()
    elapsed_o, prof_o = timed(max(2, args.steps // 2))
    steps_o = max(2, args.steps // 2)
    lp_other = lp.clone()
    rel_diff = float(((lp_main - lp_other).abs() / lp_other.abs()).max().item())
    # the throughput mode (two split terms per product: inside the 1e-4 bar, not fp32-class), same batch, same run
    two = None
    if args.precision == "f16x3":
        eng.set_precision(PREC["f16x2"])
        step()
        elapsed_2, prof_2 = timed(steps_o)
        two = (elapsed_2, prof_2, float(((lp - lp_other).abs() / lp_other.abs()).max().item()))
    eng.set_precision(PREC[args.precision])
    # accuracy of the headline arithmetic against the fp64 CPU oracle on two tiles of the same batch (rank 0)
    acc_vs_oracle = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # (N > 1: the other ranks would idle at the barrier meanwhile)
        from oracle import glowref as R
        xs = x[:2].cpu().numpy().astype(np.float64)
        ref = R.log_prob(xs, R.cast_params(params, np.float64), cfg.as_dict())
        acc_vs_oracle = float(np.max(np.abs(lp_main[:2].cpu().numpy() - ref) / np.abs(ref)))

    if rank == 0:
        passes = n * world * args.steps
        value = passes / elapsed
        h0, w0, c0 = cfg.level_shapes()[0]
        flop_launch = net_flop_per_pixel(c0, cfg.F) * n * h0 * w0

        def roofline_obj(precision, pr):
            ms0, launches0 = pr[0]
            avg_ms = ms0 / max(launches0, 1)
            achieved = flop_launch / (avg_ms * 1e-3) / 1e12 if launches0 else None
            if precision == "f32":
                kernel, peak = "k_net_f32<CI=%d,NF=%d> (level 0)" % (c0 // 2, cfg.F // 32), PEAK_F32_MFMA_TFLOPS
                note = "fp32-input MFMA peak"
            elif precision == "f16x2":
                kernel, peak = "k_net_h3s<CI=%d,NF=%d,two-term> (level 0)" % (c0 // 2, cfg.F // 32), PEAK_F16_MFMA_TFLOPS / 2.0
                note = "fp16 dense MFMA peak / 2 (two fp16 MFMAs per product)"
            else:
                kernel, peak = "k_net_h3s<CI=%d,NF=%d> (level 0)" % (c0 // 2, cfg.F // 32), PEAK_F16_MFMA_TFLOPS / 3.0
                note = "fp16 dense MFMA peak / 3 (three fp16 MFMAs per fp32-equivalent product)"
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
            if os.path.exists(tpath):
                try:
                    key = "k_net_level0_hbm_bytes_per_launch_per_tile" if precision == "f32" else "k_net_h3_level0_hbm_bytes_per_launch_per_tile"
                    t = json.load(open(tpath)).get(key)
                    traffic = t * n if t is not None else None
                except Exception:
                    traffic = None
            # MFMA-pipe busy fraction of the same kernel from PMC counters (profiles/mfma_utilisation.json, scripts/pmc_mfma.py)
            mfma_busy = None
            upath = os.path.join(ROOT, "profiles", "mfma_utilisation.json")
            if os.path.exists(upath) and args.config == "B":
                try:
                    kk = {"f32": "void k_net_f32<2, 36, 16, 0>(NetArgs)", "f16x3": "void k_net_h3s<2, 36, 16, 0, 2, false>(NetArgs)",
                          "f16x2": "void k_net_h3s<2, 36, 16, 3, 2, false>(NetArgs)"}[precision]
                    mfma_busy = json.load(open(upath))["kernels"][kk]["mfma_utilisation"]
                except Exception:
                    mfma_busy = None
            # what a bare MFMA loop with this kernel's operand pattern sustains on this chip under its power management
            # (scripts/mfma_shape.hip, random data): 1 697 TFLOP/s for 16x16x32 f16 (/3), 150.8 TFLOP/s for 32x32x2 f32
            sustained = SUSTAINED_F32_MFMA_TFLOPS if precision == "f32" else SUSTAINED_F16_MFMA_TFLOPS / (2.0 if precision == "f16x2" else 3.0)
            return {"kernel": kernel, "bound": "mfma", "achieved": achieved, "peak": peak, "peak_note": note, "unit": "TFLOP/s",
                    "frac": (achieved / peak) if achieved else None, "traffic": traffic, "mfma_busy_pmc": mfma_busy, "avg_launch_ms": avg_ms,
                    "launches": launches0, "flop_per_launch": flop_launch, "sustained_mfma_rate_measured": sustained,
                    "frac_of_sustained": (achieved / sustained) if achieved else None}

        passes = n * world * args.steps
        value = passes / elapsed
        value_o = n * world * steps_o / elapsed_o
        net_ms_total = sum(m for m, _ in prof)
        dtype_name = {"f32": "f32", "f16x3": "f16x3", "f16x2": "f16x2"}
        dtype_note = {"f32": "fp32 operands on v_mfma_f32_32x32x2_f32, fp32 accumulate",
                      "f16x3": "fp32 operands split into fp16 hi + lo, 3 fp16 MFMAs per product, fp32 accumulate",
                      "f16x2": "weights split into fp16 hi + lo, activations rounded to fp16, 2 fp16 MFMAs per product, fp32 accumulate"}
        out = {
            "metric": baseline_metric() if args.config == "B" else "Glow fwd+logdet passes/sec (config %s)" % args.config,
            "value": value, "unit": "passes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype_name[args.precision], "dtype_note": dtype_note[args.precision],
            "data": "synthetic",
            "config": {"workload": "Glow log_prob, %dx%dx%d mel tiles, L=%d K=%d n_filters=%d, %d tiles/GPU/step"
                                   % (cfg.H, cfg.W, cfg.C, cfg.L, cfg.K, cfg.F, n),
                       "tiles_per_gpu": n, "sharding": "batch shards, 1 all-reduce(sum log-lik)/step"},
            "gflop_per_pass": cfg.flop_per_tile() / 1e9,
            "whole_path_tflops_fp32_equivalent": value / world * cfg.flop_per_tile() / 1e12,
            "hbm_frac_activations": value / world * cfg.act_bytes_per_tile() / 1e9 / PEAK_HBM_GBS,
            "k_net_share_of_step_time": net_ms_total * 1e-3 / elapsed,
            "accuracy": {"max_rel_err_log_prob_vs_fp64_oracle_2_tiles": acc_vs_oracle,
                         "max_rel_diff_log_prob_%s_vs_%s_%d_tiles" % (args.precision, other, n): rel_diff,
                         "north_star_bar": 1e-4},
            "roofline": roofline_obj(args.precision, prof),
            ("exact_fp32" if other == "f32" else "split_fp16"): {
                "value": value_o, "unit": "passes/s", "steps": steps_o, "ms_per_step": elapsed_o / steps_o * 1e3,
                "dtype": dtype_name[other], "roofline": roofline_obj(other, prof_o)},
        }
        if two is not None:
            out["two_term_split_fp16"] = {
                "value": n * world * steps_o / two[0], "unit": "passes/s", "steps": steps_o, "ms_per_step": two[0] / steps_o * 1e3,
                "dtype": "f16x2", "dtype_note": dtype_note["f16x2"], "max_rel_diff_log_prob_vs_f32_%d_tiles" % n: two[2],
                "roofline": roofline_obj("f16x2", two[1])}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, params, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
