"""Noise-conditioned Glow priors for BASIS: the serial fine-tuning ladder of the reference's ``train_noisy_glow.py``.

The reference trains ONE Glow per noise level of the BASIS schedule: for every sigma of ``get_sigmas(sigma1, sigmaL, n)`` --
largest first -- it restores the previous level's checkpoint (the clean model for the first level), trains on
``X + tf.random.normal(X.shape) * sigma`` (train_noisy_glow.py:29-31) with the ordinary training step (:37-44) and saves the
result under ``sigma_<sigma>/tf_ckpts`` (:309-358); ``run_basis_sep.py:217-260`` then restores model k's checkpoint of the
current sigma before every inner loop.  Here the same ladder runs on the engine's own training step
(``GlowFlow.train_step(noise_std=sigma)``: split-arithmetic parameter-gradient sweep, Adamax, device-side refresh of the kernel
images), and the per-sigma weights are kept either as state dicts (what ``basis_outer_loop(restore_k=...)`` swaps in) or as
resident flows, one engine per noise level (no weight swap inside the chain: 288 GB of HBM hold all of them).

Units.  The flows are built with ``data_type='melspec'`` and own their SpecPreprocessing, so tiles, chain state and sigma live
in dB.  The reference's BASIS script works on tiles normalised to [0, 1] (run_basis_sep.py:352-356) with
sigma from 1.0 to 0.01 and delta = 2e-5; the same dynamics in dB are sigma_dB = (maxval - minval) * sigma and
delta_dB = (maxval - minval)^2 * delta (``db_schedule``).
"""
import numpy as np
import torch

from .basis import get_sigmas
from .engine import GlowEngine
from .flow_models.flow_glow import GlowFlow


def db_schedule(cfg, sigma1=1.0, sigmaL=0.01, num_classes=10, delta=2e-5, progression="geometric"):
    """-> (sigmas in dB, delta in dB^2): the reference's normalised-unit schedule (run_basis_sep.py:465-468, :152-161) carried to
    the dB space the melspec flows live in."""
    span = float(cfg.maxval - cfg.minval)
    return get_sigmas(sigma1, sigmaL, num_classes, progression) * np.float32(span), float(delta) * span * span


def clone_flow(flow, precision=None):
    """A second engine with the same configuration and variables (a resident copy for one noise level)."""
    eng = GlowEngine(flow.cfg, device=flow.engine.device.index)
    new = GlowFlow(eng)
    new.load_state_dict(flow.state_dict())
    eng.set_precision(flow.engine.get_precision() if precision is None else precision)
    eng.set_range_policy(int(flow.engine.lib.glowk_get_range_policy(flow.engine.h)))
    return new


def fine_tune_ladder(flow, tiles, sigmas, steps_per_level, lr=1e-3, optimizer="adamax", batch_size=None, seed=0, group=None,
                     resident=True, on_step=None):
    """train_noisy_glow.py:309-358: for sigma in sigmas (in the order given: the reference goes from the largest down), fine-tune
    ``flow`` -- continuing from the previous level's weights -- for ``steps_per_level`` training steps on ``tiles + N(0, sigma^2)``
    (fresh noise every step: the flow's own step counter drives the device RNG) and keep the level's weights.

    tiles: [n, H, W, C] float32 on the flow's device (this rank's shard when ``group`` is given); ``batch_size`` (default: all
    tiles) are taken round-robin per step.  ``steps_per_level``: int or one int per level.
    Returns ``{float(sigma): GlowFlow}`` (``resident``: one engine per level, ready for ``basis_outer_loop(restore_k=...)``) or
    ``{float(sigma): state_dict}``, plus the per-level lists of losses: ``(models, losses)``."""
    tiles = flow.engine._in(tiles, flow.engine.data_shape)
    n = tiles.shape[0]
    bs = n if batch_size is None else min(int(batch_size), n)
    steps = [int(steps_per_level)] * len(sigmas) if np.isscalar(steps_per_level) else [int(s) for s in steps_per_level]
    if len(steps) != len(sigmas):
        raise ValueError("steps_per_level must be an int or one int per sigma")
    world = 1
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        world = torch.distributed.get_world_size(group)
    elif group is not None:
        raise ValueError("fine_tune_ladder: a process group was given but torch.distributed is not initialised")
    models, losses = {}, {}
    pos = 0
    for sigma, n_steps in zip(sigmas, steps):
        level_losses = []
        for t in range(n_steps):
            idx = (torch.arange(bs, device=tiles.device) + pos) % n
            pos = (pos + bs) % n
            xb = tiles if bs == n else tiles.index_select(0, idx)
            loss = flow.train_step(xb, optimizer=optimizer, lr=lr, global_batch_size=bs * world, group=group, noise_std=float(sigma), seed=seed)
            level_losses.append(loss)
            if on_step is not None:
                on_step(float(sigma), t, loss)
        losses[float(sigma)] = [float(v) for v in level_losses]      # (one host read per level, after its last step)
        models[float(sigma)] = clone_flow(flow) if resident else flow.state_dict()
    return models, losses


def psnr_db(estimate, truth, span=120.0):
    """Peak signal-to-noise ratio of dB tiles against the ground truth, peak = the dB range of the representation."""
    e = torch.as_tensor(estimate, dtype=torch.float64).cpu()
    t = torch.as_tensor(truth, dtype=torch.float64).cpu()
    return float(10.0 * torch.log10(span * span / torch.mean((e - t) ** 2)))
