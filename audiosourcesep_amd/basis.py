"""BASIS separation loop (annealed Langevin dynamics with two flow priors) on top of the glowk log_prob + gradient path.

Mirrors the glow branch of the reference's ``run_basis_sep.py``: ``get_sigmas`` (ncsn/utils.py:7-14), the dB mixing
process ``g`` / ``grad_g`` (run_basis_sep.py:131-147), ``basis_inner_loop`` (:152-214) and ``basis_outer_loop`` (:217-260).
On the GPU the per-step arithmetic outside ``compute_grad_logprob`` -- the two noise draws, the dB mixture ``g``, its softmax
weights ``grad_g`` and both updates -- is ONE HIP kernel (``glowk_basis_update``, csrc/glowk_basis.h) with the engine's
counter-based device RNG (Philox4x32-10: the draw of element e at step t is a pure function of (seed, t, e)); torch supplies
storage and streams only.  The noise source stays injectable so that tests can replay the oracle's draws (the reference draws
fresh ``tf.random.normal`` noise, unseeded).  CPU tensors (the host-side tests of ``g`` / ``grad_g``) take the torch formulas.

Convention: ``x1, x2, mixed`` live in the space the two flows were built for -- with
``build_glow(..., data_type='melspec')`` that is dB; the flows' own SpecPreprocessing maps it to the network's range.
Tiles are independent, so ``shard`` splits ``n_mixed`` over the ranks of a process group with no collective in the loop.
At the reference's 30 mixture tiles that alone cannot scale far (a step is a latency-bound chain of ~400 small launches per prior:
4 tiles per GPU take almost as long as 30), so the loop also runs PRIOR-PARALLEL (``prior_parallel_layout``): the two priors of a
tile shard live on two ranks, each evaluates its own prior's gradient, ONE all-gather of the two gradient tensors per Langevin
step (2 x 737 KB for 30 tiles of 96x64) makes both visible, and both ranks run the identical update kernel (same Philox counters:
the replicated state stays bit-identical without a broadcast).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from .distributed import shard_bounds


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _s(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def get_sigmas(sigma1, sigmaL, num_classes, progression="geometric"):
    """ncsn/utils.py:7-14."""
    if progression == "geometric":
        sigmas = np.exp(np.linspace(np.log(sigma1), np.log(sigmaL), num=num_classes))
    elif progression == "logarithmic":
        sigmas = np.logspace(np.log(sigma1) / np.log(10), np.log(sigmaL) / np.log(10), num=num_classes)
    else:
        raise ValueError("progression should be geometric or logarithmic")
    return sigmas.astype(np.float32)


def mixing_db(*sources):
    """``g`` of the dB branch, sum in power (run_basis_sep.py:133-141): 10/ln10 * (logsumexp(s ln10/10) - ln K)."""
    k = len(sources)
    if k == 2 and sources[0].is_cuda and sources[0].dtype == torch.float32:
        a, b = sources[0].contiguous(), sources[1].contiguous()
        out = torch.empty_like(a)
        _lib.check(_lib.load().glowk_basis_mix(_p(a), _p(b), _p(out), a.numel(), _s(a)))
        return out
    s = torch.stack(sources, dim=0)
    return (10.0 / math.log(10.0)) * (torch.logsumexp(s * (math.log(10.0) / 10.0), dim=0) - math.log(float(k)))


def device_randn(shape, device, seed, step=0, which=0, uniform=False, offset=0):
    """Standard-normal (or U(0, 1)) tensor from the engine's Philox stream (seed, step, which): the draws
    ``glowk_basis_update`` makes itself when no noise is injected.  ``offset``: position of element 0 in the stream (a
    multiple of 4) -- a rank holding tiles [a, b) of a batch passes ``a * H * W * C`` and gets the draws one process would have
    made for those tiles, so the noise a tile sees does not depend on how the batch is sharded."""
    out = torch.empty(shape, device=device, dtype=torch.float32)
    _lib.check(_lib.load().glowk_random(_p(out), out.numel(), int(seed), int(step), int(which), int(bool(uniform)), int(offset), _s(out)))
    return out


def add_device_noise(x, sigma, seed, step=0, which=2, offset=0):
    """x + sigma * N(0, I) as ONE kernel of the engine (``glowk_add_noise``; the draws are ``device_randn(seed, step, which,
    offset)``): train_noisy_glow.py:31."""
    x = x.contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.load().glowk_add_noise(_p(x), _p(out), x.numel(), float(sigma), int(seed), int(step), int(which), int(offset), _s(x)))
    return out


def langevin_update(mixed, x1, x2, g1, g2, eta, lambda_recon, eps1=None, eps2=None, seed=0, step=0, nonfinite=None, offset=0):
    """run_basis_sep.py:163-181 for two sources, IN PLACE on x1 / x2 (contiguous float32 CUDA tensors): one kernel."""
    _lib.check(_lib.load().glowk_basis_update(_p(x1), _p(x2), _p(g1), _p(g2), _p(mixed), x1.numel(), float(eta), float(lambda_recon),
                                              _p(eps1), _p(eps2), int(seed), int(step), int(offset), _p(nonfinite), _s(x1)))


def grad_mixing_db(*sources):
    """``grad_g`` (run_basis_sep.py:143-147): softmax over the sources of s ln10/10."""
    s = torch.stack(sources, dim=0)
    return torch.unbind(torch.softmax(s * (math.log(10.0) / 10.0), dim=0), dim=0)


def prior_parallel_layout(n_mixed, world_size, rank):
    """Prior-parallel BASIS over ``world_size`` = 2 S ranks: rank r holds prior ``r % 2`` (0: model1, 1: model2) of tile shard
    ``r // 2`` of S.  -> dict(prior, shard, n_shards, bounds=(a, b), pair=(rank of prior 0, rank of prior 1)).  Every rank of the
    job then creates the pair groups in the same order: ``[dist.new_group(list(p)) for p in all_pairs(world_size)]`` and keeps its own
    (``make_pair_group``)."""
    if world_size < 2 or world_size % 2:
        raise ValueError("prior-parallel BASIS needs an even number of ranks (two priors per tile shard)")
    shards = world_size // 2
    s = rank // 2
    return {"prior": rank % 2, "shard": s, "n_shards": shards, "bounds": shard_bounds(n_mixed, shards, s), "pair": (2 * s, 2 * s + 1)}


def make_pair_group(world_size, rank):
    """The process group of this rank's prior pair.  Collective over the whole job (torch.distributed.new_group must be called by
    every rank for every group, in the same order)."""
    import torch.distributed as dist
    mine = None
    for s in range(world_size // 2):
        g = dist.new_group([2 * s, 2 * s + 1])
        if rank // 2 == s:
            mine = g
    return mine


def exchange_prior_gradients(g_mine, prior_index, pair_group):
    """-> (g1, g2): all-gather of the two priors' gradients over the pair (rank order within the pair = prior order).  RCCL over
    xGMI when the backend is nccl; under gloo (CPU tests, one-GPU rehearsal) device tensors go through the host."""
    import torch.distributed as dist
    g_mine = g_mine.contiguous()
    if g_mine.is_cuda and dist.get_backend(pair_group) == "gloo":
        mine = g_mine.cpu()
        parts = [torch.empty_like(mine), torch.empty_like(mine)]
        dist.all_gather(parts, mine, group=pair_group)
        parts = [p.to(g_mine.device) for p in parts]
    else:
        parts = [torch.empty_like(g_mine), torch.empty_like(g_mine)]
        dist.all_gather(parts, g_mine, group=pair_group)
    parts[prior_index] = g_mine
    return parts[0], parts[1]


def compute_grad_logprob(inputs, model):
    """run_basis_sep.py:73-79 -- d log_prob / d inputs through the engine (no autograd tape needed)."""
    _, g = model.engine.log_prob_grad(inputs)
    return g


def _grad_pair(x1, x2, model1, model2, streams):
    """The two priors' gradients are independent: evaluate them concurrently on two HIP streams (at BASIS batch sizes the
    deeper levels launch only tens of workgroups each, so the two kernel sequences interleave on the 256 CUs).

    The range guard of the split arithmetics would make each call wait for its own stream before returning (policy "error" /
    "fallback"), i.e. serialise the two models; so both calls run under "ignore" (fully asynchronous) and the flags are read
    once both sequences are enqueued -- a tripped one re-runs that model's gradient on the exact fp32 kernels, like "fallback"."""
    if streams is None or x1.device.type != "cuda":
        return compute_grad_logprob(x1, model1), compute_grad_logprob(x2, model2)
    cur = torch.cuda.current_stream(x1.device)
    s1, s2 = streams
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    engines = (model1.engine, model2.engine)
    guarded = [e.get_precision() != _lib.PREC_F32 for e in engines]
    saved = [int(e.lib.glowk_get_range_policy(e.h)) for e in engines]
    for e, gd in zip(engines, guarded):
        if gd:
            e.set_range_policy("ignore")
    try:
        with torch.cuda.stream(s1):
            g1 = compute_grad_logprob(x1, model1)
        with torch.cuda.stream(s2):
            g2 = compute_grad_logprob(x2, model2)
        out = [g1, g2]
        for i, (e, gd, st, x, m) in enumerate(zip(engines, guarded, (s1, s2), (x1, x2), (model1, model2))):
            if not gd or saved[i] == _lib.RANGE_IGNORE:
                continue
            with torch.cuda.stream(st):
                if e.range_status()[0]:
                    if saved[i] == _lib.RANGE_ERROR:
                        raise _lib.GlowkRangeError("BASIS: a prior's gradient left the fp16 range of the split arithmetic")
                    prec = e.get_precision()
                    e.set_precision(_lib.PREC_F32)
                    out[i] = compute_grad_logprob(x, m)
                    e.set_precision(prec)
        g1, g2 = out
    finally:
        for e, gd, pol in zip(engines, guarded, saved):
            if gd:
                e.set_range_policy(pol)
    cur.wait_stream(s1)
    cur.wait_stream(s2)
    g1.record_stream(cur)
    g2.record_stream(cur)
    return g1, g2


def basis_inner_loop(mixed, x1, x2, model1, model2, sigma_idx, sigmas, delta=2e-5, T=100, noise_fn=None, debug=False,
                     streams="auto", seed=0, step0=0, offset=0, prior_group=None, prior_index=None):
    """run_basis_sep.py:152-214 (model_type == 'glow').  ``noise_fn(t, which, shape) -> standard normal tensor`` replays given
    draws; without it the update kernel draws from the device RNG stream (seed, step0 + t), element ``offset`` onwards
    (``offset`` = this shard's first tile * H * W * C: the draws of a tile are the same whatever the sharding).
    ``streams``: "auto" (two side streams when on the GPU and the models are distinct engines), None, or (s1, s2).
    ``debug``: the reference's NaN asserts (:183-191), from a flag the update kernel raises (one word read back per step).
    ``prior_group`` / ``prior_index``: prior-parallel mode -- this rank evaluates the gradient of prior ``prior_index`` only (the
    other model may be None), the pair exchanges the two gradients (``exchange_prior_gradients``) and both ranks take the same
    update; the noise must then be the same on both (the device RNG with equal seed / step0 / offset is; an injected ``noise_fn``
    has to be)."""
    pp = prior_group is not None
    if pp and prior_index not in (0, 1):
        raise ValueError("prior_index must be 0 or 1 in prior-parallel mode")
    if pp:
        streams = None
    if streams == "auto":
        streams = None
        if x1.device.type == "cuda" and getattr(model1, "engine", None) is not getattr(model2, "engine", None):
            streams = (torch.cuda.Stream(device=x1.device), torch.cuda.Stream(device=x1.device))
    sigma = float(sigmas[sigma_idx])
    sigma_l = float(sigmas[-1])
    eta = float(np.float32(delta * (sigma / sigma_l) ** 2))
    lambda_recon = 1.0 / (sigma ** 2)
    def grads(a, b):
        if not pp:
            return _grad_pair(a, b, model1, model2, streams)
        mine = compute_grad_logprob(a if prior_index == 0 else b, model1 if prior_index == 0 else model2)
        return exchange_prior_gradients(mine, prior_index, prior_group)

    if x1.device.type != "cuda":
        return _inner_loop_host(mixed, x1, x2, grads, eta, lambda_recon, T, noise_fn, debug)
    mixed = mixed.to(torch.float32).contiguous()
    x1, x2 = x1.to(torch.float32).clone().contiguous(), x2.to(torch.float32).clone().contiguous()   # (the update is in place)
    flag = torch.zeros(1, dtype=torch.int32, device=x1.device) if debug else None
    for t in range(T):
        g1, g2 = grads(x1, x2)
        e1 = noise_fn(t, 0, x1.shape).to(torch.float32).contiguous() if noise_fn is not None else None
        e2 = noise_fn(t, 1, x2.shape).to(torch.float32).contiguous() if noise_fn is not None else None
        langevin_update(mixed, x1, x2, g1, g2, eta, lambda_recon, e1, e2, seed=seed, step=step0 + t, nonfinite=flag, offset=offset)
        if debug:
            assert int(flag.item()) == 0, (sigma, t)   # run_basis_sep.py:183-191
    return x1, x2


def _inner_loop_host(mixed, x1, x2, grads, eta, lambda_recon, T, noise_fn, debug):
    """The same loop on torch formulas (CPU tensors: host-side tests with stand-in models).  ``grads(x1, x2) -> (g1, g2)``."""
    if noise_fn is None:
        noise_fn = lambda t, which, shape: torch.randn(shape, dtype=torch.float32)  # noqa: E731
    for t in range(T):
        eps1 = math.sqrt(2.0 * eta) * noise_fn(t, 0, x1.shape)
        eps2 = math.sqrt(2.0 * eta) * noise_fn(t, 1, x2.shape)
        g1, g2 = grads(x1, x2)
        mix = mixing_db(x1, x2)
        m1, m2 = grad_mixing_db(x1, x2)
        x1, x2 = x1 + eta * (g1 + lambda_recon * m1 * (mixed - mix)) + eps1, x2 + eta * (g2 + lambda_recon * m2 * (mixed - mix)) + eps2
        if debug:
            assert torch.isfinite(x1).all() and torch.isfinite(x2).all(), t
    return x1, x2


def basis_outer_loop(mixed, x1, x2, model1, model2, sigmas, restore_1=None, restore_2=None, T=100, delta=2e-5, noise_fn=None,
                     debug=False, seed=0, tile_offset=0, prior_group=None, prior_index=None):
    """run_basis_sep.py:217-260.  ``restore_k``: optional ``{sigma: state_dict | path | GlowFlow}`` with the noise-conditioned
    weights of model k for each noise level (the per-sigma checkpoints of train_noisy_glow.py:309-358); a ``GlowFlow`` value is
    used as is (all ten noise levels of both priors resident: 2 x 10 x 0.5 GB of packed weights).
    ``tile_offset``: index of ``mixed[0]`` in the whole set of mixture tiles (``shard_bounds(n_mixed, world, rank)[0]`` on a rank
    that holds a shard): folded into the device RNG's counter, so every tile sees the Langevin noise it would see in a
    one-process run -- ranks do not repeat each other's draws and the result does not depend on the world size.
    ``prior_group`` / ``prior_index``: prior-parallel mode (``basis_inner_loop``): only the own prior's model and ``restore_k`` are
    used, the other may be None."""
    elems_per_tile = int(np.prod(mixed.shape[1:]))
    x_arr = {"x1": [x1.cpu().numpy()], "x2": [x2.cpu().numpy()]}
    for sigma_idx, sigma in enumerate(sigmas):
        current = []
        for k, (model, restore) in enumerate(((model1, restore_1), (model2, restore_2))):
            if prior_group is not None and k != prior_index:
                current.append(None)          # the pair partner owns this prior
                continue
            if restore is not None:
                state = restore[float(sigma)] if float(sigma) in restore else restore[sigma]
                if hasattr(state, "log_prob"):      # a resident flow for this noise level: no weight swap at all
                    model = state
                elif isinstance(state, str):
                    model.restore(state)            # ~0.4 s for config B (host re-pack on 16 threads + 0.5 GB upload)
                else:
                    model.load_state_dict(state)
            current.append(model)
        model1_s, model2_s = current
        nf = None if noise_fn is None else (lambda t, which, shape, _s=sigma_idx: noise_fn(_s, t, which, shape))
        x1, x2 = basis_inner_loop(mixed, x1, x2, model1_s, model2_s, sigma_idx, sigmas, delta=delta, T=T, noise_fn=nf, debug=debug,
                                  seed=seed, step0=sigma_idx * T, offset=int(tile_offset) * elems_per_tile, prior_group=prior_group,
                                  prior_index=prior_index)
        x_arr["x1"].append(x1.cpu().numpy())
        x_arr["x2"].append(x2.cpu().numpy())
    return x1, x2, x_arr


def shard(tensor, world_size, rank):
    """This rank's contiguous slice of the ``n_mixed`` tiles (tiles are independent: no collective inside the loop)."""
    a, b = shard_bounds(tensor.shape[0], world_size, rank)
    return tensor[a:b]
