"""Batch sharding of the Glow path over the GPUs of one node.

Tiles are independent (no cross-sample op anywhere in log_prob), so the path shards with no data-path collective:
one process per GPU, weights replicated, contiguous batch shards.  The only exchange is the all-reduce of the
summed log-likelihood (1 fp64 element per batch; RCCL over xGMI when the backend is "nccl").  The helpers take the
local ``log_prob`` callable, so the same code runs under gloo on CPU in the tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world_size, rank):
    """Contiguous shard [start, stop) of ``n`` tiles for ``rank``; the first ``n % world_size`` ranks get one extra."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world size")
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def sharded_log_prob(log_prob_fn, x_local, group=None, total=None, out=None):
    """Evaluate the local shard and all-reduce the summed log-likelihood.

    Returns (lp_local [n_local] as produced by ``log_prob_fn``, total fp64 scalar tensor identical on every rank).
    The sum is accumulated in fp64 so that its value does not depend on how tiles were sharded to ~1e-12.
    ``log_prob_fn`` may be a :class:`GlowEngine` (or anything with ``log_prob_sum``): the engine then leaves the fp64 sum on the
    device itself (``glowk_log_prob_sum``, fixed summation order) in ``total`` (a [1] float64 tensor, allocated if missing), and
    the only thing between the engine's kernels and the collective is the collective -- no tensor-library kernel."""
    if hasattr(log_prob_fn, "log_prob_sum"):
        lp, total = log_prob_fn.log_prob_sum(x_local, out=out, total=total)
    else:
        lp = log_prob_fn(x_local) if x_local.shape[0] else torch.zeros(0, dtype=torch.float32, device=x_local.device)
        total = lp.sum(dtype=torch.float64).reshape(1)
    if dist.is_available() and dist.is_initialized():   # (a one-rank group still makes the call: same code path at every N)
        if total.is_cuda and dist.get_backend(group) == "gloo":   # rehearsal of the multi-process path without RCCL
            t = total.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            total.copy_(t)
        else:
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return lp, total[0]


def distributed_test_step(log_prob_fn, x_local, global_batch_size, group=None):
    """The evaluation step of the trainer (train_glow.py:33-35, 48-50, 64-68): every replica computes
    ``sum(-log_prob(X_local)) / global_batch_size`` (tf.nn.compute_average_loss) and the per-replica losses are summed --
    i.e. the mean negative log-likelihood of the global batch, identical on every rank (fp64)."""
    if global_batch_size <= 0:
        raise ValueError("global_batch_size must be positive")
    _, total = sharded_log_prob(log_prob_fn, x_local, group=group)
    return -total / float(global_batch_size)


def _all_reduce_sum(t, group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":   # rehearsal of the multi-process path without RCCL
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def distributed_train_step(param_grad_fn, apply_fn, x_local, global_batch_size, group=None):
    """The training step of the trainer (train_glow.py:29-31, 37-44, 52-54) over batch shards, one process per GPU.

    ``param_grad_fn(x_local, scale) -> (log_prob [n_local], grad [P])`` returns this rank's
    ``scale * d sum_n log_prob(x_n) / d theta`` as ONE flat fp32 vector (``GlowEngine.param_grad``); with
    ``scale = -1 / global_batch_size`` that is the gradient of the replica's share of ``tf.nn.compute_average_loss``.
    MirroredStrategy sums the per-replica gradients with an all-reduce before ``apply_gradients``: here ONE all-reduce(sum) of
    the flat vector (RCCL over xGMI when the backend is "nccl": 128 MB for the 64x64 K=32 L=3 flow -- ring all-reduce is bound
    by one xGMI link, ~153 GB/s: ~1.5 ms at 8 GPUs), then every rank takes the identical optimizer step
    (``apply_fn(grad)``), so the replicas stay bit-identical without a broadcast.  Returns the global loss (fp64 scalar tensor,
    the same on every rank; ``strategy.reduce(SUM, per_replica_losses)``, :54)."""
    if global_batch_size <= 0:
        raise ValueError("global_batch_size must be positive")
    lp, grad = param_grad_fn(x_local, -1.0 / float(global_batch_size))
    eng = getattr(param_grad_fn, "__self__", None)
    if lp.is_cuda and hasattr(eng, "sum_f64"):    # the engine's own fixed-order fp64 reduction (no tensor-library kernel in the step)
        loss = eng.sum_f64(lp, scale=-1.0 / float(global_batch_size))
    else:
        loss = (-lp.sum(dtype=torch.float64) / float(global_batch_size)).reshape(1)
    _all_reduce_sum(grad, group)
    _all_reduce_sum(loss, group)
    apply_fn(grad)
    return loss[0]


def gather_log_prob(lp_local, n_total, group=None):
    """All-gather the per-tile log_prob vectors of contiguous shards back into batch order ([n_total] on every rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return lp_local
    world = dist.get_world_size(group)
    cap = (n_total + world - 1) // world
    buf = torch.zeros(cap, dtype=lp_local.dtype, device=lp_local.device)
    buf[:lp_local.shape[0]] = lp_local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = []
    for r in range(world):
        a, b = shard_bounds(n_total, world, r)
        out.append(parts[r][:b - a])
    return torch.cat(out)
