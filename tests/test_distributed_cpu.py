"""World-size-2 (and 3) gloo runs of the sharding logic on CPU: the N>1 path of bench.py / BASIS without a GPU.
The per-shard evaluator is the CPU oracle (test infrastructure), so this checks the collective plumbing only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.distributed import shard_bounds, sharded_log_prob, gather_log_prob, distributed_test_step
from audiosourcesep_amd.synthetic import synthetic_params, synthetic_mel_tiles

CFG = GlowConfig(H=8, W=8, C=1, L=2, K=2, F=128)
N_TILES = 7   # ragged over 2 and 3 ranks


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_log_prob(x):
    from oracle import glowref as R
    p = R.cast_params(synthetic_params(CFG), np.float64)
    return torch.from_numpy(R.log_prob(x.numpy().astype(np.float64), p, CFG.as_dict()).astype(np.float32))


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.from_numpy(synthetic_mel_tiles(N_TILES, CFG))
    a, b = shard_bounds(N_TILES, world, rank)
    lp, total = sharded_log_prob(_oracle_log_prob, x[a:b])
    full = gather_log_prob(lp, N_TILES)
    loss = distributed_test_step(_oracle_log_prob, x[a:b], N_TILES)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.concatenate([[total.item(), loss.item()], full.numpy().astype(np.float64)]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 30, 1024):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sum_equals_single_process(world, tmp_path):
    x = torch.from_numpy(synthetic_mel_tiles(N_TILES, CFG))
    ref = _oracle_log_prob(x).numpy().astype(np.float64)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / ("r%d.npy" % r))
        np.testing.assert_allclose(got[0], ref.sum(), rtol=1e-6)
        np.testing.assert_allclose(got[1], -ref.mean(), rtol=1e-6)    # train_glow.py:33-35: the global batch's mean NLL
        np.testing.assert_array_equal(got[2:], ref)                   # gathered back in batch order


def test_test_step_without_process_group():
    x = torch.from_numpy(synthetic_mel_tiles(3, CFG))
    ref = _oracle_log_prob(x).numpy().astype(np.float64)
    np.testing.assert_allclose(distributed_test_step(_oracle_log_prob, x, 3).item(), -ref.mean(), rtol=1e-6)
    with pytest.raises(ValueError):
        distributed_test_step(_oracle_log_prob, x, 0)


def _train_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from audiosourcesep_amd.distributed import distributed_train_step, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    x = torch.randn(11, 6, dtype=torch.float32)                   # 11 "tiles": ragged over 2 and 3 ranks
    theta = torch.linspace(-1, 1, 6)

    def param_grad(xl, scale):   # stand-in for GlowEngine.param_grad: log_prob_n = -0.5 |x_n - theta|^2
        d = xl - theta
        return -0.5 * (d * d).sum(1), scale * d.sum(0)

    applied = {}
    a, b = shard_bounds(11, world, rank)
    loss = distributed_train_step(param_grad, lambda g: applied.setdefault("g", g.clone()), x[a:b], 11)
    lp_all, g_all = param_grad(x, -1.0 / 11)
    q.put((rank, float(loss), float(-lp_all.sum() / 11), applied["g"].tolist(), g_all.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gradients_sum_to_the_single_process_ones(world):
    """train_glow.py:37-54 over batch shards: per-rank gradients of the pre-scaled loss, all-reduced, equal the gradient of the
    whole batch in one process; every rank applies the same vector and sees the same global loss."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, loss, loss_ref, g, g_ref in res:
        assert abs(loss - loss_ref) < 1e-6 * abs(loss_ref)
        np.testing.assert_allclose(g, g_ref, rtol=1e-5, atol=1e-6)
    assert len({tuple(r[3]) for r in res}) == 1          # bit-identical on every rank


# ---- prior-parallel BASIS (round-3 verdict, item 4): the two priors of a tile shard on two ranks, one all-gather of their gradients per
# Langevin step, the identical update on both (run_basis_sep.py:171-181) --------------------------------------------------------------
class _StandInPrior:
    """A prior with a closed-form gradient (diagonal Gaussian in dB space): what compute_grad_logprob needs of a GlowFlow."""

    def __init__(self, mu, s):
        self.mu, self.s = float(mu), float(s)
        self.engine = self

    def log_prob_grad(self, x):
        d = (x - self.mu) / self.s
        return -0.5 * (d * d).flatten(1).sum(1), -d / self.s


def _basis_noise(sig, t, which, shape, lo=0, n=9):
    """Injected Langevin noise, a pure function of (level, step, source, tile index): every rank draws the whole batch's noise and takes
    its tiles, as the device RNG does with its tile offset."""
    g = torch.Generator().manual_seed(1000 * sig + 10 * t + which)
    return torch.randn((n,) + tuple(shape[1:]), generator=g)[lo:lo + shape[0]]


def _basis_problem():
    g = torch.Generator().manual_seed(5)
    s1 = -40.0 + 8.0 * torch.randn(9, 6, 4, 1, generator=g)
    s2 = -50.0 + 8.0 * torch.randn(9, 6, 4, 1, generator=g)
    from audiosourcesep_amd import basis
    mixed = basis.mixing_db(s1, s2)
    x1 = -100.0 + 120.0 * torch.rand(9, 6, 4, 1, generator=g)
    x2 = -100.0 + 120.0 * torch.rand(9, 6, 4, 1, generator=g)
    return mixed, x1, x2, _StandInPrior(-40.0, 8.0), _StandInPrior(-50.0, 8.0), basis.get_sigmas(30.0, 1.2, 3)


def _basis_pp_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from audiosourcesep_amd import basis
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mixed, x1, x2, p1, p2, sigmas = _basis_problem()
    lay = basis.prior_parallel_layout(mixed.shape[0], world, rank)
    pair = basis.make_pair_group(world, rank)
    a, b = lay["bounds"]
    # this rank holds ONE prior: the other model is None and must never be asked
    m1, m2 = (p1, None) if lay["prior"] == 0 else (None, p2)
    y1, y2, arr = basis.basis_outer_loop(mixed[a:b], x1[a:b], x2[a:b], m1, m2, sigmas, T=6, delta=0.05,
                                         noise_fn=lambda s, t, w, shape: _basis_noise(s, t, w, shape, lo=a), tile_offset=a,
                                         prior_group=pair, prior_index=lay["prior"])
    q.put((rank, lay["prior"], a, b, y1.numpy(), y2.numpy(), len(arr["x1"])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_prior_parallel_basis_equals_the_one_process_chain(world):
    """world = 2 S ranks: rank r owns prior r % 2 of tile shard r // 2.  Every rank ends with the state the one-process chain
    (both priors, all tiles) reaches for its tiles -- bit for bit: the gradients are gathered, not reduced, and the update is the
    same arithmetic on the same numbers -- and the two ranks of a pair hold identical copies."""
    from audiosourcesep_amd import basis
    mixed, x1, x2, p1, p2, sigmas = _basis_problem()
    r1, r2, _ = basis.basis_outer_loop(mixed, x1, x2, p1, p2, sigmas, T=6, delta=0.05, noise_fn=lambda s, t, w, shape: _basis_noise(s, t, w, shape))
    assert torch.isfinite(r1).all() and float((r1 - x1).abs().mean()) > 1.0          # the chain moved
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_basis_pp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = []
    for rank, prior, a, b, y1, y2, nlev in res:
        assert prior == rank % 2 and nlev == len(sigmas) + 1
        np.testing.assert_array_equal(y1, r1[a:b].numpy())
        np.testing.assert_array_equal(y2, r2[a:b].numpy())
        if prior == 0:
            covered.append((a, b))
    assert covered[0][0] == 0 and covered[-1][1] == mixed.shape[0] and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
    with pytest.raises(ValueError):
        basis.prior_parallel_layout(30, 3, 0)


def _offset_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from audiosourcesep_amd.flow_models.flow_glow import _shard_offset
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard_bounds(11, world, rank)
    q.put((rank, a, _shard_offset(b - a, torch.device("cpu"))))
    dist.destroy_process_group()


def test_default_noise_offset_of_uneven_shards():
    """Round-3 advisor: GlowFlow.train_step's default tile offset was rank * local batch -- with shard_bounds-style shards (the
    first ranks hold one tile more) neighbouring ranks drew overlapping noise.  It is now the exclusive scan of the batch sizes."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_offset_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, a, off in res:
        assert off == a, (rank, a, off)          # 11 tiles over 3 ranks: 4 + 4 + 3 -> offsets 0, 4, 8 (rank * local would say 0, 4, 6)
