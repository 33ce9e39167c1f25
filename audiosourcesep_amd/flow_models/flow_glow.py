"""Host-side mirror of the objects ``build_glow`` hands back in the reference.

``GlowFlow``            <-> ``tfd.TransformedDistribution(prior, tfb.Invert(tfb.Chain([glow, prepro])))``
                             (flow_builder.py:127-144): ``log_prob``, ``sample``, ``bijector``, ``variables``.
``ChainBijector``       <-> ``tfb.Chain([GlowBijector_{2,3,4}blocks, SpecPreprocessing])`` (data -> latent).
``InvertedBijector``    <-> ``tfb.Invert(chain)``: what ``flow.bijector`` is in the reference (latent -> data).
``GlowStepView``        <-> one ``GlowStep`` (flow_glow.py:9-31), addressed as ``chain.glow.blocks[l].steps[k]``.

Nothing here computes: every method forwards to the C ABI through :class:`audiosourcesep_amd.engine.GlowEngine`.
"""
import numpy as np
import torch

from .. import _lib


class Variable:
    """Named view of one engine tensor (stands in for ``tf.Variable``: ``.name``, ``.numpy()``, ``.assign()``)."""

    def __init__(self, engine, name, shape, trainable):
        self._engine, self.name, self.shape, self.trainable = engine, name, tuple(shape), trainable

    def numpy(self):
        return self._engine.get_tensor(self.name).reshape(self.shape)

    def assign(self, value):
        value = np.asarray(value, dtype=np.float32)
        if value.shape != self.shape:
            raise ValueError("%s: expected shape %s, got %s" % (self.name, self.shape, value.shape))
        self._engine.set_tensor(self.name, value)   # re-packed lazily before the next compute call
        return self

    def __repr__(self):
        return "<Variable %s shape=%s trainable=%s>" % (self.name, self.shape, self.trainable)


def _step_variable_shapes(c, F):
    """Tensors of one GlowStep in CREATION order (GlowStep.__init__, flow_glow.py:15-20): name, shape,
    trainable.  This is NOT the order of ``flow.variables`` -- that one follows ``tf.Module``'s attribute traversal and lives in
    ``tf_checkpoint.variable_order`` (the single source of truth for it); creation order is kept as the named alternative
    ``GlowFlow.variables_in_creation_order``."""
    ci = c // 2
    return [
        ("actnorm/log_scale", (c,), True), ("actnorm/shift", (c,), True),
        ("inv1x1/P", (c, c), False), ("inv1x1/P_inv", (c, c), False), ("inv1x1/sign_S", (c,), False),   # flow_tfp_bijectors.py:281-286
        ("inv1x1/L", (c, c), True), ("inv1x1/log_S", (c,), True), ("inv1x1/U", (c, c), True),
        ("nn/conv1/kernel", (3, 3, ci, F), True), ("nn/conv1/bias", (F,), True),
        ("nn/bn1/gamma", (F,), True), ("nn/bn1/beta", (F,), True), ("nn/bn1/mean", (F,), False), ("nn/bn1/var", (F,), False),
        ("nn/conv2/kernel", (1, 1, F, F), True), ("nn/conv2/bias", (F,), True),
        ("nn/bn2/gamma", (F,), True), ("nn/bn2/beta", (F,), True), ("nn/bn2/mean", (F,), False), ("nn/bn2/var", (F,), False),
        ("nn/conv3/kernel", (3, 3, F, c), True), ("nn/conv3/bias", (c,), True),
    ]


def _shard_offset(n_local, device, group=None):
    """Tiles held by the ranks below this one (exclusive scan of the local batch sizes over ``group``); 0 without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return 0
    on_dev = device.type == "cuda" and dist.get_backend(group) != "gloo"
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device if on_dev else "cpu")
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    return int(sum(int(t.item()) for t in sizes[:rank]))


class _LogProbFn(torch.autograd.Function):
    """``log_prob`` with reverse mode wrt the input: what ``tf.GradientTape`` gives compute_grad_logprob
    (run_basis_sep.py:73-79).  Forward and input gradient both come from ``glowk_log_prob_grad``."""

    @staticmethod
    def forward(ctx, x, engine):
        lp, dx = engine.log_prob_grad(x.detach())
        ctx.save_for_backward(dx)
        return lp

    @staticmethod
    def backward(ctx, grad_out):
        (dx,) = ctx.saved_tensors
        return dx * grad_out.reshape(-1, 1, 1, 1), None


class CouplingNetView:
    """``AffineCouplingLayerSplit.shift_and_log_scale_fn`` of one step (flow_tfp_bijectors.py:132)."""

    def __init__(self, engine, level, step):
        self._e, self._l, self._k = engine, level, step

    def __call__(self, xb):
        return self._e.coupling_net(self._l, self._k, xb)


class GlowStepView:
    """One GlowStep: ActNorm -> Invertible1x1Conv -> AffineCouplingLayerSplit (flow_glow.py:21-22)."""

    def __init__(self, engine, level, step):
        self._e, self.level, self.step = engine, level, step
        self.shift_and_log_scale_fn = CouplingNetView(engine, level, step)

    def forward(self, x):
        return self._e.step_forward(self.level, self.step, x)[0]

    def inverse(self, y):
        return self._e.step_inverse(self.level, self.step, y)

    def forward_log_det_jacobian(self, x, event_ndims=3):
        return self._e.step_forward(self.level, self.step, x)[1]

    def inverse_log_det_jacobian(self, y, event_ndims=3):
        return -self._e.step_forward(self.level, self.step, self.inverse(y))[1]


class GlowBlockView:
    """One GlowBlock (flow_glow.py:34-77): squeeze + K steps; ``steps[k]`` is ``glowStep_k``."""

    def __init__(self, engine, level):
        self.level = level
        self.steps = [GlowStepView(engine, level, k) for k in range(engine.cfg.K)]
        h, w, c = engine.cfg.level_shapes()[level]
        self.event_shape_out = (h, w, c)


class GlowView:
    def __init__(self, engine):
        self.blocks = [GlowBlockView(engine, l) for l in range(engine.cfg.L)]


class ChainBijector:
    """data -> latent: ``tfb.Chain([glow, prepro])`` (flow_builder.py:127)."""

    def __init__(self, engine):
        self._e = engine
        self.glow = GlowView(engine)

    def forward(self, x):
        return self._e.forward(x, with_logdet=False)

    def inverse(self, z):
        return self._e.inverse(z)

    def forward_log_det_jacobian(self, x, event_ndims=3):
        return self._e.forward(x)[1]

    def inverse_log_det_jacobian(self, z, event_ndims=3):
        return -self._e.forward(self._e.inverse(z))[1]

    def forward_event_shape(self, input_shape):
        H, W, C = input_shape
        s = 2 ** self._e.cfg.L
        return (H // s, W // s, C * s * s)       # flow_glow.py:128-134,211-217,315-321

    def inverse_event_shape(self, output_shape):
        H, W, C = output_shape
        s = 2 ** self._e.cfg.L
        return (H * s, W * s, C // (s * s))


class InvertedBijector:
    """``tfb.Invert(chain)`` (flow_builder.py:129): forward is latent -> data."""

    def __init__(self, chain):
        self.bijector = chain

    def forward(self, z):
        return self.bijector.inverse(z)

    def inverse(self, x):
        return self.bijector.forward(x)

    def forward_log_det_jacobian(self, z, event_ndims=3):
        return self.bijector.inverse_log_det_jacobian(z, event_ndims)

    def inverse_log_det_jacobian(self, x, event_ndims=3):
        return self.bijector.forward_log_det_jacobian(x, event_ndims)

    def forward_event_shape(self, s):
        return self.bijector.inverse_event_shape(s)

    def inverse_event_shape(self, s):
        return self.bijector.forward_event_shape(s)


class GlowFlow:
    """The distribution object the reference's scripts hold (``flow``)."""

    def __init__(self, engine):
        self.engine = engine
        self.cfg = engine.cfg
        self.chain = ChainBijector(engine)
        self.bijector = InvertedBijector(self.chain)
        self.event_shape = engine.data_shape
        cfg = self.cfg
        by_name = {}
        creation = []
        for lvl, (h, w, c) in enumerate(cfg.level_shapes()):
            for k in range(cfg.K):
                for name, shape, trainable in _step_variable_shapes(c, cfg.F):
                    v = Variable(engine, "b%d/s%d/%s" % (lvl, k, name), shape, trainable)
                    by_name[v.name] = v
                    creation.append(v)
        if cfg.learntop:
            for name in ("prior/loc", "prior/log_scale"):
                by_name[name] = Variable(engine, name, cfg.latent_shape(), True)
                creation.append(by_name[name])
        # ONE source of truth for the position of a variable in ``flow.variables`` -- the checkpoint root of train_utils.py:67-68:
        # tf_checkpoint.variable_order (tf.Module's attribute traversal, derived from the reference's source)
        from ..tf_checkpoint import variable_order
        order = variable_order(cfg)
        assert sorted(order) == sorted(by_name), "variable_order and the step tensors disagree"
        self._variables = tuple(by_name[n] for n in order)
        self._creation_order = tuple(creation)
        self._noise_step = 0      # train_step(noise_std > 0) calls so far: the device RNG's step counter when the caller gives none

    # --- tfd.Distribution surface ------------------------------------------------------------------
    def log_prob(self, x):
        """[N,H,W,C] -> [N] (train_glow.py:30; run_basis_sep.py:77).  If ``x.requires_grad`` the result carries the
        input gradient (BASIS contract)."""
        if torch.is_tensor(x) and x.requires_grad:
            return _LogProbFn.apply(x, self.engine)
        return self.engine.log_prob(x)

    def sample(self, n, seed=None):
        """n -> [n,H,W,C] (train_glow.py:74): the prior's standard-normal draw comes from the engine's own device RNG
        (``glowk_random``, Philox stream ``seed``; unseeded calls take a fresh stream per call like ``tf.random``), the prior's
        affine and ``chain.inverse`` run in the engine (``glowk_sample``)."""
        from ..basis import device_randn
        if seed is None:
            seed = int.from_bytes(__import__("os").urandom(7), "little")
        eps = device_randn((int(n),) + tuple(self.cfg.latent_shape()), self.engine.device, int(seed), step=0, which=3)
        return self.engine.sample_from_eps(eps)

    def train_step(self, x, optimizer="adamax", lr=1e-3, global_batch_size=None, group=None, noise_std=0.0, seed=0, step=None,
                   tile_offset=None):
        """One step of train_glow.py's ``distributed_train_step`` (:37-54) on this rank's tiles ``x``: loss =
        sum(-log_prob(x)) / global_batch_size, gradients wrt ``trainable_variables``, summed over the ranks of ``group``
        (one all-reduce of the flat gradient vector), ``optimizer.apply_gradients``.  ``noise_std`` > 0 adds N(0, noise_std^2) to
        the tiles first (train_noisy_glow.py:31: the noise-conditioned priors of BASIS; fresh noise every step and on every
        replica), drawn by the engine's device RNG: stream ``seed``, step = ``step`` or, when None, this flow's own count of
        noisy steps (so two consecutive calls never reuse a draw), element offset = ``tile_offset`` tiles -- default: this rank's
        position in the global batch = the number of tiles the lower ranks hold (an all-gather of the local batch sizes: shards
        may be uneven, ``distributed.shard_bounds`` gives the first ranks one tile more), so no two ranks draw the same noise and
        a tile's draw does not depend on the world size.  Returns the global loss (fp64 scalar tensor)."""
        from ..distributed import distributed_train_step
        x = self.engine._in(x, self.engine.data_shape)
        if noise_std:
            import torch.distributed as dist
            from ..basis import add_device_noise
            if step is None:
                step = self._noise_step
            self._noise_step = int(step) + 1
            if tile_offset is None:
                tile_offset = _shard_offset(x.shape[0], x.device, group)
            elems = int(np.prod(self.engine.data_shape))
            x = add_device_noise(x, noise_std, seed, step, which=2, offset=int(tile_offset) * elems)
        gb = int(global_batch_size) if global_batch_size else x.shape[0]
        return distributed_train_step(self.engine.param_grad, lambda g: self.engine.apply_gradients(g, optimizer, lr), x, gb, group=group)

    def set_precision(self, precision):
        """``"f32"`` (exact fp32 MFMA) or ``"f16x3"`` (error-compensated fp16 split) for every later call."""
        modes = {"f32": _lib.PREC_F32, "f16x3": _lib.PREC_F16X3, "f16x2": _lib.PREC_F16X2}
        if precision not in modes:
            raise ValueError("precision must be 'f32' or 'f16x3'")
        self.engine.set_precision(modes[precision])
        return self

    @property
    def variables(self):
        """In the order of the reference's ``flow.variables`` (``tf_checkpoint.variable_order``: the derived ``tf.Module``
        traversal -- variable i is checkpoint key ``variables/<i>``)."""
        return self._variables

    @property
    def variables_in_creation_order(self):
        """The same variables in constructor creation order (flow_glow.py:15-20), the named alternative."""
        return self._creation_order

    @property
    def trainable_variables(self):
        return tuple(v for v in self._variables if v.trainable)

    # --- checkpoint: own container (TF-checkpoint import is section 8(f-2)) ------------
    def state_dict(self):
        return {v.name: v.numpy() for v in self._variables}

    def load_state_dict(self, state, strict=True):
        """``strict``: every variable must be present and no unknown key may be (like ``tf.train.Checkpoint.restore(...)
        .assert_consumed()``); the frozen ``inv1x1/P_inv`` alone may be absent -- the engine then uses inv(P), the value the
        reference initialises it to (flow_tfp_bijectors.py:282-284)."""
        names = {v.name for v in self._variables}
        if strict:
            missing = sorted(n for n in names if n not in state and not n.endswith("inv1x1/P_inv"))
            unexpected = sorted(k for k in state if k not in names)
            if missing or unexpected:
                raise KeyError("load_state_dict: missing %s; unexpected %s" % (missing[:5] + (["..."] if len(missing) > 5 else []),
                                                                           unexpected[:5] + (["..."] if len(unexpected) > 5 else [])))
        for v in self._variables:
            if v.name in state:
                v.assign(np.asarray(state[v.name]).reshape(v.shape))
        self.engine.finalize()

    def save(self, path):
        np.savez(path, **self.state_dict())

    def restore(self, path, order=None):
        """``path``: an ``.npz`` written by ``save`` -- or the prefix of a TensorFlow checkpoint (``.../tf_ckpts/ckpt-21``, i.e. what
        ``CheckpointManager`` of train_utils.py:62-75 wrote and run_basis_sep.py:28-38 restores): its tensor bundle is read without
        TensorFlow and mapped through the derived ``flow.variables`` order (audiosourcesep_amd/tf_checkpoint.py; ``order``
        overrides it)."""
        import os
        if os.path.exists(path + ".index"):
            from ..tf_checkpoint import state_dict_from_checkpoint
            self.load_state_dict(state_dict_from_checkpoint(path, self.cfg, order))
            return
        with np.load(path) as f:
            self.load_state_dict({k: f[k] for k in f.files})

    def save_tf(self, prefix, order=None):
        """Write the variables as a TensorFlow tensor bundle under the reference's checkpoint keys (``variables/<i>/...``)."""
        from ..tf_checkpoint import save_checkpoint_bundle
        save_checkpoint_bundle(prefix, self.state_dict(), self.cfg, order)
