"""Thin Python owner of one ``glowk_handle`` (one per GPU / process).

PyTorch-ROCm is used for storage and streams only: tensors are ``torch.cuda`` float32 NHWC buffers whose
``data_ptr()`` is handed to the C ABI together with the current HIP stream; all arithmetic happens in
``libglowk.so``.  No CPU fallback: construction raises if the library is missing or no GPU is present.
"""
import ctypes
import os
import warnings

import numpy as np
import torch

from . import _lib
from .config import GlowConfig


def _stream_ptr(device):
    """The HIP stream torch currently uses on ``device`` (not on whatever device happens to be current)."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class GlowEngine:
    def __init__(self, cfg: GlowConfig, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("glowk needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.cfg = cfg
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        c = _lib.GlowkConfigStruct(cfg.H, cfg.W, cfg.C, cfg.L, cfg.K, cfg.F, int(cfg.learntop), int(cfg.use_logit),
                                   cfg.minval, cfg.maxval, cfg.alpha, cfg.bn_eps)
        h = ctypes.c_void_p()
        _lib.check(self.lib.glowk_create(ctypes.byref(c), self.device.index, ctypes.byref(h)))
        self.h = h
        self._finalized = False
        self._max_tiles_cap = None   # tests lower it to exercise the chunk loop on small batches
        self._fallbacks_seen = 0
        # the Python mirror never hands back silently wrong numbers and never fails on a checkpoint the split arithmetic cannot
        # hold: a call that leaves the fp16 range is re-run on the exact fp32 kernels inside the engine (one warning per engine)
        self.set_range_policy("fallback")

    def _stream(self):
        return _stream_ptr(self.device)

    def close(self):
        if getattr(self, "h", None):
            self.lib.glowk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------------------
    @staticmethod
    def _split_name(name):
        if name in _lib.PRIOR_TENSOR_IDS:
            return -1, 0, _lib.PRIOR_TENSOR_IDS[name]
        lvl, stp, rest = name.split("/", 2)
        return int(lvl[1:]), int(stp[1:]), _lib.STEP_TENSOR_IDS[rest]

    def set_tensor(self, name, value):
        level, step, tid = self._split_name(name)
        a = np.ascontiguousarray(np.asarray(value, dtype=np.float32)).ravel()
        _lib.check(self.lib.glowk_set_tensor(self.h, level, step, tid, a.ctypes.data_as(_lib._fp), a.size))
        self._finalized = False

    def get_tensor(self, name):
        level, step, tid = self._split_name(name)
        n = self.lib.glowk_tensor_size(self.h, level, tid)
        a = np.empty(n, dtype=np.float32)
        _lib.check(self.lib.glowk_get_tensor(self.h, level, step, tid, a.ctypes.data_as(_lib._fp), a.size))
        return a

    def load_params(self, params):
        """``params``: flat dict (see audiosourcesep_amd.synthetic) in the reference's tensor layouts."""
        for name, value in params.items():
            self.set_tensor(name, value)
        self.finalize()

    def finalize(self):
        _lib.check(self.lib.glowk_finalize_weights(self.h))
        self._finalized = True

    def actnorm_data_init(self, minibatch, runtime_order=False, raw_minibatch_quirk=True):
        """ActNorm data-dependent init on the GPU (flow_tfp_bijectors.py:222-234 driven as flow_glow.py:40-49)."""
        x = self._in(minibatch, self.data_shape)
        _lib.check(self.lib.glowk_actnorm_data_init(self.h, _ptr(x), x.shape[0], int(runtime_order), int(raw_minibatch_quirk),
                                                    self._stream()))

    def actnorm_params(self):
        """{name: ndarray} of every step's ActNorm tensors as the engine currently holds them."""
        out = {}
        for lvl in range(self.cfg.L):
            for k in range(self.cfg.K):
                for t in ("actnorm/log_scale", "actnorm/shift"):
                    name = "b%d/s%d/%s" % (lvl, k, t)
                    out[name] = self.get_tensor(name)
        return out

    def set_precision(self, mode):
        """_lib.PREC_F32 (exact), PREC_F16X3 or PREC_F16X2.  The split modes need |hidden activation| < 16 376; what happens
        beyond that is the range policy's business (default here: re-run the call on the fp32 kernels)."""
        _lib.check(self.lib.glowk_set_precision(self.h, int(mode)))

    def get_precision(self):
        return int(self.lib.glowk_get_precision(self.h))

    def set_range_policy(self, policy):
        """"fallback" (default of this class), "error" (raise GlowkRangeError; the C ABI's default) or "ignore" (no
        synchronisation: poll ``range_status``)."""
        modes = {"ignore": _lib.RANGE_IGNORE, "error": _lib.RANGE_ERROR, "fallback": _lib.RANGE_FALLBACK}
        _lib.check(self.lib.glowk_set_range_policy(self.h, modes[policy] if isinstance(policy, str) else int(policy)))

    def range_status(self, sync=True):
        """-> (tripped, fallbacks): whether the sticky range flag was set since the last look (clears it; waits for the stream)
        and how many calls were re-run on the fp32 kernels.  ``sync=False`` only reads the counter."""
        tripped, n = ctypes.c_int(0), ctypes.c_int64(0)
        _lib.check(self.lib.glowk_range_status(self.h, ctypes.byref(tripped) if sync else None, ctypes.byref(n), self._stream()))
        return bool(tripped.value), int(n.value)

    def range_probe_begin(self):
        """Start measuring the margin of the static range bound (``glowk_range_probe_begin``)."""
        _lib.check(self.lib.glowk_range_probe_begin(self.h))

    def range_probe_end(self):
        """-> (forward, backward): largest gathered network input / its limit over the split launches since ``range_probe_begin``
        (< 1: inside the bound; 0: no such launch)."""
        f, b = ctypes.c_float(0), ctypes.c_float(0)
        _lib.check(self.lib.glowk_range_probe_end(self.h, ctypes.byref(f), ctypes.byref(b), self._stream()))
        return float(f.value), float(b.value)

    def _after_call(self):
        if self.get_precision() == _lib.PREC_F32:
            return
        n = self.range_status(sync=False)[1]
        if n > self._fallbacks_seen:
            if self._fallbacks_seen == 0:
                warnings.warn("glowk: a hidden activation left the fp16 range of the split arithmetic; the call was re-run on the "
                              "exact fp32 kernels (slower). Consider set_precision(PREC_F32) for this checkpoint.", RuntimeWarning,
                              stacklevel=3)
            self._fallbacks_seen = n

    def reserve(self, n, with_grad=False):
        m = min(int(n), self.grad_max_tiles if with_grad else self.max_tiles)
        self._reserve_or_shrink(m, with_grad)

    def _reserve_or_shrink(self, m, with_grad):
        """glowk_reserve; when the device cannot hold the gradient path's footprint for ``m`` tiles (memory went to something
        else after the chunk was sized) the chunk is halved and tried again, so a later call chunks smaller instead of failing."""
        while True:
            try:
                _lib.check(self.lib.glowk_reserve(self.h, int(m), int(with_grad)))
                break
            except _lib.GlowkError as e:
                if not with_grad or m <= 1 or "out of memory" not in str(e).lower():
                    raise
                torch.cuda.empty_cache()
                m = max(1, m // 2)
                self._reserved_grad = 0          # (a failed glowk_reserve leaves the gradient buffers released)
                self._grad_cap = m               # ... and every later chunk is at most this large
        if with_grad:
            self._reserved_grad = max(getattr(self, "_reserved_grad", 0), int(m))

    @property
    def max_tiles(self):
        """Largest batch one C-ABI call takes (glowk_max_tiles); the methods below loop over chunks of it."""
        m = int(self.lib.glowk_max_tiles(self.h))
        return min(m, self._max_tiles_cap) if self._max_tiles_cap else m

    def _chunks(self, n, m=None):
        m = m or self.max_tiles
        return [(i, min(i + m, n)) for i in range(0, n, m)]

    @property
    def grad_max_tiles(self):
        """Chunk of log_prob_grad.  The pass keeps every step's coupling input, the pre-tanh log_s inputs of its coupling and the
        two ReLU masks of its network until the backward sweep -- for 64x64, K=32, F=512: ~7 MB per tile (until round 3 it kept
        the per-tap conv3 outputs as well: 15-23 MB).  The chunk is the largest batch whose workspace + saves (``glowk_workspace_bytes``,
        the allocators' own arithmetic) fit a byte budget: min(60 % of the free HBM, GLOWK_GRAD_BUDGET_GB or 64 GiB); kernels
        saturate long before that."""
        if not self._finalized:
            self.finalize()
        # the cached chunk is only as good as the memory picture it was sized on: what this engine has already reserved stays
        # its own, anything else that was allocated since (a second prior's engine, training state, pinned staging) shrinks the
        # budget -- so the key carries the free memory in 1-GiB steps plus this engine's own reservation
        free = self._free_bytes()
        own = self.workspace_bytes(self._reserved_grad, True) if getattr(self, "_reserved_grad", 0) else 0
        key = (self.get_precision(), self._max_tiles_cap, (free + own) >> 30)
        key = key + (getattr(self, "_grad_cap", None),)
        if getattr(self, "_grad_chunk", (None, 0))[0] == key:
            return self._grad_chunk[1]
        budget = min(0.6 * (free + own), float(os.environ.get("GLOWK_GRAD_BUDGET_GB", "64")) * 2 ** 30)
        lo, hi = 1, min(self.max_tiles, getattr(self, "_grad_cap", None) or self.max_tiles)
        if self.workspace_bytes(hi, True) > budget:
            while lo < hi:      # largest n with bytes(n) <= budget (bytes is monotone in n)
                mid = (lo + hi + 1) // 2
                if self.workspace_bytes(mid, True) <= budget:
                    lo = mid
                else:
                    hi = mid - 1
        else:
            lo = hi
        # never below a batch this handle has already run through the gradient path: its buffers exist (the training buffers of
        # glowk_param_grad are not part of `own`, so free memory -- and with it the budget -- drops after the first training step;
        # round-3 advisor: a batch that ran on step 1 must not be refused on step 2)
        lo = max(lo, min(getattr(self, "_ran_grad", 0), self.max_tiles))
        self._grad_chunk = (key, lo)
        return lo

    def _free_bytes(self):
        return torch.cuda.mem_get_info(self.device)[0]

    def workspace_bytes(self, n, with_grad=False):
        return int(self.lib.glowk_workspace_bytes(self.h, int(n), int(with_grad)))

    @property
    def fused_steps(self):
        """Flow steps so far that ran as one fused network + coupling kernel (``glowk_fused_steps``)."""
        return int(self.lib.glowk_fused_steps(self.h))

    def kernel_families(self):
        """Coupling-network launches so far by kernel family: dict f32 / h3_32x32x16 / h3s_16x16x32 / h3s_half / fused, and
        "co_resident": how many of the h3s_16x16x32 / fused launches took the four-wave, two-workgroups-per-CU form; "small_grid_q": how
        many of the h3s_half launches took the form with all conv1 blocks first (k_net_h3q)."""
        out = (ctypes.c_int64 * 7)()
        _lib.check(self.lib.glowk_kernel_families(self.h, out))
        return dict(zip(("f32", "h3_32x32x16", "h3s_16x16x32", "h3s_half", "fused", "co_resident", "small_grid_q"), [int(v) for v in out]))

    def profile_begin(self):
        _lib.check(self.lib.glowk_profile_begin(self.h))

    def profile_end(self):
        """-> [(summed k_net milliseconds, launches)] per level, from HIP events on the launch stream."""
        p = _lib.GlowkProfile()
        _lib.check(self.lib.glowk_profile_end(self.h, ctypes.byref(p)))
        return [(p.net_ms[i], int(p.net_launches[i])) for i in range(self.cfg.L)]

    # ---- helpers ----------------------------------------------------------------------------------
    def _in(self, x, shape_tail):
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x, dtype=np.float32))
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(x.shape[1:]) != tuple(shape_tail):
            raise ValueError("expected a [N, %s] tensor, got %s" % (", ".join(map(str, shape_tail)), tuple(x.shape)))
        if not self._finalized:
            self.finalize()
        return x

    def _new(self, *shape):
        return torch.empty(shape, device=self.device, dtype=torch.float32)

    @property
    def data_shape(self):
        return (self.cfg.H, self.cfg.W, self.cfg.C)

    # ---- hot path ---------------------------------------------------------------------------------
    def _compute(self, rc):
        _lib.check(rc)
        self._after_call()

    def forward(self, x, with_logdet=True):
        x = self._in(x, self.data_shape)
        n = x.shape[0]
        z = self._new(n, *self.cfg.latent_shape())
        ld = self._new(n) if with_logdet else None
        if n == 0:   # an empty batch is an empty result (TF semantics), not a launch
            return (z, ld) if with_logdet else z
        for a, b in self._chunks(n):   # tiles are independent: a batch beyond max_tiles is a loop over chunks
            self._compute(self.lib.glowk_forward(self.h, _ptr(x[a:b]), b - a, _ptr(z[a:b]), _ptr(ld[a:b] if with_logdet else None),
                                              self._stream()))
        return (z, ld) if with_logdet else z

    def inverse(self, z):
        z = self._in(z, self.cfg.latent_shape())
        n = z.shape[0]
        x = self._new(n, *self.data_shape)
        if n == 0:
            return x
        for a, b in self._chunks(n):
            self._compute(self.lib.glowk_inverse(self.h, _ptr(z[a:b]), b - a, _ptr(x[a:b]), self._stream()))
        return x

    def log_prob(self, x, return_latent=False, out=None):
        x = self._in(x, self.data_shape)
        n = x.shape[0]
        lp = out if out is not None else self._new(n)
        z = self._new(n, *self.cfg.latent_shape()) if return_latent else None
        if n == 0:
            return (lp, z) if return_latent else lp
        for a, b in self._chunks(n):
            self._compute(self.lib.glowk_log_prob(self.h, _ptr(x[a:b]), b - a, _ptr(lp[a:b]), _ptr(z[a:b] if return_latent else None),
                                               self._stream()))
        return (lp, z) if return_latent else lp

    def log_prob_sum(self, x, out=None, total=None):
        """-> (log_prob [N], total [1] float64 on the device = sum_n log_prob[n]): ``glowk_log_prob_sum``.  The sum is taken by the
        engine in a fixed order (chunks accumulate in order), so nothing but the engine's kernels runs between the tiles and the
        one-element all-reduce of train_glow.py:52-54."""
        x = self._in(x, self.data_shape)
        n = x.shape[0]
        lp = out if out is not None else self._new(n)
        if total is None:
            total = torch.empty(1, device=self.device, dtype=torch.float64)
        if n == 0:
            _lib.check(self.lib.glowk_sum_f64(_ptr(None), 0, _ptr(total), 0, 1.0, self._stream()))
            return lp, total
        for i, (a, b) in enumerate(self._chunks(n)):
            self._compute(self.lib.glowk_log_prob_sum(self.h, _ptr(x[a:b]), b - a, _ptr(lp[a:b]), _ptr(None), _ptr(total), int(i > 0),
                                                   self._stream()))
        return lp, total

    def sum_f64(self, v, scale=1.0, out=None):
        """scale * sum(v) as one fp64 device value, fixed summation order (``glowk_sum_f64``)."""
        v = v.contiguous()
        if out is None:
            out = torch.empty(1, device=v.device, dtype=torch.float64)
        _lib.check(self.lib.glowk_sum_f64(_ptr(v), v.numel(), _ptr(out), 0, float(scale), _stream_ptr(v.device)))
        return out

    def log_prob_grad(self, x):
        x = self._in(x, self.data_shape)
        n = x.shape[0]
        lp, dx = self._new(n), torch.empty_like(x)
        if n == 0:
            return lp, dx
        chunk = self.grad_max_tiles
        if min(n, chunk) > getattr(self, "_reserved_grad", 0):
            self._reserve_or_shrink(min(n, chunk), True)      # (may shrink the chunk when the memory is no longer there)
            chunk = min(chunk, self.grad_max_tiles)
        for a, b in self._chunks(n, chunk):
            self._compute(self.lib.glowk_log_prob_grad(self.h, _ptr(x[a:b]), b - a, _ptr(lp[a:b]), _ptr(dx[a:b]), self._stream()))
        self._ran_grad = max(getattr(self, "_ran_grad", 0), min(n, chunk))
        return lp, dx

    # ---- training step (train_glow.py:29-44) ------------------------------------------------------------
    @property
    def param_vector_size(self):
        return int(self.lib.glowk_param_vector_size(self.h))

    def param_slice(self, name):
        """(offset, count) of a named tensor in the flat parameter / gradient vector (glowk_param_offset)."""
        level, step, tid = self._split_name(name)
        off, cnt = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _lib.check(self.lib.glowk_param_offset(self.h, level, step, tid, ctypes.byref(off), ctypes.byref(cnt)))
        return int(off.value), int(cnt.value)

    def param_grad(self, x, scale, grad=None):
        """-> (log_prob [N], grad [param_vector_size]) with grad = scale * d sum_n log_prob(x_n) / d theta, in the handle's
        arithmetic (the split sweep with hidden stores in f16x3 / f16x2 where every level has training instances, else -- and in
        f32 -- the exact kernels); under the range policy like every compute call (a fallback is counted and warned about)."""
        x = self._in(x, self.data_shape)
        n = x.shape[0]
        if n > self.grad_max_tiles:
            raise ValueError("param_grad takes at most %d tiles per call (split the batch and add the gradients)" % self.grad_max_tiles)
        lp = self._new(n)
        if grad is None:
            grad = self._new(self.param_vector_size)
        self._compute(self.lib.glowk_param_grad(self.h, _ptr(x), n, float(scale), _ptr(lp), _ptr(grad), self._stream()))
        if n > getattr(self, "_ran_grad", 0):
            self._ran_grad = n
            self._grad_chunk = (None, 0)      # (new training buffers changed the memory picture: size the chunk again)
        return lp, grad

    def apply_gradients(self, grad, optimizer="adamax", lr=1e-3):
        opt = {"adam": 0, "adamax": 1}
        if optimizer not in opt:
            raise ValueError("optimizer argument should be adam or adamax")      # train_utils.py:40
        _lib.check(self.lib.glowk_apply_gradients(self.h, _ptr(grad), opt[optimizer], float(lr), self._stream()))
        self._finalized = True

    def sample_from_eps(self, eps):
        eps = self._in(eps, self.cfg.latent_shape())
        n = eps.shape[0]
        x = self._new(n, *self.data_shape)
        if n == 0:
            return x
        for a, b in self._chunks(n):
            self._compute(self.lib.glowk_sample(self.h, _ptr(eps[a:b]), b - a, _ptr(x[a:b]), self._stream()))
        return x

    def prior_log_prob(self, z):
        z = self._in(z, self.cfg.latent_shape())
        lp = self._new(z.shape[0])
        for a, b in self._chunks(z.shape[0]):
            _lib.check(self.lib.glowk_prior_log_prob(self.h, _ptr(z[a:b]), b - a, _ptr(lp[a:b]), self._stream()))
        return lp

    # ---- sub-bijectors ----------------------------------------------------------------------------
    def preprocess_forward(self, x):
        x = self._in(x, self.data_shape)
        y, ld = torch.empty_like(x), self._new(x.shape[0])
        _lib.check(self.lib.glowk_preprocess_forward(self.h, _ptr(x), x.shape[0], _ptr(y), _ptr(ld), self._stream()))
        return y, ld

    def preprocess_inverse(self, y):
        y = self._in(y, self.data_shape)
        x = torch.empty_like(y)
        _lib.check(self.lib.glowk_preprocess_inverse(self.h, _ptr(y), y.shape[0], _ptr(x), self._stream()))
        return x

    def step_forward(self, level, step, u):
        u = self._in(u, self.cfg.level_shapes()[level])
        y, ld = torch.empty_like(u), self._new(u.shape[0])
        self._compute(self.lib.glowk_step_forward(self.h, level, step, _ptr(u), u.shape[0], _ptr(y), _ptr(ld), self._stream()))
        return y, ld

    def step_inverse(self, level, step, y):
        y = self._in(y, self.cfg.level_shapes()[level])
        u = torch.empty_like(y)
        self._compute(self.lib.glowk_step_inverse(self.h, level, step, _ptr(y), y.shape[0], _ptr(u), self._stream()))
        return u

    def coupling_net(self, level, step, xb):
        h, w, c = self.cfg.level_shapes()[level]
        xb = self._in(xb, (h, w, c // 2))
        log_s, t = torch.empty_like(xb), torch.empty_like(xb)
        self._compute(self.lib.glowk_coupling_net(self.h, level, step, _ptr(xb), xb.shape[0], _ptr(log_s), _ptr(t), self._stream()))
        return log_s, t


def squeeze(x):
    """Squeeze._forward (flow_tfp_bijectors.py:170-174) on the GPU: [N,H,W,C] -> [N,H/2,W/2,4C]."""
    lib = _lib.load()
    x = x.contiguous()
    n, H, W, C = x.shape
    y = torch.empty((n, H // 2, W // 2, 4 * C), device=x.device, dtype=torch.float32)
    _lib.check(lib.glowk_squeeze(_ptr(x), n, H, W, C, _ptr(y), _stream_ptr(x.device)))
    return y


def unsqueeze(y):
    """Squeeze._inverse (flow_tfp_bijectors.py:176-180)."""
    lib = _lib.load()
    y = y.contiguous()
    n, h, w, c4 = y.shape
    x = torch.empty((n, 2 * h, 2 * w, c4 // 4), device=y.device, dtype=torch.float32)
    _lib.check(lib.glowk_unsqueeze(_ptr(y), n, h, w, c4, _ptr(x), _stream_ptr(y.device)))
    return x
