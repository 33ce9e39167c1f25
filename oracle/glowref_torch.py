"""CPU ORACLE (test infrastructure only) -- second, independent restatement on torch-CPU.

Same op graph as ``oracle/glowref.py`` (see its header for scope, citations and pinning status) but
built on ``torch.nn.functional.conv2d`` so that

* ``tests/`` can cross-check the NumPy restatement against a different convolution implementation,
* reverse-mode autodiff gives the oracle for ``compute_grad_logprob`` (run_basis_sep.py:73-79),
* ``bench.py``'s ``cpu_baseline`` leg has a CPU path whose speed is representative of a framework
  CPU backend (oneDNN convolutions, all host cores), as BASELINE.md section 3 asks.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.
Never runs on the GPU: every tensor is forced to ``device='cpu'``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as Fnn


def to_torch(params, dtype=torch.float64):
    return {k: torch.as_tensor(np.asarray(v), dtype=dtype, device="cpu") for k, v in params.items()}


def squeeze(x):
    """flow_tfp_bijectors.py:170-174."""
    N, H, W, C = x.shape
    return x.reshape(N, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(N, H // 2, W // 2, 4 * C)


def _conv_same(x_nhwc, kernel_hwio, bias):
    w = kernel_hwio.permute(3, 2, 0, 1)  # OIHW; F.conv2d is a cross-correlation like tf.nn.conv2d
    pad = kernel_hwio.shape[0] // 2
    y = Fnn.conv2d(x_nhwc.permute(0, 3, 1, 2), w, bias, stride=1, padding=pad)
    return y.permute(0, 2, 3, 1)


def _bn(x, p, pre, eps):
    return p[pre + "gamma"] * (x - p[pre + "mean"]) / torch.sqrt(p[pre + "var"] + eps) + p[pre + "beta"]


def convnet(xb, p, pre, eps):
    """flow_tfk_layers.py:73-84."""
    x = torch.relu(_conv_same(xb, p[pre + "nn/conv1/kernel"], p[pre + "nn/conv1/bias"]))
    x = _bn(x, p, pre + "nn/bn1/", eps)
    x = torch.relu(_conv_same(x, p[pre + "nn/conv2/kernel"], p[pre + "nn/conv2/bias"]))
    x = _bn(x, p, pre + "nn/bn2/", eps)
    x = _conv_same(x, p[pre + "nn/conv3/kernel"], p[pre + "nn/conv3/bias"])
    c = x.shape[-1] // 2
    return torch.tanh(x[..., :c]), x[..., c:]


def inv1x1_weight(p, pre):
    """flow_tfp_bijectors.py:300-303."""
    P, L, U = p[pre + "inv1x1/P"], p[pre + "inv1x1/L"], p[pre + "inv1x1/U"]
    c = P.shape[0]
    mask = torch.tril(torch.ones(c, c, dtype=P.dtype), -1)
    Lm = L * mask + torch.eye(c, dtype=P.dtype)
    Um = U * mask.T + torch.diag(p[pre + "inv1x1/sign_S"] * torch.exp(p[pre + "inv1x1/log_S"]))
    return P @ (Lm @ Um)


def step_forward(u, p, pre, cfg, evals_per_step=1):
    """flow_glow.py:21-22 (+ the three fldj terms).  ``evals_per_step=2`` re-evaluates the coupling network
    for the log-det like TFP's forward + forward_log_det_jacobian pair does -- used
    only to time the 'faithful' CPU baseline; the value is identical."""
    _, h, w, _ = u.shape
    a = u * torch.exp(p[pre + "actnorm/log_scale"]) + p[pre + "actnorm/shift"]
    v = a @ inv1x1_weight(p, pre)
    c = v.shape[-1] // 2
    va, vb = v[..., :c], v[..., c:]
    log_s, t = convnet(vb, p, pre, cfg["bn_eps"])
    ya = torch.exp(log_s) * va + t
    if evals_per_step == 2:
        log_s, _ = convnet(vb, p, pre, cfg["bn_eps"])
    ld = h * w * (p[pre + "actnorm/log_scale"].sum() + p[pre + "inv1x1/log_S"].sum()) + log_s.sum(dim=(1, 2, 3))
    return torch.cat([ya, vb], dim=-1), ld


def log_prob(x, p, cfg, evals_per_step=1):
    """flow_builder.py:127-144 + flow_glow.py forward graphs; x: [N,H,W,C] torch CPU tensor."""
    dt = x.dtype
    N = x.shape[0]
    mn, mx = cfg["minval"], cfg["maxval"]
    u01 = (x - mn) / (mx - mn)
    ld = torch.full((N,), -math.log(mx - mn) * x[0].numel(), dtype=dt)
    if cfg["use_logit"]:
        a = cfg["alpha"]
        pp = (1.0 - 2.0 * a) * u01 + a
        y = torch.log(pp) - torch.log(1.0 - pp)
        ld = ld + (-torch.log(pp) - torch.log(1.0 - pp) + math.log(1.0 - 2.0 * a)).sum(dim=(1, 2, 3))
    else:
        y = u01 - 0.5
    s = 2 ** cfg["L"]
    Hl, Wl = cfg["H"] // s, cfg["W"] // s
    zs = []
    hcur = y
    for lvl in range(cfg["L"]):
        u = squeeze(hcur)
        for k in reversed(range(cfg["K"])):
            u, l = step_forward(u, p, "b%d/s%d/" % (lvl, k), cfg, evals_per_step)
            ld = ld + l
        if lvl < cfg["L"] - 1:
            c = u.shape[-1] // 2
            zs.append(u[..., :c].reshape(N, Hl, Wl, -1))
            hcur = u[..., c:]
        else:
            zs.append(u)
    z = torch.cat(zs, dim=-1)
    half_log_2pi = 0.5 * math.log(2.0 * math.pi)
    if cfg["learntop"]:
        e = (z - p["prior/loc"]) / torch.exp(p["prior/log_scale"])
        lp = -0.5 * e * e - p["prior/log_scale"] - half_log_2pi
    else:
        lp = -0.5 * z * z - half_log_2pi
    return lp.sum(dim=(1, 2, 3)) + ld, z


def log_prob_and_grad(x_np, params, cfg, dtype=torch.float64):
    """Oracle for compute_grad_logprob (run_basis_sep.py:73-79): (log_prob[N], d sum(log_prob) / dx)."""
    p = to_torch(params, dtype)
    x = torch.tensor(np.asarray(x_np), dtype=dtype, device="cpu", requires_grad=True)
    lp, _ = log_prob(x, p, cfg)
    (g,) = torch.autograd.grad(lp.sum(), x)
    return lp.detach().numpy(), g.numpy()
