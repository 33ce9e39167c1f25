"""CPU ORACLE (test infrastructure only) -- NumPy/torch-CPU restatement of the BASIS Langevin update
(run_basis_sep.py:131-181, glow branch) with the noise supplied by the caller.  Gradients of log_prob come from
oracle/glowref_torch.py (fp64 autograd).  Only tests/ may import this."""
import numpy as np

from . import glowref_torch as RT


def g_db(*sources):
    """run_basis_sep.py:133-141."""
    s = np.stack(sources, axis=0) * np.log(10.0) / 10.0
    m = s.max(axis=0)
    lse = m + np.log(np.exp(s - m).sum(axis=0))
    return (10.0 / np.log(10.0)) * (lse - np.log(float(len(sources))))


def grad_g_db(*sources):
    """run_basis_sep.py:143-147."""
    s = np.stack(sources, axis=0) * np.log(10.0) / 10.0
    e = np.exp(s - s.max(axis=0))
    return list(e / e.sum(axis=0))


def inner_loop(mixed, x1, x2, params1, params2, cfg, sigma_idx, sigmas, noise, delta=2e-5, T=100):
    """run_basis_sep.py:152-181; noise[t][which] are standard-normal arrays."""
    sigma, sigma_l = float(sigmas[sigma_idx]), float(sigmas[-1])
    eta = float(np.float32(delta * (sigma / sigma_l) ** 2))
    lam = 1.0 / sigma ** 2
    for t in range(T):
        e1 = np.sqrt(2.0 * eta) * noise[t][0]
        e2 = np.sqrt(2.0 * eta) * noise[t][1]
        _, g1 = RT.log_prob_and_grad(x1, params1, cfg)
        _, g2 = RT.log_prob_and_grad(x2, params2, cfg)
        mix = g_db(x1, x2)
        m1, m2 = grad_g_db(x1, x2)
        x1, x2 = x1 + eta * (g1 + lam * m1 * (mixed - mix)) + e1, x2 + eta * (g2 + lam * m2 * (mixed - mix)) + e2
    return x1, x2
