#!/usr/bin/env python3
"""Generate tests/golden/*.npz: seeded inputs + expected outputs of the Glow path from the fp64 CPU oracle.

The reference (TF 2.2 / TFP 0.9) cannot run in the build container, so these vectors come from the repo's own
restatement (oracle/glowref.py + oracle/glowref_torch.py for the input gradient), not from TensorFlow; they pin the
oracle against regressions and give the GPU tests a data-only target.  Weights are regenerated from the seed by
audiosourcesep_amd.synthetic.synthetic_params (NumPy PCG64 streams are stable across versions) -- only inputs and
expected outputs are stored.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from audiosourcesep_amd.config import GlowConfig  # noqa: E402
from audiosourcesep_amd.synthetic import synthetic_params, synthetic_mel_tiles  # noqa: E402
from oracle import glowref as R  # noqa: E402
from oracle import glowref_torch as RT  # noqa: E402

CASES = {
    "L2_K3_F128": dict(H=16, W=16, C=1, L=2, K=3, F=128),
    "L3_K2_F128_rect": dict(H=16, W=24, C=1, L=3, K=2, F=128),
    "L4_K2_F128": dict(H=16, W=16, C=1, L=4, K=2, F=128),
    "L2_K2_F128_logit_notop": dict(H=8, W=8, C=1, L=2, K=2, F=128, use_logit=True, alpha=1e-4, learntop=False),
    "L3_K2_F512": dict(H=16, W=16, C=1, L=3, K=2, F=512),
}
SEED_W, SEED_X, N = 2024, 1234, 3


def make(name, kw):
    cfg = GlowConfig(**kw)
    p = R.cast_params(synthetic_params(cfg, seed=SEED_W), np.float64)
    x = synthetic_mel_tiles(N, cfg, seed=SEED_X, dtype=np.float32).astype(np.float64)   # inputs are exactly fp32-representable
    z, ld = R.bijector_forward(x, p, cfg.as_dict())
    lp = R.prior_log_prob(z, p, cfg.as_dict()) + ld
    lp_t, grad = RT.log_prob_and_grad(x, p, cfg.as_dict())
    assert np.allclose(lp, lp_t, rtol=1e-12)
    eps = np.random.default_rng(SEED_X + 1).standard_normal((N,) + cfg.latent_shape()).astype(np.float32).astype(np.float64)
    xs = R.sample_from_eps(eps, p, cfg.as_dict())
    # one step in isolation (level 0, step 0) and its coupling network
    h, w, c = cfg.level_shapes()[0]
    u = np.random.default_rng(SEED_X + 2).standard_normal((N, h, w, c)).astype(np.float32).astype(np.float64)
    y_step, ld_step = R.step_forward(u, p, "b0/s0/", cfg.as_dict())
    log_s, t = R.convnet(u[..., c // 2:], p, "b0/s0/", cfg.bn_eps)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"),
                        cfg=np.array([repr(kw)]), seed_w=SEED_W, x=x.astype(np.float32), z=z, logdet=ld, log_prob=lp,
                        grad=grad, eps=eps.astype(np.float32), x_sample=xs, u_step=u.astype(np.float32), y_step=y_step,
                        ld_step=ld_step, log_s=log_s, t=t)
    print(name, "log_prob", lp)


if __name__ == "__main__":
    for name, kw in CASES.items():
        make(name, kw)
