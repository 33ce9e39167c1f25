"""The committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the fp64 oracle) against
both CPU restatements.  Guards the oracle and the synthetic generators against drift; the GPU tests use the same files."""
import ast
import glob
import os

import numpy as np
import pytest

from audiosourcesep_amd.config import GlowConfig
from audiosourcesep_amd.synthetic import synthetic_params
from oracle import glowref as R
from oracle import glowref_torch as RT

FILES = sorted(f for f in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz"))
               if not os.path.basename(f).startswith(("real_", "basis_real_")))   # (those two hold tiles, not oracle vectors)


def load(path):
    g = dict(np.load(path))
    cfg = GlowConfig(**ast.literal_eval(str(g["cfg"][0])))
    return cfg, g, synthetic_params(cfg, seed=int(g["seed_w"]))


def test_golden_files_present():
    assert len(FILES) >= 5


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracles_reproduce_golden(path):
    cfg, g, params = load(path)
    p = R.cast_params(params, np.float64)
    x = g["x"].astype(np.float64)
    z, ld = R.bijector_forward(x, p, cfg.as_dict())
    np.testing.assert_allclose(z, g["z"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ld, g["logdet"], rtol=1e-12)
    np.testing.assert_allclose(R.log_prob(x, p, cfg.as_dict()), g["log_prob"], rtol=1e-12)
    lp, grad = RT.log_prob_and_grad(x, p, cfg.as_dict())
    np.testing.assert_allclose(lp, g["log_prob"], rtol=1e-12)
    np.testing.assert_allclose(grad, g["grad"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(R.sample_from_eps(g["eps"].astype(np.float64), p, cfg.as_dict()), g["x_sample"], rtol=1e-9, atol=1e-9)
    y, lds = R.step_forward(g["u_step"].astype(np.float64), p, "b0/s0/", cfg.as_dict())
    np.testing.assert_allclose(y, g["y_step"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(lds, g["ld_step"], rtol=1e-12)
    # fp32 oracle mode stays within the north-star bar of its own fp64 result
    lp32 = R.log_prob(g["x"], R.cast_params(params, np.float32), cfg.as_dict())
    assert lp32.dtype == np.float32
    np.testing.assert_allclose(lp32, g["log_prob"], rtol=1e-4)
