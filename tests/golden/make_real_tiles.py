#!/usr/bin/env python3
"""Generate tests/golden/basis_real_tiles.npz: the 30 real 96x64 dB mel tiles of each stem that the reference ships with its
BASIS result (basis_sep_results/beethoven_sonata_1_sep_1min/results.npz: gt1 = piano, gt2 = violin, mixed = their mixture, and
x1 / x2 = what the reference's own run separated) -- data only, no code.  They are the training set of the two noise-conditioned
priors and the mixture of the config-5 chain test (tests/test_gpu_config5_chain.py), and the workload of
`bench.py --workload basis`.  Stored as float16 (0.06 dB at -85 dB: far below the mel front end's own noise), so every value
the tests see is exactly representable in the engine's fp32.  Run from the repo root (needs /root/reference):
    python tests/golden/make_real_tiles.py
"""
import os

import numpy as np

SRC = "/root/reference/basis_sep_results/beethoven_sonata_1_sep_1min/results.npz"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "basis_real_tiles.npz")

if __name__ == "__main__":
    f = np.load(SRC)
    out = {k: f[k].astype(np.float16) for k in ("gt1", "gt2", "mixed", "x1", "x2")}
    for k, v in out.items():
        assert v.shape == (30, 96, 64), (k, v.shape)
    np.savez_compressed(DST, source=np.array(["SamArgt/AudioSourceSep basis_sep_results/beethoven_sonata_1_sep_1min/results.npz "
                                              "(arrays gt1, gt2, mixed, x1, x2; float32 -> float16)"]), **out)
    print(DST, os.path.getsize(DST), "bytes")
    g = np.load(DST)
    for k in out:
        print(k, "max |fp16 - fp32| = %.3f dB" % float(np.abs(g[k].astype(np.float32) - f[k]).max()))
