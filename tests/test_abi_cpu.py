"""CPU-side checks of the drop-in boundary: the library builds for gfx950, loads without a GPU and exports every
symbol include/glowk.h declares; host-only entry points (create / set / get / destroy) validate their arguments
the way the reference constructor does.  No compute call is made here."""
import ctypes
import os
import re

import numpy as np
import pytest

import __graft_entry__ as graft
from audiosourcesep_amd import _lib
from audiosourcesep_amd.config import GlowConfig, CONFIG_A, CONFIG_B, CONFIG_YAML


@pytest.fixture(scope="module")
def lib():
    graft.build()
    return _lib.load()


def header_functions(repo_root):
    text = open(os.path.join(repo_root, "include", "glowk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(glowk_[a-z0-9_]+)\s*\(", text))


def test_every_declared_symbol_is_exported_and_bound(lib, repo_root):
    declared = header_functions(repo_root)
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    text = open(os.path.join(repo_root, "include", "glowk.h")).read()
    assert lib.glowk_version() == int(re.search(r"#define GLOWK_VERSION (\d+)", text).group(1))


def test_config_struct_matches_header(repo_root):
    text = open(os.path.join(repo_root, "include", "glowk.h")).read()
    body = re.search(r"typedef struct glowk_config \{(.*?)\} glowk_config;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names += [n.strip() for n in decl.split(None, 1)[1].split(",")]
    assert names == [f[0] for f in _lib.GlowkConfigStruct._fields_]
    assert ctypes.sizeof(_lib.GlowkConfigStruct) == 12 * 4


def test_tensor_ids_match_header(repo_root):
    text = open(os.path.join(repo_root, "include", "glowk.h")).read()
    enum = dict((k, int(v)) for k, v in re.findall(r"(GLOWK_[A-Z0-9_]+) = (\d+)", text))
    for name, tid in _lib.STEP_TENSOR_IDS.items():
        key = "GLOWK_" + name.replace("nn/", "").replace("/", "_").upper()
        assert enum[key] == tid, name
    assert enum["GLOWK_PRIOR_LOC"] == 100 and enum["GLOWK_PRIOR_LOG_SCALE"] == 101
    assert enum["GLOWK_NUM_STEP_TENSORS"] == len(_lib.STEP_TENSOR_IDS)


def _create(lib, cfg_kwargs):
    c = _lib.GlowkConfigStruct(**cfg_kwargs)
    h = ctypes.c_void_p()
    rc = lib.glowk_create(ctypes.byref(c), 0, ctypes.byref(h))
    return rc, h


BASE = dict(H=64, W=64, C=1, L=3, K=2, F=512, learntop=1, use_logit=0, minval=-100.0, maxval=20.0, alpha=1e-10, bn_eps=1e-3)


def test_create_validates_like_build_glow(lib):
    rc, h = _create(lib, dict(BASE, L=5))
    assert rc != 0 and b"L should be 2, 3 or 4" in lib.glowk_last_error()   # flow_builder.py:76-77
    rc, h = _create(lib, dict(BASE, H=60))
    assert rc != 0                                                            # Squeeze asserts even sizes, :165-166
    rc, h = _create(lib, dict(BASE, F=100))
    assert rc != 0
    rc, h = _create(lib, BASE)
    assert rc == 0
    # tensor sizes follow the reference's variable shapes for c = 4, 8, 16
    for level, c in enumerate([4, 8, 16]):
        assert lib.glowk_tensor_size(h, level, _lib.STEP_TENSOR_IDS["actnorm/log_scale"]) == c
        assert lib.glowk_tensor_size(h, level, _lib.STEP_TENSOR_IDS["inv1x1/L"]) == c * c
        assert lib.glowk_tensor_size(h, level, _lib.STEP_TENSOR_IDS["nn/conv1/kernel"]) == 9 * (c // 2) * 512
        assert lib.glowk_tensor_size(h, level, _lib.STEP_TENSOR_IDS["nn/conv2/kernel"]) == 512 * 512
        assert lib.glowk_tensor_size(h, level, _lib.STEP_TENSOR_IDS["nn/conv3/kernel"]) == 9 * 512 * c
    assert lib.glowk_tensor_size(h, -1, 100) == 8 * 8 * 64
    # host round trip of a tensor; wrong size is refused
    a = np.arange(16, dtype=np.float32)
    assert lib.glowk_set_tensor(h, 0, 1, 4, a.ctypes.data_as(_lib._fp), 16) == 0
    b = np.zeros(16, np.float32)
    assert lib.glowk_get_tensor(h, 0, 1, 4, b.ctypes.data_as(_lib._fp), 16) == 0
    np.testing.assert_array_equal(a, b)
    assert lib.glowk_set_tensor(h, 0, 1, 4, a.ctypes.data_as(_lib._fp), 15) != 0
    assert lib.glowk_set_tensor(h, 0, 7, 4, a.ctypes.data_as(_lib._fp), 16) != 0   # no such step
    # compute before finalize is refused (and never silently falls back)
    assert lib.glowk_log_prob(h, None, 1, None, None, None) != 0
    assert lib.glowk_workspace_bytes(h, 4, 0) > 0
    assert lib.glowk_destroy(h) == 0


def test_config_shapes_and_flop_model():
    assert CONFIG_A.level_shapes() == [(16, 16, 4), (8, 8, 8)] and CONFIG_A.latent_shape() == (8, 8, 16)
    assert CONFIG_B.level_shapes() == [(32, 32, 4), (16, 16, 8), (8, 8, 16)] and CONFIG_B.latent_shape() == (8, 8, 64)
    assert CONFIG_YAML.level_shapes() == [(48, 32, 4), (24, 16, 8), (12, 8, 16)]
    # 3.024 / 25.72 / 48.23 GFLOP per tile, 0.1966 / 1.835 / 3.441 MB of activations
    assert abs(CONFIG_A.flop_per_tile() / 1e9 - 3.024) < 0.01
    assert abs(CONFIG_B.flop_per_tile() / 1e9 - 25.72) < 0.01
    assert abs(CONFIG_YAML.flop_per_tile() / 1e9 - 48.23) < 0.02
    assert abs(CONFIG_B.act_bytes_per_tile() / 1e6 - 1.835) < 0.001
    with pytest.raises(ValueError):
        GlowConfig(L=1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libglowk.so"))
    with pytest.raises(_lib.GlowkLibraryMissing):
        _lib.load()


def test_batch_chunking_host_logic():
    """GlowEngine splits a batch beyond glowk_max_tiles() into contiguous chunks (pure host logic, no GPU needed)."""
    from audiosourcesep_amd.engine import GlowEngine
    from audiosourcesep_amd.config import GlowConfig

    class FakeLib:
        def glowk_max_tiles(self, h):
            return 10

        def glowk_get_precision(self, h):
            return 0

        def glowk_workspace_bytes(self, h, n, with_grad):
            return n * (3000 if with_grad else 1000)

    e = GlowEngine.__new__(GlowEngine)
    e.lib, e.h, e._max_tiles_cap, e.cfg = FakeLib(), None, None, GlowConfig(H=64, W=64, C=1, L=3, K=2, F=128)
    e._finalized = True
    e._free_bytes = lambda: 10 ** 12
    assert e.max_tiles == 10 and e._chunks(0) == [] and e._chunks(10) == [(0, 10)]
    assert e._chunks(25) == [(0, 10), (10, 20), (20, 25)]
    e._max_tiles_cap = 3
    assert e._chunks(7) == [(0, 3), (3, 6), (6, 7)]
    assert e.grad_max_tiles == 3                      # never above max_tiles
    e._max_tiles_cap = None
    assert e.grad_max_tiles == 10
    # the gradient chunk is the largest batch whose workspace + saves fit 60 % of the free memory
    e._free_bytes = lambda: 20000          # budget 12 000 bytes at 3 000 per tile
    e._grad_chunk = (None, 0)
    assert e.grad_max_tiles == 4
    e._free_bytes = lambda: 100            # not even one tile fits: chunks of one, the allocation itself will say so
    e._grad_chunk = (None, 0)
    assert e.grad_max_tiles == 1


def test_status_and_policy_enums_match_header(repo_root):
    text = open(os.path.join(repo_root, "include", "glowk.h")).read()
    enum = dict((k, int(v)) for k, v in re.findall(r"(GLOWK_[A-Z0-9_]+) = (\d+)", text))
    assert (enum["GLOWK_OK"], enum["GLOWK_ERR"], enum["GLOWK_ERR_RANGE"]) == (_lib.OK, _lib.ERR, _lib.ERR_RANGE)
    assert (enum["GLOWK_RANGE_IGNORE"], enum["GLOWK_RANGE_ERROR"], enum["GLOWK_RANGE_FALLBACK"]) == (_lib.RANGE_IGNORE, _lib.RANGE_ERROR, _lib.RANGE_FALLBACK)
    assert (enum["GLOWK_PREC_F32"], enum["GLOWK_PREC_F16X3"], enum["GLOWK_PREC_F16X2"]) == (_lib.PREC_F32, _lib.PREC_F16X3, _lib.PREC_F16X2)


def test_range_policy_and_widths_on_the_host_side(lib):
    """Host-only entry points: the range policy defaults to GLOWK_RANGE_ERROR and validates; glowk_create takes exactly the
    instantiated network widths (the reference's own trained flows used n_filters = 256)."""
    for F in (128, 256, 384, 512):
        rc, h = _create(lib, dict(BASE, F=F))
        assert rc == 0, F
        assert lib.glowk_get_range_policy(h) == _lib.RANGE_ERROR
        assert lib.glowk_set_range_policy(h, _lib.RANGE_FALLBACK) == 0 and lib.glowk_get_range_policy(h) == _lib.RANGE_FALLBACK
        assert lib.glowk_set_range_policy(h, 7) != 0
        assert lib.glowk_tensor_size(h, 0, _lib.STEP_TENSOR_IDS["inv1x1/P_inv"]) == 16
        # gradient-path bytes need the launch policy, i.e. finalised weights, only for the split modes: fp32 is closed form
        assert lib.glowk_workspace_bytes(h, 8, 1) > lib.glowk_workspace_bytes(h, 8, 0) > 0
        assert lib.glowk_destroy(h) == 0
    for F in (0, 64, 100, 640, 1024):
        rc, h = _create(lib, dict(BASE, F=F))
        assert rc != 0 and b"n_filters" in lib.glowk_last_error()


def test_bench_launches_its_own_ranks(monkeypatch, repo_root, capsys):
    """`python bench.py --gpus N` outside torch.distributed.run must start N ranks itself, before touching the GPU, and
    hand back their exit code; a WORLD_SIZE that contradicts --gpus is an error."""
    import importlib
    import sys
    bench = importlib.import_module("bench")
    calls = {}

    class Done:
        returncode = 7
        stdout = "[Gloo] Rank 0 is connected to 1 peer ranks.\n{\"metric\": \"m\", \"value\": 1}\n"

    def fake_run(cmd, env=None, **kw):
        calls["cmd"], calls["env"] = cmd, env
        return Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "m", "value": 1}' and "[Gloo]" in out.err      # only the JSON line reaches stdout
    cmd = calls["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE" in str(e.value.code)
