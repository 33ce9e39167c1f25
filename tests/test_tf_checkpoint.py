"""TensorFlow checkpoint bundles without TensorFlow (audiosourcesep_amd/tf_checkpoint.py): container round trips, checksums,
the derived flow.variables order against the shapes it must produce.  No fixture from real TensorFlow exists in the reference
tree (it ships no Glow checkpoint), so the byte-level format is pinned by its published constants only: the table magic, the
CRC-32C check value and the masked-CRC formula shared with the TFRecord reader."""
import os
import struct

import numpy as np
import pytest

import __graft_entry__ as graft
from audiosourcesep_amd import tf_checkpoint as T
from audiosourcesep_amd.config import GlowConfig, CONFIG_B
from audiosourcesep_amd.synthetic import synthetic_params
from audiosourcesep_amd.tile_io import crc32c as crc32c_bytewise, _masked_crc


@pytest.fixture(scope="module", autouse=True)
def built():
    graft.build()


def test_crc32c_of_the_library_matches_the_known_answers():
    assert T.crc32c(b"123456789") == 0xE3069283                    # the CRC-32C check value (RFC 3720 appendix B.4)
    assert T.crc32c(b"") == 0 and T.crc32c(bytes(32)) == 0x8A9136AA  # 32 zero bytes, RFC 3720
    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 63, 64, 65, 10007):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert T.crc32c(b) == crc32c_bytewise(b)
        assert T.masked_crc32c(b) == _masked_crc(b)


def test_bundle_round_trip_many_blocks_and_dtypes(tmp_path):
    rng = np.random.default_rng(1)
    tensors = {"variables/%d%s" % (i, T.VALUE_SUFFIX): rng.standard_normal((3, i % 7 + 1, 2)).astype(np.float32) for i in range(300)}
    tensors["save_counter" + T.VALUE_SUFFIX] = np.asarray(21, dtype=np.int64)
    tensors["optimizer/iter" + T.VALUE_SUFFIX] = np.asarray(12345, dtype=np.int64)
    tensors["some/double"] = rng.standard_normal(5)
    tensors["some/empty"] = np.zeros((0, 4), np.float32)
    prefix = str(tmp_path / "ckpt-21")
    T.write_bundle(prefix, tensors)
    assert sorted(os.listdir(tmp_path)) == ["ckpt-21.data-00000-of-00001", "ckpt-21.index"]
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xdb4775248b80fb57           # the table format's magic number
    keys = [k for k, _ in T.read_table(prefix + ".index")]
    assert keys[0] == b"" and keys == sorted(keys) and len(keys) == len(tensors) + 1
    back = T.read_bundle(prefix)
    assert set(back) == set(tensors)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape
        np.testing.assert_array_equal(back[k], v)


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "ckpt-1")
    T.write_bundle(prefix, {"a": np.arange(100, dtype=np.float32), "b": np.ones((4, 4), np.float32)})
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[17] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError, match="checksum mismatch for 'a'"):
        T.read_bundle(prefix)
    assert T.read_bundle(prefix, verify=False)["b"].sum() == 16
    T.write_bundle(prefix, {"a": np.arange(100, dtype=np.float32)})
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[5] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="block checksum"):
        T.read_bundle(prefix)
    open(prefix + ".index", "wb").write(b"not a table")
    with pytest.raises(ValueError, match="bad magic"):
        T.read_bundle(prefix)


@pytest.mark.parametrize("cfg", [GlowConfig(H=16, W=16, C=1, L=2, K=3, F=128), GlowConfig(H=32, W=16, C=1, L=4, K=2, F=256, learntop=False), CONFIG_B])
def test_variable_order_covers_every_variable_once(cfg):
    order = T.variable_order(cfg)
    per_step = 22                                                  # flow_tfp_bijectors.py:236-239, 281-294 + flow_tfk_layers.py:56-70
    assert len(order) == cfg.L * cfg.K * per_step + (2 if cfg.learntop else 0) and len(set(order)) == len(order)
    shapes = T._expected_shapes(cfg)
    assert set(order) <= set(shapes)
    # within a step: ActNorm, then the coupling network, then the 1x1 (GlowStep's attributes in sorted order, flow_glow.py:15-22)
    s0 = [n.split("/", 2)[2] for n in order[:per_step]]
    assert s0[:2] == ["actnorm/log_scale", "actnorm/shift"] and s0[2] == "nn/conv1/kernel" and s0[-6:] == ["inv1x1/L", "inv1x1/log_S", "inv1x1/P", "inv1x1/P_inv", "inv1x1/sign_S", "inv1x1/U"]
    assert order[per_step].startswith("b0/s1/") and order[cfg.K * per_step].startswith("b1/s0/")


def test_checkpoint_import_export_round_trip_and_shape_check(tmp_path):
    cfg = GlowConfig(H=16, W=16, C=1, L=2, K=2, F=128)
    state = synthetic_params(cfg)
    for k in list(state):
        if k.endswith("inv1x1/P"):
            state[k[:-1] + "P_inv"] = np.linalg.inv(state[k]).astype(np.float32)
    prefix = str(tmp_path / "tf_ckpts" / "ckpt-3")
    os.makedirs(os.path.dirname(prefix))
    T.save_checkpoint_bundle(prefix, state, cfg)
    back = T.state_dict_from_checkpoint(prefix, cfg, reference_batchnorm=False)   # (synthetic weights carry trained-looking moving statistics)
    assert set(back) == set(state)
    for k in state:
        np.testing.assert_array_equal(back[k], np.asarray(state[k], np.float32))
    # another configuration does not fit: count, then shapes
    with pytest.raises(ValueError, match="flow variables"):
        T.state_dict_from_checkpoint(prefix, GlowConfig(H=16, W=16, C=1, L=2, K=3, F=128))
    with pytest.raises(ValueError, match="has shape"):
        T.state_dict_from_checkpoint(prefix, GlowConfig(H=16, W=16, C=1, L=2, K=2, F=256))
    # an explicit order that swaps two differently shaped tensors is refused, one that swaps equal shapes is honoured
    order = T.variable_order(cfg)
    bad = list(order)
    bad[0], bad[2] = bad[2], bad[0]
    with pytest.raises(ValueError, match="has shape"):
        T.state_dict_from_checkpoint(prefix, cfg, order=bad)
    swapped = T.state_dict_from_checkpoint(prefix, cfg, order=T.variable_order(cfg, prior_order=("log_scale", "loc")), reference_batchnorm=False)
    np.testing.assert_array_equal(swapped["prior/loc"], np.asarray(state["prior/log_scale"], np.float32))


def _swap(order, a, b):
    o = list(order)
    i, j = o.index(a), o.index(b)
    o[i], o[j] = o[j], o[i]
    return o


def test_value_invariants_catch_permutations_of_same_shaped_variables(tmp_path):
    """The shape check cannot separate the [F] vectors of a step, L / U / P / P_inv or log_S / sign_S; the values of a reference
    checkpoint can (round-2 advisor finding): a state as build_glow creates it (flow_builder.initial_variables: the reference's
    own initialisation) passes, every same-shape permutation that changes the KIND of a tensor is refused by name."""
    from audiosourcesep_amd.flow_models.flow_builder import initial_variables
    cfg = GlowConfig(H=16, W=16, C=1, L=2, K=2, F=128)
    rng = np.random.default_rng(5)
    state = initial_variables(cfg, rng)
    for k in list(state):        # a trained-looking checkpoint: everything trainable moved, frozen / masked entries untouched
        kind = k.split("/", 2)[-1]
        if kind in ("inv1x1/L",):
            state[k] = state[k] + np.tril(rng.normal(0, 0.01, state[k].shape), -1).astype(np.float32)
        elif kind in ("inv1x1/U",):
            state[k] = state[k] + np.triu(rng.normal(0, 0.01, state[k].shape), 1).astype(np.float32)
        elif kind in ("actnorm/log_scale", "actnorm/shift", "inv1x1/log_S", "nn/conv1/bias", "nn/conv2/bias", "nn/conv3/bias", "nn/bn1/gamma",
                      "nn/bn1/beta", "nn/bn2/gamma", "nn/bn2/beta", "nn/conv3/kernel"):
            state[k] = state[k] + rng.normal(0, 0.05, state[k].shape).astype(np.float32)
    prefix = str(tmp_path / "ckpt-7")
    T.save_checkpoint_bundle(prefix, state, cfg)
    back = T.state_dict_from_checkpoint(prefix, cfg)               # all invariants hold
    for k in state:
        np.testing.assert_array_equal(back[k], np.asarray(state[k], np.float32))
    order = T.variable_order(cfg)
    cases = [("b0/s1/inv1x1/P", "b0/s1/inv1x1/L", "permutation matrix|lower triangular"),
             ("b0/s0/inv1x1/U", "b0/s0/inv1x1/P_inv", "upper triangular|transpose"),
             ("b1/s0/inv1x1/sign_S", "b1/s0/inv1x1/log_S", "-1 / \\+1"),
             ("b0/s0/nn/bn1/mean", "b0/s0/nn/bn1/gamma", "moving statistics"),
             ("b1/s1/nn/bn2/var", "b1/s1/nn/conv2/bias", "moving variances|moving statistics"),
             ("b0/s1/nn/bn1/var", "b0/s1/nn/bn1/beta", "moving variances|moving statistics")]
    for a, b, msg in cases:
        with pytest.raises(ValueError, match=msg):
            T.state_dict_from_checkpoint(prefix, cfg, order=_swap(order, a, b))
    # an explicit opt-out reads anything that has the right shapes
    T.state_dict_from_checkpoint(prefix, cfg, order=_swap(order, *cases[0][:2]), check_values=False)
    # a checkpoint with TRAINED moving statistics is only accepted when the caller says the writer was not the reference
    state2 = dict(state)
    state2["b0/s0/nn/bn1/mean"] = state["b0/s0/nn/bn1/mean"] + 0.1
    T.save_checkpoint_bundle(prefix, state2, cfg)
    with pytest.raises(ValueError, match="moving statistics"):
        T.state_dict_from_checkpoint(prefix, cfg)
    T.state_dict_from_checkpoint(prefix, cfg, reference_batchnorm=False)
