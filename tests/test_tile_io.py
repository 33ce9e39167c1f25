"""Tile I/O without TensorFlow: the reference's unittest_pipeline.py cases (TFRecord save -> load preserves shapes/counts for
1-D / 2-D / 3-D tensors, :20-49) plus framing / protobuf checks, and the shipped real mel tiles as engine input."""
import os
import struct

import numpy as np
import pytest

from audiosourcesep_amd import tile_io
from audiosourcesep_amd.config import GlowConfig

REAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_mel_tiles.npz")


def test_crc32c_known_answers():
    assert tile_io.crc32c(b"123456789") == 0xE3069283          # standard CRC-32C check value
    assert tile_io.crc32c(b"") == 0


@pytest.mark.parametrize("shape", [(5,), (6, 10), (5, 10, 3), (96, 64, 1)])
def test_tfrecord_round_trip(tmp_path, shape):
    rng = np.random.default_rng(0)
    arrays = [rng.standard_normal(shape).astype(np.float32) for _ in range(7)]
    path = tile_io.write_tfrecord(str(tmp_path / "ds"), arrays)
    assert path.endswith(".tfrecord")
    back = list(tile_io.read_tfrecord(path))
    assert len(back) == 7 and all(b.shape == shape and b.dtype == np.float32 for b in back)
    for a, b in zip(arrays, back):
        np.testing.assert_array_equal(a, b)


def test_example_bytes_are_the_tf_wire_format():
    """Hand-checked encoding of a tiny Example (field numbers of tf.train.Example / Features / Feature / FloatList)."""
    data = tile_io.serialize_example(np.array([[1.0, 2.0]], np.float32))
    np.testing.assert_array_equal(tile_io.parse_example(data), [[1.0, 2.0]])
    assert data[0] == 0x0A                                       # Example.features, length delimited
    assert b"\x0a\x05array" in data and b"\x0a\x05shape" in data  # map keys
    assert struct.pack("<ff", 1.0, 2.0) in data                  # packed floats
    assert b"\x1a\x04\x0a\x02\x01\x02" in data                   # Feature.int64_list{ value: [1, 2] packed }
    # corrupting the payload is detected by the masked CRC
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = tile_io.write_tfrecord(os.path.join(d, "x"), [np.arange(4, dtype=np.float32)])
        raw = bytearray(open(p, "rb").read())
        raw[20] ^= 0xFF
        open(p, "wb").write(raw)
        with pytest.raises(ValueError):
            list(tile_io.read_tfrecord(p))


def test_tiling_and_real_tile_statistics():
    g = np.load(REAL)
    gt1 = g["gt1"]
    assert gt1.shape == (4, 96, 64) and gt1.dtype == np.float32
    # real tiles live inside the dB clip range with strong time / mel correlation
    assert gt1.min() >= -100 and gt1.max() <= 20
    spec = np.concatenate(list(gt1), axis=1)                    # [96, 256] spectrogram of 4 consecutive tiles
    tiles = tile_io.tiles_from_spectrogram(np.concatenate([spec, spec[:, :10]], axis=1))
    assert tiles.shape == (4, 96, 64, 1)
    np.testing.assert_array_equal(tiles[..., 0], gt1)
    assert tile_io.MEL_FRONTEND["n_mels"] == 96 and tile_io.MEL_FRONTEND["hop_length"] == 512


@pytest.mark.gpu
def test_real_tiles_through_the_engine_yaml_geometry():
    """The reference's melspec geometry (configs/melspec_glow.yml: 96x64, L=3) on real tiles: log_prob, gradient and BASIS
    mixing consistency against the oracle (K reduced so that the fp64 oracle finishes in seconds)."""
    import torch
    from audiosourcesep_amd import basis
    from audiosourcesep_amd.synthetic import calibrated_engine
    from oracle import glowref as R
    from oracle import glowref_torch as RT
    g = np.load(REAL)
    cfg = GlowConfig(H=96, W=64, C=1, L=3, K=3, F=512)
    eng, params = calibrated_engine(cfg, device=0, init_tiles=32)
    x = g["gt1"][:2, :, :, None].astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    lp, dx = eng.log_prob_grad(xt)
    lp_ref, g_ref = RT.log_prob_and_grad(x.astype(np.float64), params, cfg.as_dict())
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref, rtol=1e-6)
    np.testing.assert_allclose(dx.cpu().numpy(), g_ref, atol=2e-4 * np.abs(g_ref).max(), rtol=2e-3)
    np.testing.assert_allclose(eng.log_prob(xt).cpu().numpy(), R.log_prob(x.astype(np.float64), R.cast_params(params, np.float64), cfg.as_dict()),
                               rtol=1e-6)
    # the shipped mixture is the power sum of the shipped sources up to the K = 2 normalisation of g (run_basis_sep.py:139)
    mix = basis.mixing_db(torch.from_numpy(g["gt1"]), torch.from_numpy(g["gt2"])).numpy()
    assert np.median(np.abs(mix + 10 * np.log10(2.0) - g["mixed"])) < 1.5


def test_truncated_records_raise_clean_errors(tmp_path):
    """A TFRecord file cut anywhere inside a record is a ValueError naming the part that is missing (never a struct.error)."""
    from audiosourcesep_amd import tile_io
    p = str(tmp_path / "a.tfrecord")
    tile_io.write_tfrecord(p, [np.arange(12, dtype=np.float32).reshape(3, 4)])
    raw = open(p, "rb").read()
    assert len(list(tile_io.read_tfrecord(p))) == 1
    for cut in (3, 8, 10, 12, 20, len(raw) - 5, len(raw) - 1):
        open(p, "wb").write(raw[:cut])
        with pytest.raises(ValueError, match="truncated"):
            list(tile_io.read_tfrecord(p))
    bad = bytearray(raw)
    bad[2] ^= 1                                        # a flipped length byte is caught by the length CRC before it is believed
    open(p, "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="CRC mismatch"):
        list(tile_io.read_tfrecord(p))
