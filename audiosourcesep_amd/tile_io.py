"""Tile I/O of the reference's data pipeline, without TensorFlow.

* TFRecord files written by ``datasets/preprocessing.py:197-271``: one ``tf.train.Example`` per tensor with features
  ``'array'`` (float_list, the flattened values) and ``'shape'`` (int64_list).  ``read_tfrecord`` / ``write_tfrecord``
  implement the TFRecord framing (length, masked CRC-32C, payload, masked CRC-32C) and the two protobuf messages by hand.
* mel front-end constants of ``datasets/wav_to_spec.py:84-98`` and the 2.04 s tiling of ``run_basis_sep.py:346-351``.
Host-side only; feeds ``[N, 96, 64, 1]`` float32 dB tiles to the engine.
"""
import struct

import numpy as np

MEL_FRONTEND = dict(sampling_rate=16000, n_fft=2048, hop_length=512, n_mels=96, fmin=125.0, fmax=7600.0,
                    length_sec=2.04, frames_per_tile=64, db_min=-100.0, db_max=20.0)

# ---- CRC-32C (Castagnoli), table driven ---------------------------------------------------------------------------
_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- minimal protobuf -----------------------------------------------------------------------------------------------
def _varint(n: int) -> bytes:
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _ld(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _fields(buf):
    """Yield (field_number, wire_type, value) of one message; length-delimited values are memoryviews."""
    pos, mv = 0, memoryview(buf)
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            val = mv[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = mv[pos:pos + 4]
            pos += 4
        elif wt == 1:
            val = mv[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, val


def serialize_example(array) -> bytes:
    """preprocessing.py:197-216: Example{features{feature{'array': float_list, 'shape': int64_list}}}."""
    a = np.asarray(array, dtype=np.float32)
    float_list = _ld(1, a.reshape(-1).astype("<f4").tobytes())                  # FloatList.value, packed
    int_list = _ld(1, b"".join(_varint(int(d)) for d in a.shape))               # Int64List.value, packed
    feat_array = _ld(2, float_list)                                             # Feature.float_list = 2
    feat_shape = _ld(3, int_list)                                               # Feature.int64_list = 3
    entries = b"".join(_ld(1, _ld(1, k) + _ld(2, v)) for k, v in ((b"array", feat_array), (b"shape", feat_shape)))
    return _ld(1, entries)                                                      # Example.features = 1 (Features.feature = 1)


def parse_example(buf: bytes) -> np.ndarray:
    """preprocessing.py:247-271: parse one Example and reshape 'array' to 'shape'."""
    values, shape = None, None
    for f, _, features in _fields(bytes(buf)):
        if f != 1:
            continue
        for f2, _, entry in _fields(bytes(features)):
            if f2 != 1:
                continue
            key, feature = None, None
            for f3, _, v in _fields(bytes(entry)):
                if f3 == 1:
                    key = bytes(v)
                elif f3 == 2:
                    feature = bytes(v)
            for f4, _, lst in _fields(feature):
                payload = bytes(lst)
                if key == b"array" and f4 == 2:
                    vals = []
                    for f5, wt, v in _fields(payload):
                        if f5 == 1 and wt == 2:
                            vals.append(np.frombuffer(bytes(v), dtype="<f4"))
                        elif f5 == 1 and wt == 5:
                            vals.append(np.frombuffer(bytes(v), dtype="<f4"))
                    values = np.concatenate(vals) if vals else np.zeros(0, np.float32)
                elif key == b"shape" and f4 == 3:
                    dims = []
                    for f5, wt, v in _fields(payload):
                        if f5 == 1 and wt == 2:
                            b, p = bytes(v), 0
                            while p < len(b):
                                d, p = _read_varint(b, p)
                                dims.append(d)
                        elif f5 == 1 and wt == 0:
                            dims.append(v)
                    shape = tuple(int(d) for d in dims)
    if values is None or shape is None:
        raise ValueError("record is not a {'array', 'shape'} Example")
    return values.reshape(shape)


def write_tfrecord(path, arrays):
    """save_tf_records (preprocessing.py:228-244); appends '.tfrecord' like the reference if missing."""
    if not path.endswith("tfrecord"):
        path += ".tfrecord"
    with open(path, "wb") as f:
        for a in arrays:
            data = serialize_example(a)
            hdr = struct.pack("<Q", len(data))
            f.write(hdr + struct.pack("<I", _masked_crc(hdr)) + data + struct.pack("<I", _masked_crc(data)))
    return path


def read_tfrecord(path, check_crc=True):
    """load_tf_records (preprocessing.py:247-271) -> iterator of float32 arrays."""
    with open(path, "rb") as f:
        while True:
            hdr = f.read(8)
            if not hdr:
                return
            if len(hdr) != 8:
                raise ValueError("truncated TFRecord header")
            (n,) = struct.unpack("<Q", hdr)
            hc = f.read(4)
            if len(hc) != 4:
                raise ValueError("truncated TFRecord header checksum")
            (hcrc,) = struct.unpack("<I", hc)
            if check_crc and hcrc != _masked_crc(hdr):      # (before trusting the length it protects)
                raise ValueError("TFRecord CRC mismatch (length)")
            data = f.read(n)
            dc = f.read(4)
            if len(data) != n or len(dc) != 4:
                raise ValueError("truncated TFRecord payload")
            (dcrc,) = struct.unpack("<I", dc)
            if check_crc and (hcrc != _masked_crc(hdr) or dcrc != _masked_crc(data)):
                raise ValueError("TFRecord CRC mismatch")
            yield parse_example(data)


def tiles_from_spectrogram(spec_db, frames=MEL_FRONTEND["frames_per_tile"]):
    """Cut a [n_mels, T] dB mel spectrogram into independent [n, n_mels, frames, 1] tiles (run_basis_sep.py:346-351:
    long audio is handled as a batch of fixed 2.04 s tiles; a trailing partial tile is dropped)."""
    spec_db = np.asarray(spec_db, dtype=np.float32)
    n = spec_db.shape[1] // frames
    return np.ascontiguousarray(spec_db[:, :n * frames].reshape(spec_db.shape[0], n, frames).transpose(1, 0, 2)[..., None])
