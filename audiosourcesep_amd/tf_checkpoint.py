"""TensorFlow checkpoint interchange without TensorFlow (train_utils.py:62-75, run_basis_sep.py:28-38).

The reference saves ``tf.train.Checkpoint(variables=model.variables, optimizer=optimizer)`` through a CheckpointManager.  On disk
that is a *tensor bundle*: ``<prefix>.index`` -- an immutable sorted string table (the LevelDB table format: prefix-compressed
key/value blocks with restart arrays, a 5-byte trailer per block = compression type + masked CRC-32C, an index block, a
48-byte footer ending in the magic 0xdb4775248b80fb57) whose values are ``BundleEntryProto`` messages (dtype, shape, shard,
offset, size, masked CRC-32C of the bytes) under the tensors' checkpoint keys, plus a ``BundleHeaderProto`` under the empty
key -- and ``<prefix>.data-00000-of-00001`` with the raw little-endian tensor bytes.  A tuple of variables is tracked as a
list, so variable ``i`` of ``flow.variables`` is stored under ``variables/<i>/.ATTRIBUTES/VARIABLE_VALUE``.

``read_bundle`` / ``write_bundle`` implement the container (uncompressed blocks, which is what TF's BundleWriter emits;
every CRC is verified).  ``variable_order`` is the part that CANNOT be validated in this environment (no TensorFlow, and the
reference ships no Glow checkpoint): the position of each variable in ``flow.variables`` follows ``tf.Module``'s attribute
traversal (attributes of an object in sorted order, leaves first, then sub-modules in discovery order, each object once),
which is re-derived here from the attribute names in the reference's source -- see the function.  The import therefore
checks every tensor's shape against the slot it lands in and refuses a checkpoint that does not fit; the order remains
overridable (``order=``).  Whoever has TensorFlow at hand can confirm it with
``[v.name for v in flow.variables]``.

Shapes cannot tell apart the groups a wrong guess would permute -- per step the eleven ``[F]`` vectors (conv1 / conv2 bias, and
moving_mean, moving_variance, gamma, beta of both BatchNorm layers), ActNorm's ``log_scale`` / ``shift`` (``[c]``), ``log_S`` /
``sign_S`` (``[c]``), ``L`` / ``U`` / ``P`` / ``P_inv`` (``[c, c]``), and the prior pair -- so after the shape check the import tests
what the VALUES of a reference checkpoint must satisfy (``check_value_invariants``): ``P`` is a 0/1 permutation matrix and
``P_inv`` its transpose, ``sign_S`` is +-1, ``L`` is zero above and ``U`` zero on and below the diagonal (the masked entries get
zero gradient, flow_tfp_bijectors.py:300-303), moving variances are positive -- and, because the reference never runs
BatchNormalization in training mode (the layers are called without ``training=``), moving_mean == 0 and moving_variance == 1
exactly.  A checkpoint that violates one of them is refused with the offending variable named.

Optimizer state: the reference's checkpoint also tracks ``optimizer`` (Adam / Adamax slot variables ``m`` / ``v`` and the
iteration count, under ``variables/<i>/.OPTIMIZER_SLOT/...`` and ``optimizer/...`` keys).  The import reads the flow's
variables only: a run resumed from an imported checkpoint starts its moments from zero (``glowk_apply_gradients`` creates them
at the first step) -- fine-tuning works as in train_noisy_glow.py:331-335, the first steps are not bit-for-bit a TensorFlow resume.
"""
import ctypes
import struct

import numpy as np

from . import _lib
from .tile_io import _varint, _read_varint, _ld, _fields

TABLE_MAGIC = 0xdb4775248b80fb57
HEADER_KEY = b""
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}
DTYPE_IDS = {np.dtype(v): k for k, v in DTYPES.items()}
VALUE_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"


def crc32c(data) -> int:
    buf = np.frombuffer(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview, np.ndarray)) else data, dtype=np.uint8)
    buf = np.ascontiguousarray(buf)
    return int(_lib.load().glowk_crc32c(ctypes.c_void_p(buf.ctypes.data), buf.size))


def masked_crc32c(data) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- sorted string table (LevelDB table format) ------------------------------------------------------------------------
def _read_block(buf, offset, size):
    """Contents of the block at (offset, size) after checking its trailer: 1 byte compression type, 4 bytes masked CRC-32C."""
    body = bytes(buf[offset:offset + size])
    ctype = buf[offset + size]
    (crc,) = struct.unpack_from("<I", buf, offset + size + 1)
    if masked_crc32c(body + bytes([ctype])) != crc:
        raise ValueError("tensor bundle index: block checksum mismatch at offset %d" % offset)
    if ctype != 0:
        raise ValueError("tensor bundle index: compressed blocks (type %d) are not supported" % ctype)
    return body


def _block_entries(body):
    (n_restarts,) = struct.unpack_from("<I", body, len(body) - 4)
    end = len(body) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _read_varint(body, pos)
        non_shared, pos = _read_varint(body, pos)
        vlen, pos = _read_varint(body, pos)
        key = key[:shared] + body[pos:pos + non_shared]
        pos += non_shared
        yield key, body[pos:pos + vlen]
        pos += vlen


def _handle(buf, pos=0):
    off, pos = _read_varint(buf, pos)
    size, pos = _read_varint(buf, pos)
    return off, size, pos


def read_table(path):
    """-> [(key bytes, value bytes)] of a table file, in key order."""
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError("%s is not a tensor bundle index (bad magic)" % path)
    footer = buf[len(buf) - 48:]
    _, _, pos = _handle(footer, 0)                 # metaindex handle (unused)
    ioff, isize, _ = _handle(footer, pos)
    out = []
    for _, hv in _block_entries(_read_block(buf, ioff, isize)):
        boff, bsize, _ = _handle(hv, 0)
        out.extend(_block_entries(_read_block(buf, boff, bsize)))
    return out


def _build_block(entries, restart_interval=16):
    body, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        body += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def write_table(path, entries, block_entries=64):
    """entries: [(key bytes, value bytes)], written in sorted key order."""
    entries = sorted(entries)
    out, index = bytearray(), []

    def emit(body):
        off = len(out)
        out.extend(body)
        out.extend(b"\x00" + struct.pack("<I", masked_crc32c(body + b"\x00")))
        return off, len(body)

    for i in range(0, max(len(entries), 1), block_entries):
        chunk = entries[i:i + block_entries]
        off, size = emit(_build_block(chunk))
        index.append(((chunk[-1][0] if chunk else b""), _varint(off) + _varint(size)))
    moff, msize = emit(_build_block([]))
    ioff, isize = emit(_build_block(index, restart_interval=1))
    footer = _varint(moff) + _varint(msize) + _varint(ioff) + _varint(isize)
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    with open(path, "wb") as f:
        f.write(bytes(out))


# ---- tensor bundle ------------------------------------------------------------------------------------------------------
def _parse_entry(value):
    e = dict(dtype=0, shape=[], shard=0, offset=0, size=0, crc=None)
    for field, wt, val in _fields(value):
        if field == 1:
            e["dtype"] = val
        elif field == 2:       # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
            for f2, _, dim in _fields(bytes(val)):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _fields(bytes(dim)):
                        if f3 == 1:
                            size = v3
                    e["shape"].append(size)
        elif field == 3:
            e["shard"] = val
        elif field == 4:
            e["offset"] = val
        elif field == 5:
            e["size"] = val
        elif field == 6:
            e["crc"] = struct.unpack("<I", bytes(val))[0]
    return e


def read_bundle(prefix, verify=True):
    """-> {checkpoint key: ndarray} of every numeric tensor of the bundle ``prefix`` (string tensors -- the object graph -- are skipped)."""
    items = read_table(prefix + ".index")
    if not items or items[0][0] != HEADER_KEY:
        raise ValueError("tensor bundle index has no header entry")
    num_shards, endian = 1, 0
    for field, _, val in _fields(items[0][1]):
        if field == 1:
            num_shards = val
        elif field == 2:
            endian = val
    if endian != 0:
        raise ValueError("big-endian tensor bundles are not supported")
    shards = {}
    out = {}
    for key, value in items[1:]:
        e = _parse_entry(value)
        if e["dtype"] not in DTYPES:
            continue
        if e["shard"] not in shards:
            shards[e["shard"]] = np.memmap("%s.data-%05d-of-%05d" % (prefix, e["shard"], num_shards), dtype=np.uint8, mode="r")
        raw = shards[e["shard"]][e["offset"]:e["offset"] + e["size"]]
        if raw.size != e["size"]:
            raise ValueError("tensor bundle data file is truncated at %r" % key.decode())
        if verify and e["crc"] is not None and masked_crc32c(np.ascontiguousarray(raw)) != e["crc"]:
            raise ValueError("tensor bundle: checksum mismatch for %r" % key.decode())
        out[key.decode()] = np.frombuffer(raw.tobytes(), dtype=DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
    return out


def write_bundle(prefix, tensors):
    """{checkpoint key: ndarray} -> ``prefix.index`` + ``prefix.data-00000-of-00001`` (one shard, little endian)."""
    header = _varint((1 << 3) | 0) + _varint(1) + _ld(3, _varint((1 << 3) | 0) + _varint(1))     # num_shards = 1, version.producer = 1
    entries = [(HEADER_KEY, header)]
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for key in sorted(tensors):
            a = np.asarray(tensors[key])
            a = a if a.ndim == 0 else np.ascontiguousarray(a)      # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in DTYPE_IDS:
                raise TypeError("unsupported dtype %s for %s" % (a.dtype, key))
            raw = a.tobytes()
            shape = b"".join(_ld(2, _varint((1 << 3) | 0) + _varint(d)) for d in a.shape)
            e = _varint((1 << 3) | 0) + _varint(DTYPE_IDS[a.dtype]) + _ld(2, shape)
            if offset:
                e += _varint((4 << 3) | 0) + _varint(offset)
            e += _varint((5 << 3) | 0) + _varint(len(raw)) + _varint((6 << 3) | 5) + struct.pack("<I", masked_crc32c(np.frombuffer(raw, np.uint8)))
            entries.append((key.encode(), e))
            f.write(raw)
            offset += len(raw)
    write_table(prefix + ".index", entries)


# ---- position of every variable in flow.variables -------------------------------------------------------------------------
def variable_order(cfg, prior_order=("loc", "log_scale")):
    """Names (this repository's) of ``flow.variables[0], flow.variables[1], ...`` for a flow built by the reference's
    ``build_glow`` -- DERIVED from the source, not observed (see the module docstring).

    ``tf.Module.variables`` walks ``vars(obj)`` in sorted attribute order, yields the variables it finds in (nests of) those
    attributes, remembers every object it has seen, then recurses into the sub-modules in the order it met them:

    * TransformedDistribution: ``_bijector`` sorts before ``_distribution`` -> all bijector variables, then the prior's;
    * Invert -> Chain([glow, preprocessing]) -> GlowBijector_{2,3,4}blocks: ``glow_block1`` < ``glow_block2`` < ...;
    * GlowBlock (flow_glow.py:40-52): ``bijector`` (a Chain over objects met again under ``chain``) < ``chain`` = steps 0..K-1;
    * GlowStep (:15-22): ``actnorm`` < ``bijector`` (Chain of already-seen objects) < ``coupling_layer`` < ``inv1x1conv``;
    * ActNorm: ``log_scale`` < ``shift``;  Invertible1x1Conv: ``L`` < ``Log_s`` < ``P`` < ``P_inv`` < ``Sign_s`` < ``U``
      (flow_tfp_bijectors.py:281-294; capitals sort before ``_`` and lower case);
    * AffineCouplingLayerSplit.shift_and_log_scale_fn is a Keras layer: its private ``_layers`` list (creation order conv1,
      batch_norm_1, conv2, batch_norm_2, conv3; flow_tfk_layers.py:56-71) is met first; inside a layer ``_non_trainable_weights``
      sorts before ``_trainable_weights``: BatchNormalization -> moving_mean, moving_variance, gamma, beta; Conv2D -> kernel, bias;
    * prior (flow_builder.py:131-139): ``prior_order`` -- loc and the TransformedVariable's pre-transformed (log) scale have the
      same shape and no value-level invariant separates them; pass the other order if TensorFlow says so.
    Same-shaped groups the shape check cannot separate (all but the prior pair are covered by ``check_value_invariants`` where
    their values differ in kind): the eleven [F] vectors of a step, log_scale / shift, log_S / sign_S, L / U / P / P_inv."""
    names = []
    for lvl in range(cfg.L):
        for k in range(cfg.K):
            pre = "b%d/s%d/" % (lvl, k)
            names += [pre + "actnorm/log_scale", pre + "actnorm/shift"]
            names += [pre + "nn/conv1/kernel", pre + "nn/conv1/bias", pre + "nn/bn1/mean", pre + "nn/bn1/var", pre + "nn/bn1/gamma", pre + "nn/bn1/beta",
                      pre + "nn/conv2/kernel", pre + "nn/conv2/bias", pre + "nn/bn2/mean", pre + "nn/bn2/var", pre + "nn/bn2/gamma", pre + "nn/bn2/beta",
                      pre + "nn/conv3/kernel", pre + "nn/conv3/bias"]
            names += [pre + "inv1x1/L", pre + "inv1x1/log_S", pre + "inv1x1/P", pre + "inv1x1/P_inv", pre + "inv1x1/sign_S", pre + "inv1x1/U"]
    if cfg.learntop:
        names += ["prior/" + n for n in prior_order]
    return names


def _expected_shapes(cfg):
    shapes = {}
    F = cfg.F
    for lvl, (h, w, c) in enumerate(cfg.level_shapes()):
        ci = c // 2
        for k in range(cfg.K):
            pre = "b%d/s%d/" % (lvl, k)
            shapes.update({pre + "actnorm/log_scale": (c,), pre + "actnorm/shift": (c,), pre + "inv1x1/L": (c, c), pre + "inv1x1/U": (c, c),
                           pre + "inv1x1/P": (c, c), pre + "inv1x1/P_inv": (c, c), pre + "inv1x1/log_S": (c,), pre + "inv1x1/sign_S": (c,),
                           pre + "nn/conv1/kernel": (3, 3, ci, F), pre + "nn/conv1/bias": (F,), pre + "nn/conv2/kernel": (1, 1, F, F),
                           pre + "nn/conv2/bias": (F,), pre + "nn/conv3/kernel": (3, 3, F, c), pre + "nn/conv3/bias": (c,)})
            for bn in ("bn1", "bn2"):
                for t in ("gamma", "beta", "mean", "var"):
                    shapes[pre + "nn/%s/%s" % (bn, t)] = (F,)
    shapes["prior/loc"] = shapes["prior/log_scale"] = tuple(cfg.latent_shape())
    return shapes


def check_value_invariants(state, cfg, reference_batchnorm=True):
    """Value-level invariants of a reference checkpoint (module docstring): raises ValueError naming the first variable that
    breaks one -- which is what a wrong ``variable_order`` produces when it permutes same-shaped neighbours.
    ``reference_batchnorm``: also require moving_mean == 0 and moving_variance == 1 (the reference never updates them)."""
    def bad(name, what):
        raise ValueError("checkpoint tensor mapped to %s %s: either the checkpoint was not written by the reference's build_glow "
                         "or the derived variable order is wrong for it (pass order=)" % (name, what))

    for lvl, (h, w, c) in enumerate(cfg.level_shapes()):
        eye = np.eye(c)
        for k in range(cfg.K):
            pre = "b%d/s%d/" % (lvl, k)
            P, Pinv = np.asarray(state[pre + "inv1x1/P"], np.float64), np.asarray(state[pre + "inv1x1/P_inv"], np.float64)
            if not (np.isin(P, (0.0, 1.0)).all() and (P.sum(0) == 1).all() and (P.sum(1) == 1).all()):
                bad(pre + "inv1x1/P", "is not a 0/1 permutation matrix")
            if not np.array_equal(Pinv, P.T):
                bad(pre + "inv1x1/P_inv", "is not the transpose (= inverse) of P")
            if not np.isin(np.asarray(state[pre + "inv1x1/sign_S"]), (-1.0, 1.0)).all():
                bad(pre + "inv1x1/sign_S", "has entries other than -1 / +1")
            if np.any(np.triu(np.asarray(state[pre + "inv1x1/L"]), 1) != 0):
                bad(pre + "inv1x1/L", "is not lower triangular")
            if np.any(np.tril(np.asarray(state[pre + "inv1x1/U"]), 0) != 0):
                bad(pre + "inv1x1/U", "is not strictly upper triangular")
            if not np.isfinite(np.asarray(state[pre + "inv1x1/log_S"])).all():
                bad(pre + "inv1x1/log_S", "is not finite")
            for bn in ("bn1", "bn2"):
                var, mean = np.asarray(state[pre + "nn/%s/var" % bn]), np.asarray(state[pre + "nn/%s/mean" % bn])
                if not (var > 0).all():
                    bad(pre + "nn/%s/var" % bn, "has non-positive moving variances")
                if reference_batchnorm and not (np.all(var == 1.0) and np.all(mean == 0.0)):
                    bad(pre + "nn/%s/{mean,var}" % bn, "are not the untouched moving statistics (mean 0, variance 1) the reference leaves behind "
                                                      "(it never runs BatchNormalization in training mode; reference_batchnorm=False skips this test)")
    del eye


def state_dict_from_checkpoint(prefix, cfg, order=None, check_values=True, reference_batchnorm=True):
    """Read ``variables/<i>/.ATTRIBUTES/VARIABLE_VALUE`` of the checkpoint ``prefix`` (e.g. ``tf_ckpts/ckpt-21``) into this
    repository's ``{name: ndarray}``, checking count and every shape against ``order`` (default: ``variable_order(cfg)``), then
    the value-level invariants of a reference checkpoint (``check_value_invariants``; ``check_values=False`` skips them,
    ``reference_batchnorm=False`` accepts trained moving statistics).  Optimizer slots in the checkpoint are not imported."""
    order = list(order) if order is not None else variable_order(cfg)
    tensors = read_bundle(prefix)
    vals = {}
    for key, a in tensors.items():
        if key.startswith("variables/") and key.endswith(VALUE_SUFFIX):
            vals[int(key[len("variables/"):-len(VALUE_SUFFIX)])] = a
    if sorted(vals) != list(range(len(order))):
        raise ValueError("checkpoint holds %d flow variables, this configuration has %d" % (len(vals), len(order)))
    shapes = _expected_shapes(cfg)
    state = {}
    for i, name in enumerate(order):
        if tuple(vals[i].shape) != shapes[name]:
            raise ValueError("variables/%d has shape %s but position %d of the variable order is %s %s: the checkpoint was written by another "
                             "configuration or the derived order is wrong (pass order=)" % (i, tuple(vals[i].shape), i, name, shapes[name]))
        state[name] = vals[i].astype(np.float32)
    if check_values:
        check_value_invariants(state, cfg, reference_batchnorm)
    return state


def save_checkpoint_bundle(prefix, state, cfg, order=None, extra=None):
    """Write ``state`` under the reference's checkpoint keys.  (The bundle carries no ``_CHECKPOINTABLE_OBJECT_GRAPH`` entry:
    ``tf.train.load_checkpoint`` reads it, ``tf.train.Checkpoint.restore`` needs the object graph TensorFlow itself writes.)"""
    order = list(order) if order is not None else variable_order(cfg)
    tensors = {"variables/%d%s" % (i, VALUE_SUFFIX): np.asarray(state[name], dtype=np.float32) for i, name in enumerate(order)}
    tensors["save_counter" + VALUE_SUFFIX] = np.asarray(1, dtype=np.int64)
    if extra:
        tensors.update(extra)
    write_bundle(prefix, tensors)
