"""TensorFlow V2-FORMAT checkpoints ("tensor bundles") without TensorFlow: what tf.compat.v1.train.Saver writes for the reference
(neural_speech_coding_module.py:548 `saver.save(sess, './check/model_bnn_ac_<id>_<save_id>.ckpt')`; cmrl.py:64-73, 332-347 restore per
scope): `<prefix>.index` - an SSTable (the LevelDB table format) mapping "" to a BundleHeaderProto and every variable name to a
BundleEntryProto (dtype, shape, shard, offset, size, masked crc32c) - and `<prefix>.data-00000-of-00001`, the tensors' raw little-endian
bytes.  read_checkpoint(prefix) -> {variable name: ndarray}; write_checkpoint(prefix, named) writes the same format (uncompressed blocks,
one shard), so that a reference-side run can restore what this package trained.

Restated from the published formats (TensorFlow is not installable in the build container and the reference ships no checkpoint):
tensorflow/core/util/tensor_bundle (BundleHeaderProto / BundleEntryProto, tensor_bundle.proto), tensorflow/core/lib/io/table (format.cc:
block handles, 48-byte footer, magic 0xdb4775248b80fb57; block.cc: prefix-compressed entries + restart array; one type byte + masked
crc32c behind every block), tensorflow/core/lib/hash/crc32c (Castagnoli polynomial, mask ((crc >> 15 | crc << 17) + 0xa282ead8)).
UNVERIFIED against a TensorFlow-written file: tests/test_host.py holds the reader to the writer, to the published constants (crc32c of
"123456789" = 0xe3069283, the footer magic) and to a hand-assembled table with key prefix compression and several blocks.
Snappy-compressed blocks (type 1) are refused: tensor bundles are written uncompressed.
"""
from __future__ import annotations

import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
# tensorflow/core/framework/types.proto: the dtypes a v1 Saver of this model can contain
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_, 4: np.uint8, 6: np.int8, 5: np.int16}
_DTYPE_CODE = {np.dtype(v): k for k, v in _DTYPES.items()}


def _crc_tables():
    t0 = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
        t0.append(c)
    tabs = [t0]
    for _ in range(7):
        prev = tabs[-1]
        tabs.append([(prev[i] >> 8) ^ t0[prev[i] & 0xff] for i in range(256)])
    return tabs


_T0, _T1, _T2, _T3, _T4, _T5, _T6, _T7 = _crc_tables()


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C (Castagnoli, reflected 0x82f63b78), eight bytes per step (a 2.8 MB checkpoint: ~0.5 s); crc: the value of the bytes
    before `data` (crc32c(a + b) == crc32c(b, crc32c(a)))."""
    c = crc ^ 0xffffffff
    n8 = len(data) // 8 * 8
    for lo, hi in struct.iter_unpack("<II", memoryview(data)[:n8]):
        lo ^= c
        c = (_T7[lo & 0xff] ^ _T6[(lo >> 8) & 0xff] ^ _T5[(lo >> 16) & 0xff] ^ _T4[lo >> 24] ^
             _T3[hi & 0xff] ^ _T2[(hi >> 8) & 0xff] ^ _T1[(hi >> 16) & 0xff] ^ _T0[hi >> 24])
    for b_ in memoryview(data)[n8:]:
        c = _T0[(c ^ b_) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def mask_crc(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + _MASK_DELTA) & 0xffffffff


def unmask_crc(masked: int) -> int:
    rot = (masked - _MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- protobuf wire format, the few pieces needed ----
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf):
    """(field number, wire type, value) of a serialized message; value = int (varint / fixed) or bytes (length-delimited)."""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"tf_checkpoint: unsupported protobuf wire type {wt}")
        yield fn, wt, v


def _parse_shape(buf):
    """TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }."""
    dims = []
    for fn, _, v in _fields(buf):
        if fn == 2:
            size = 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    size = v2
            dims.append(size)
    return tuple(dims)


def _parse_entry(buf):
    """BundleEntryProto: dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6 (fixed32), slices = 7."""
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for fn, _, v in _fields(buf):
        if fn == 1:
            e["dtype"] = v
        elif fn == 2:
            e["shape"] = _parse_shape(v)
        elif fn == 3:
            e["shard_id"] = v
        elif fn == 4:
            e["offset"] = v
        elif fn == 5:
            e["size"] = v
        elif fn == 6:
            e["crc32c"] = v
        elif fn == 7:
            e["sliced"] = True
    return e


def _entry_bytes(dtype_code, shape, offset, size, crc_masked):
    shp = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    out = b"\x08" + _put_varint(dtype_code) + b"\x12" + _put_varint(len(shp)) + shp
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc_masked)
    return out


# ---- the table (SSTable) ----
def _read_block(f, offset, size):
    f.seek(offset)
    raw = f.read(size + 5)
    body, ctype, crc = raw[:size], raw[size], struct.unpack_from("<I", raw, size + 1)[0]
    if unmask_crc(crc) != crc32c(raw[:size + 1]):
        raise ValueError("tf_checkpoint: block checksum mismatch in the .index file")
    if ctype != 0:
        raise ValueError("tf_checkpoint: compressed table blocks (type %d) are not supported" % ctype)
    return body


def _block_entries(body):
    nrestart = struct.unpack_from("<I", body, len(body) - 4)[0]
    end = len(body) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(body, pos)
        non_shared, pos = _varint(body, pos)
        vlen, pos = _varint(body, pos)
        key = key[:shared] + bytes(body[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(body[pos:pos + vlen])
        pos += vlen


def _handle(buf, pos=0):
    off, pos = _varint(buf, pos)
    size, pos = _varint(buf, pos)
    return off, size, pos


def read_index(path):
    """{key (bytes): value (bytes)} of an SSTable file, in key order."""
    out = {}
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        n = f.tell()
        if n < 48:
            raise ValueError("tf_checkpoint: %s is too short to be a table" % path)
        f.seek(n - 48)
        footer = f.read(48)
        if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
            raise ValueError("tf_checkpoint: %s does not end in the table magic number" % path)
        _, _, pos = _handle(footer)                      # metaindex handle (unused)
        ioff, isize, _ = _handle(footer, pos)
        for _, hv in _block_entries(_read_block(f, ioff, isize)):
            boff, bsize, _ = _handle(hv)
            for k, v in _block_entries(_read_block(f, boff, bsize)):
                out[k] = v
    return out


def read_checkpoint(prefix, names=None):
    """{variable name: ndarray} of the checkpoint `<prefix>.index` + `<prefix>.data-?????-of-?????`; names: optional filter
    (callable or container).  Checks every tensor's crc32c."""
    table = read_index(prefix + ".index")
    header = table.get(b"")
    num_shards = 1
    if header is not None:
        for fn, _, v in _fields(header):
            if fn == 1:
                num_shards = v
            elif fn == 2 and v != 0:
                raise ValueError("tf_checkpoint: big-endian bundles are not supported")
    shards = {}
    out = {}
    for k, v in table.items():
        if k == b"":
            continue
        name = k.decode("utf-8")
        if names is not None and not (names(name) if callable(names) else name in names):
            continue
        e = _parse_entry(v)
        if e["sliced"]:
            raise ValueError("tf_checkpoint: %s is a partitioned variable (slices): not supported" % name)
        if e["dtype"] not in _DTYPES:
            continue                                     # (string tensors such as _CHECKPOINTABLE_OBJECT_GRAPH: not variables of the model)
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), "rb")
        f = shards[sid]
        f.seek(e["offset"])
        raw = f.read(e["size"])
        if e["crc32c"] is not None and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise ValueError("tf_checkpoint: checksum mismatch in tensor %s" % name)
        out[name] = np.frombuffer(raw, dtype=np.dtype(_DTYPES[e["dtype"]]).newbyteorder("<")).reshape(e["shape"]).copy()
    for f in shards.values():
        f.close()
    return out


def _block(entries, restart_interval=16):
    """A table block of (key, value) pairs in key order, prefix-compressed with a restart point every `restart_interval` keys."""
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _emit_block(f, body):
    off = f.tell()
    f.write(body + b"\x00" + struct.pack("<I", mask_crc(crc32c(body + b"\x00"))))
    return _put_varint(off) + _put_varint(len(body))


def write_index(path, items, block_bytes=4096):
    """SSTable of {key bytes: value bytes}: data blocks of ~block_bytes, an (empty) metaindex block, an index block, the footer."""
    keys = sorted(items)
    with open(path, "wb") as f:
        index, cur, cur_size = [], [], 0
        for k in keys:
            cur.append((k, items[k]))
            cur_size += len(k) + len(items[k]) + 3
            if cur_size >= block_bytes:
                index.append((cur[-1][0], _emit_block(f, _block(cur))))
                cur, cur_size = [], 0
        if cur:
            index.append((cur[-1][0], _emit_block(f, _block(cur))))
        meta = _emit_block(f, _block([]))
        idx = _emit_block(f, _block(index, restart_interval=1))
        footer = meta + idx
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))


def write_checkpoint(prefix, named):
    """Write {variable name: array} as `<prefix>.index` + `<prefix>.data-00000-of-00001` (one shard, little-endian, uncompressed)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = {b"": b"\x08\x01" + b"\x1a\x02\x08\x01"}       # BundleHeaderProto: num_shards = 1, version { producer = 1 }
    off = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(named, key=lambda s: s.encode("utf-8")):
            a = np.asarray(named[name])                     # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DTYPE_CODE:
                raise ValueError("tf_checkpoint: dtype %s of %s is not supported" % (a.dtype, name))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes(order="C")
            f.write(raw)
            items[name.encode("utf-8")] = _entry_bytes(_DTYPE_CODE[a.dtype], a.shape, off, len(raw), mask_crc(crc32c(raw)))
            off += len(raw)
    write_index(prefix + ".index", items)
