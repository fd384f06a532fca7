"""Dependency-free reader (and minimal writer) for TensorFlow checkpoint BUNDLES -- the ``model.ckpt-N.index`` +
``model.ckpt-N.data-00000-of-00001`` pair the zk / lds predict drivers restore from
(code/imagebert_zk/evaluate_normal.py:204-212: ``tf.train.Saver(ema.variables_to_restore()).restore(sess, ckpt)``;
code/imagebert_lds/src/run_pretraining_predict_score.py:558-563: ``saver.restore(sess, init_checkpoint)``).

TensorFlow is not available where this library runs, so the on-disk format is restated here from its published layout:

* ``<prefix>.index`` is an immutable sorted string table in the LevelDB table format (tensorflow/core/lib/io/table_format.txt):
  a sequence of blocks, each followed by a 5-byte trailer (1 byte compression type, 4 bytes masked CRC32C of block + type);
  a block is a run of prefix-compressed entries ``varint32 shared | varint32 non_shared | varint32 value_len | key suffix | value``
  followed by the array of restart offsets (uint32 LE) and their count; the 48-byte footer holds two BlockHandles
  (``varint64 offset, varint64 size``: metaindex block, index block), zero padding to 40 bytes and the magic
  0xdb4775248b80fb57 (LE).  The index block maps a key >= the last key of each data block to that block's handle.
* keys are variable names; the empty key holds a ``BundleHeaderProto`` (1: num_shards, 2: endianness, 3: version) and every
  other key a ``BundleEntryProto`` (1: dtype, 2: TensorShapeProto {2: dim {1: size}}, 3: shard_id, 4: offset, 5: size,
  6: fixed32 masked crc32c of the tensor bytes, 7: slices) -- tensorflow/core/protobuf/tensor_bundle.proto.
* ``<prefix>.data-SSSSS-of-NNNNN`` holds the raw little-endian tensor bytes at [offset, offset + size) of shard SSSSS.

Supported: uncompressed blocks (what BundleWriter emits), unsliced DT_FLOAT / DT_DOUBLE / DT_INT32 / DT_INT64 / DT_HALF /
DT_BFLOAT16 variables, any number of shards.  Snappy-compressed tables, partitioned (sliced) variables and string tensors raise.

``BundleReader`` offers ``has_tensor`` / ``get_tensor`` like ``tf.train.load_checkpoint``'s reader, so
``weights.from_tf_variables(cfg, BundleReader(prefix))`` is the whole importer (``weights.from_tf_checkpoint``).
``write_bundle`` produces a bundle this reader (and TensorFlow's) can load; the tests use it to make their own fixtures.
"""
from __future__ import annotations

import os
import struct

import numpy as np

MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8"), 19: np.dtype("<f2"), 14: np.dtype("<u2")}
_DT_OF = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9, np.dtype("float16"): 19}
DT_BFLOAT16 = 14


class BundleError(ValueError):
    pass


# ------------------------------------------------------------------------------------------------------------------
# CRC32C (Castagnoli), table driven, and LevelDB's mask
# ------------------------------------------------------------------------------------------------------------------
def _crc_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t[i] = c
    return t


_TABLE = _crc_table()
_TABLE_L = [int(x) for x in _TABLE]


def _crc32c_serial(data) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE_L[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _gf2_times(mat, vec):
    out, i = 0, 0
    while vec:
        if vec & 1:
            out ^= mat[i]
        vec >>= 1
        i += 1
    return out


def _zeros_operator(nbytes: int):
    """32x32 GF(2) matrix (list of columns) that advances a CRC32C register over ``nbytes`` zero bytes (zlib's crc32_combine idea)."""
    m = [0x82F63B78] + [1 << n for n in range(31)]          # one zero BIT
    out, bits = None, nbytes * 8
    while bits:
        if bits & 1:
            out = m if out is None else [_gf2_times(m, col) for col in out]
        m = [_gf2_times(m, col) for col in m]
        bits >>= 1
    return out if out is not None else [1 << n for n in range(32)]


def crc32c(data) -> int:
    """CRC32C of a bytes-like object.  Long inputs are cut into 4096-byte chunks whose registers advance in lock step as one numpy
    vector (a pure-Python byte loop does ~10 MB/s), and the chunk CRCs are chained with crc(A || B) = zeros_|B|(crc(A)) ^ crc(B)."""
    buf = np.frombuffer(memoryview(data).cast("B"), dtype=np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
    L = 4096
    n_chunks = buf.size // L
    if n_chunks < 8:
        return _crc32c_serial(buf.tobytes())
    body = buf[:n_chunks * L].reshape(n_chunks, L)
    c = np.full(n_chunks, 0xFFFFFFFF, np.uint32)
    for j in range(L):
        c = _TABLE[(c ^ body[:, j]) & 0xFF] ^ (c >> np.uint32(8))
    c ^= np.uint32(0xFFFFFFFF)
    op = _zeros_operator(L)
    crc = int(c[0])
    for v in c[1:]:
        crc = _gf2_times(op, crc) ^ int(v)
    tail = buf[n_chunks * L:]
    if tail.size:
        crc = _gf2_times(_zeros_operator(int(tail.size)), crc) ^ _crc32c_serial(tail.tobytes())
    return crc


def mask_crc(c: int) -> int:
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------------------------
# varints / the two protobuf messages
# ------------------------------------------------------------------------------------------------------------------
def _get_varint(buf, pos):
    out = shift = 0
    while True:
        if pos >= len(buf):
            raise BundleError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise BundleError("varint too long")


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_fields(buf):
    """-> list of (field number, wire type, value) for wire types 0 (varint), 1 (fixed64), 2 (bytes), 5 (fixed32)."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise BundleError("unsupported protobuf wire type %d" % wt)
        out.append((f, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": 0}
    for f, _wt, v in _parse_fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            for f2, _w2, v2 in _parse_fields(v):
                if f2 == 2:      # dim
                    size = 0
                    for f3, _w3, v3 in _parse_fields(v2):
                        if f3 == 1:
                            size = _signed64(v3)
                    e["shape"].append(size)
                elif f2 == 3 and v2:
                    raise BundleError("tensor of unknown rank")
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:
            e["slices"] += 1
    return e


def _entry_proto(dtype, shape, shard_id, offset, size, crc):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    out = b"\x08" + _put_varint(dtype) + b"\x12" + _put_varint(len(dims)) + dims
    if shard_id:
        out += b"\x18" + _put_varint(shard_id)
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return out


# ------------------------------------------------------------------------------------------------------------------
# table (SSTable) reading
# ------------------------------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify=True):
    if offset + size + 5 > len(buf):
        raise BundleError("block handle beyond end of file")
    body = buf[offset:offset + size]
    ctype = buf[offset + size]
    if verify:
        want = struct.unpack_from("<I", buf, offset + size + 1)[0]
        if mask_crc(crc32c(bytes(buf[offset:offset + size + 1]))) != want:
            raise BundleError("block checksum mismatch at offset %d" % offset)
    if ctype != 0:
        raise BundleError("compressed table block (type %d): only uncompressed bundle indexes are supported" % ctype)
    return body


def _block_entries(block):
    if len(block) < 4:
        raise BundleError("block too small")
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    if limit < 0:
        raise BundleError("bad restart array")
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise BundleError("corrupt block entry")
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True) -> dict:
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    buf = memoryview(open(path, "rb").read())
    if len(buf) < 48:
        raise BundleError("%s: too small for a table footer" % path)
    footer = buf[len(buf) - 48:]
    if struct.unpack_from("<Q", footer, 40)[0] != MAGIC:
        raise BundleError("%s: not a table file (bad magic)" % path)
    pos = 0
    _mo, pos = _get_varint(footer, pos)
    _ms, pos = _get_varint(footer, pos)
    io, pos = _get_varint(footer, pos)
    isz, pos = _get_varint(footer, pos)
    out = {}
    for _k, handle in _block_entries(_read_block(buf, io, isz, verify)):
        off, p = _get_varint(handle, 0)
        size, p = _get_varint(handle, p)
        for k, v in _block_entries(_read_block(buf, off, size, verify)):
            out[k] = v
    return out


# ------------------------------------------------------------------------------------------------------------------
# bundle reader
# ------------------------------------------------------------------------------------------------------------------
class BundleReader:
    """``tf.train.load_checkpoint(prefix)``-like access: ``has_tensor``, ``get_tensor``, ``get_variable_to_shape_map``."""

    def __init__(self, prefix: str, verify_tensors: bool = False):
        self.prefix = prefix
        idx = prefix + ".index"
        if not os.path.exists(idx):
            raise FileNotFoundError(idx)
        tab = read_table(idx)
        if b"" not in tab:
            raise BundleError("%s: no bundle header entry" % idx)
        self.num_shards, endian = 1, 0
        for f, _wt, v in _parse_fields(tab[b""]):
            if f == 1:
                self.num_shards = v
            elif f == 2:
                endian = v
        if endian != 0:
            raise BundleError("big-endian bundle")
        self.entries = {k.decode("utf-8"): _parse_entry(v) for k, v in tab.items() if k != b""}
        self.verify_tensors = verify_tensors
        self._maps = {}

    def has_tensor(self, name):
        return name in self.entries

    def get_variable_to_shape_map(self):
        return {k: list(e["shape"]) for k, e in self.entries.items()}

    def _shard(self, i):
        if i not in self._maps:
            p = "%s.data-%05d-of-%05d" % (self.prefix, i, self.num_shards)
            self._maps[i] = np.memmap(p, dtype=np.uint8, mode="r")
        return self._maps[i]

    def get_tensor(self, name):
        e = self.entries.get(name)
        if e is None:
            raise KeyError(name)
        if e["slices"]:
            raise BundleError("%s is a partitioned (sliced) variable: not supported" % name)
        dt = _DTYPES.get(e["dtype"])
        if dt is None:
            raise BundleError("%s: unsupported dtype enum %d" % (name, e["dtype"]))
        n = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if n * dt.itemsize != e["size"]:
            raise BundleError("%s: %d bytes stored, shape %s needs %d" % (name, e["size"], e["shape"], n * dt.itemsize))
        shard = self._shard(e["shard_id"])
        if e["offset"] + e["size"] > shard.shape[0]:
            raise BundleError("%s: data beyond end of shard %d" % (name, e["shard_id"]))
        raw = np.asarray(shard[e["offset"]:e["offset"] + e["size"]])
        if self.verify_tensors and e["crc32c"] is not None and mask_crc(crc32c(raw.tobytes())) != e["crc32c"]:
            raise BundleError("%s: tensor checksum mismatch" % name)
        a = raw.view(dt).reshape(e["shape"])
        if e["dtype"] == DT_BFLOAT16:
            a = (a.astype(np.uint32) << 16).view(np.float32)
        return np.array(a)


# ------------------------------------------------------------------------------------------------------------------
# minimal writer (one shard, uncompressed; restart interval 16 like TensorFlow's table builder)
# ------------------------------------------------------------------------------------------------------------------
def _build_block(items, restart_interval=16):
    out, restarts, last, i = bytearray(), [], b"", 0
    for k, v in items:
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
        i += 1
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_bundle(prefix: str, tensors: dict, block_bytes: int = 4096):
    """{name: array} -> ``prefix.index`` + ``prefix.data-00000-of-00001``."""
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    entries, off = [], 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for n in names:
            a = np.asarray(tensors[n], order="C")          # (ascontiguousarray would turn a scalar into shape [1])
            if a.dtype not in _DT_OF:
                raise BundleError("%s: dtype %s not supported by the writer" % (n, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
            f.write(raw)
            entries.append((n.encode("utf-8"), _entry_proto(_DT_OF[a.dtype], a.shape, 0, off, len(raw), mask_crc(crc32c(raw)))))
            off += len(raw)
    items = [(b"", b"\x08\x01\x1a\x02\x08\x01")] + entries     # header: num_shards 1, little endian, version {producer 1}
    out = bytearray()
    index_items = []

    def flush(block_items):
        blk = _build_block(block_items)
        handle = _put_varint(len(out)) + _put_varint(len(blk))
        out.extend(blk)
        out.append(0)
        out.extend(struct.pack("<I", mask_crc(crc32c(blk + b"\x00"))))
        index_items.append((block_items[-1][0], handle))

    cur, size = [], 0
    for k, v in items:
        cur.append((k, v))
        size += len(k) + len(v) + 3
        if size >= block_bytes:
            flush(cur)
            cur, size = [], 0
    if cur:
        flush(cur)
    meta = _build_block([])
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta))
    out.extend(meta); out.append(0); out.extend(struct.pack("<I", mask_crc(crc32c(meta + b"\x00"))))
    idx = _build_block(index_items, restart_interval=1)
    idx_handle = _put_varint(len(out)) + _put_varint(len(idx))
    out.extend(idx); out.append(0); out.extend(struct.pack("<I", mask_crc(crc32c(idx + b"\x00"))))
    footer = meta_handle + idx_handle
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
