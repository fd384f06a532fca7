"""Assembles a TensorFlow checkpoint bundle BYTE BY BYTE from the published formats -- independently of
kddcup_2020_multimodalitiesrecall_2nd_place_amd/tf_checkpoint.py (this script imports nothing from the package: its CRC, varints,
protobuf encoding and table builder are written out again here), so that a layout misreading shared by that module's reader and
writer cannot hide (VERDICT r3 item 8 / ADVICE r3).  TensorFlow itself is not installable in the build container; what is restated:

* tensorflow/core/lib/io/table_format.txt + table_builder.cc / block_builder.cc (= LevelDB's): data blocks of prefix-compressed entries
  with a restart point every 16 entries, flushed when they exceed the block size; index block (restart interval 1) whose keys are the
  SHORTEST SEPARATORS between a block's last key and the next block's first key (not keys that exist); an empty metaindex block; 5-byte
  block trailers (type 0 + masked CRC32C of block and type); 48-byte footer (metaindex handle, index handle, padding, magic).
* tensorflow/core/protobuf/tensor_bundle.proto + tensor_bundle.cc: key "" -> BundleHeaderProto {num_shards = 1, endianness = 2 (LITTLE = 0,
  omitted in proto3), version = 3 {producer = 1}}; every variable -> BundleEntryProto {dtype 1, shape 2, shard_id 3, offset 4, size 5,
  crc32c 6 (fixed32, MASKED crc of the tensor bytes)}, proto3 omitting zero fields; tensor bytes little-endian, concatenated per shard.

What this fixture has that the package's own writer never produces: two shards, a table of several data blocks (block size 256 bytes here:
the index block has more than one entry and the reader must follow handles), shared key prefixes ("bert/encoder/layer_0/..." chains with
shared > 0) across restart points, shortened separator keys in the index block, a version sub-message in the header, a rank-0 variable,
one variable of every supported dtype, and an ExponentialMovingAverage shadow name next to its raw variable.

usage:  python tests/golden/make_tf_bundle_golden.py      (writes tests/golden/tf_bundle/{model.ckpt-7.index, .data-0000?-of-00002, expected.json})
"""
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "tf_bundle")


# ---- CRC32C, bit by bit (reflected polynomial 0x82F63B78), LevelDB's mask ----
def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for byte in data:
        c ^= byte
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def masked(c: int) -> int:
    return (((c >> 15) | ((c << 17) & 0xFFFFFFFF)) + 0xA282EAD8) & 0xFFFFFFFF


assert crc32c(b"123456789") == 0xE3069283            # the check value of CRC-32C (RFC 3720 appendix B.4)


def varint(v: int) -> bytes:
    out = b""
    while v >= 0x80:
        out += bytes([(v & 0x7F) | 0x80])
        v >>= 7
    return out + bytes([v])


def pb_varint_field(num, v):
    return varint(num << 3 | 0) + varint(v)


def pb_bytes_field(num, b):
    return varint(num << 3 | 2) + varint(len(b)) + b


# ---- the variables: values are closed formulas, so the test can recompute them without this script ----
DT = {"float32": 1, "float64": 2, "int32": 3, "int64": 9, "bfloat16": 14, "float16": 19}


def values(name: str, shape, dtype: str) -> np.ndarray:
    n = int(np.prod(shape)) if shape else 1
    seed = sum(name.encode()) % 97
    x = (np.arange(n, dtype=np.float64) * 0.37 + seed) % 11.0 - 5.0
    if dtype in ("int32", "int64"):
        return np.round(x * 1000).astype(dtype).reshape(shape)
    if dtype == "bfloat16":                      # stored as the upper 16 bits of the float32 (values chosen exactly representable)
        f = np.round(x * 4).astype(np.float32) / 4
        return (f.view(np.uint32) >> 16).astype(np.uint16).reshape(shape)
    return x.astype(dtype).reshape(shape)


VARS = [  # (name, shape, dtype, shard)
    ("bert/embeddings/LayerNorm/beta", (8,), "float32", 0),
    ("bert/embeddings/LayerNorm/gamma", (8,), "float32", 0),
    ("bert/encoder/layer_0/attention/self/key/kernel", (8, 8), "float32", 0),
    ("bert/encoder/layer_0/attention/self/key/kernel/ExponentialMovingAverage", (8, 8), "float32", 1),
    ("bert/encoder/layer_0/attention/self/query/bias", (8,), "float32", 0),
    ("bert/encoder/layer_0/attention/self/query/kernel", (8, 8), "float64", 1),
    ("bert/encoder/layer_0/intermediate/dense/bias", (16,), "float16", 0),
    ("bert/encoder/layer_0/intermediate/dense/kernel", (8, 16), "bfloat16", 1),
    ("bert/encoder/layer_1/attention/self/key/kernel", (8, 8), "float32", 1),
    ("cls/seq_relationship/am_kernel", (8, 2), "float32", 0),
    ("global_step", (), "int64", 0),
    ("kdd_conv1/weights", (1, 8, 4, 4), "float32", 1),
    ("lengths", (5,), "int32", 0),
]
for i in range(24):      # enough keys with long shared prefixes to cross several restart points (16) and data blocks
    VARS.append(("bert/encoder/layer_2/filler/var_%02d/kernel" % i, (3,), "float32", i % 2))
VARS.sort(key=lambda v: v[0].encode())


def shortest_separator(a: bytes, b: bytes) -> bytes:
    """leveldb::BytewiseComparator::FindShortestSeparator: a key k with a <= k < b, as short as the common prefix allows."""
    n = 0
    while n < min(len(a), len(b)) and a[n] == b[n]:
        n += 1
    if n < min(len(a), len(b)) and a[n] < 0xFF and a[n] + 1 < b[n]:
        return a[:n] + bytes([a[n] + 1])
    return a


def shortest_successor(a: bytes) -> bytes:
    for i, ch in enumerate(a):
        if ch != 0xFF:
            return a[:i] + bytes([ch + 1])
    return a


class Block:
    def __init__(self, restart_interval):
        self.buf, self.restarts, self.count, self.last, self.ri = b"", [0], 0, b"", restart_interval

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count < self.ri:
            while shared < min(len(self.last), len(key)) and self.last[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self) -> bytes:
        return self.buf + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def build_table(items, block_size=256) -> bytes:
    out = b""

    def emit(block_bytes):
        nonlocal out
        handle = varint(len(out)) + varint(len(block_bytes))
        out += block_bytes + b"\x00" + struct.pack("<I", masked(crc32c(block_bytes + b"\x00")))
        return handle

    index, data, pending, last_key = Block(1), Block(16), None, b""
    for key, value in items:
        if pending is not None:                   # the previous block was flushed: its index key separates it from this key
            index.add(shortest_separator(last_key, key), pending)
            pending = None
        data.add(key, value)
        last_key = key
        if data.size() >= block_size:
            pending = emit(data.finish())
            data = Block(16)
    if data.buf:
        pending = emit(data.finish())
    if pending is not None:
        index.add(shortest_successor(last_key), pending)
    meta_handle = emit(Block(1).finish())          # empty metaindex block
    index_handle = emit(index.finish())
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    return out + footer


def main():
    os.makedirs(OUT, exist_ok=True)
    shards = [b"", b""]
    items = [(b"", pb_varint_field(1, 2) + pb_bytes_field(3, pb_varint_field(1, 1)))]      # header: num_shards 2, version {producer 1}
    expected = {}
    for name, shape, dtype, shard in VARS:
        arr = values(name, shape, dtype)
        raw = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
        tshape = b"".join(pb_bytes_field(2, pb_varint_field(1, d)) for d in shape)
        entry = pb_varint_field(1, DT[dtype]) + pb_bytes_field(2, tshape)
        if shard:
            entry += pb_varint_field(3, shard)
        if len(shards[shard]):
            entry += pb_varint_field(4, len(shards[shard]))
        entry += pb_varint_field(5, len(raw)) + varint(6 << 3 | 5) + struct.pack("<I", masked(crc32c(raw)))
        shards[shard] += raw
        items.append((name.encode(), entry))
        expected[name] = {"shape": list(shape), "dtype": dtype, "shard": shard}
    table = build_table(items)
    with open(os.path.join(OUT, "model.ckpt-7.index"), "wb") as f:
        f.write(table)
    for i, s in enumerate(shards):
        with open(os.path.join(OUT, "model.ckpt-7.data-%05d-of-00002" % i), "wb") as f:
            f.write(s)
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump({"prefix": "model.ckpt-7", "variables": expected, "value_formula": "see values() in make_tf_bundle_golden.py"}, f, indent=1, sort_keys=True)
    print("wrote %d variables, index %d bytes, shards %s bytes" % (len(VARS), len(table), [len(s) for s in shards]))


if __name__ == "__main__":
    main()
