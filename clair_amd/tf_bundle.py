"""Reader (and minimal writer) for TensorFlow "tensor bundle" checkpoints, without TensorFlow.

The reference saves/restores its weights with ``tf.train.Saver`` (/root/reference/clair/model.py:712,
1016-1020); a checkpoint is ``prefix.index`` + ``prefix.data-00000-of-00001`` (README.md:231,
clair/callVarBam.py:72 checks ``prefix.meta``).  The published Clair models (README.md:94-110) are
only downloadable, so this reader cannot be tested against a real file here: the format below is the
published one (LevelDB table + BundleEntryProto), exercised by a round trip through the writer in
tests/test_weights.py, and the variable names come from clair_amd/weights.py:tf_variable_names().

.index   LevelDB SSTable, uncompressed blocks: key "" -> BundleHeaderProto, key <variable name> ->
         BundleEntryProto {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset, 5: size, 6: crc32c}
.data-*  raw little-endian tensor bytes at [offset, offset+size)
"""
import os
import struct
from collections import OrderedDict

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DT_FLOAT = 1


# ---- varint / protobuf helpers ---------------------------------------------------------------
def _get_varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_message(buf):
    """-> list of (field number, wire type, value) for the wire types a bundle uses."""
    fields, pos = [], 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _get_varint(buf, pos)
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            val, pos = bytes(buf[pos:pos + ln]), pos + ln
        elif wt == 5:
            val, pos = struct.unpack_from("<I", buf, pos)[0], pos + 4
        elif wt == 1:
            val, pos = struct.unpack_from("<Q", buf, pos)[0], pos + 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        fields.append((num, wt, val))
    return fields


def _parse_entry(buf):
    entry = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0)
    for num, _, val in _parse_message(buf):
        if num == 1:
            entry["dtype"] = val
        elif num == 2:
            dims = []
            for n2, _, v2 in _parse_message(val):
                if n2 == 2:   # TensorShapeProto.dim
                    size = 0
                    for n3, _, v3 in _parse_message(v2):
                        if n3 == 1:
                            size = v3
                    dims.append(size)
            entry["shape"] = tuple(dims)
        elif num == 3:
            entry["shard_id"] = val
        elif num == 4:
            entry["offset"] = val
        elif num == 5:
            entry["size"] = val
    return entry


# ---- LevelDB table -------------------------------------------------------------------------------
def _block_entries(block):
    """Decode one uncompressed block (prefix-compressed keys + restart array)."""
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _read_block(data, offset, size):
    if data[offset + size] != 0:
        raise ValueError("compressed table blocks are not supported (TF writes bundle indexes uncompressed)")
    return data[offset:offset + size]


def read_index(path):
    """-> OrderedDict name -> entry dict, from a ``.index`` file."""
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError("%s is not a tensor-bundle index (bad table magic)" % path)
    footer = data[-48:]
    pos = 0
    _, pos = _get_varint(footer, pos)          # metaindex handle
    _, pos = _get_varint(footer, pos)
    idx_off, pos = _get_varint(footer, pos)
    idx_size, pos = _get_varint(footer, pos)
    entries = OrderedDict()
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size)):
        off, p = _get_varint(handle, 0)
        size, _ = _get_varint(handle, p)
        for key, value in _block_entries(_read_block(data, off, size)):
            if key:                             # key "" is the BundleHeaderProto
                entries[key.decode()] = _parse_entry(value)
    return entries


def read_tensors(prefix):
    """All float32 tensors of a checkpoint prefix -> OrderedDict name -> ndarray."""
    entries = read_index(prefix + ".index")
    shards = {}
    out = OrderedDict()
    for name, e in entries.items():
        if e["dtype"] != DT_FLOAT:
            continue
        if e["shard_id"] not in shards:
            n_shards = 1 + max(x["shard_id"] for x in entries.values())
            shards[e["shard_id"]] = np.memmap("%s.data-%05d-of-%05d" % (prefix, e["shard_id"], n_shards), dtype=np.uint8, mode="r")
        raw = shards[e["shard_id"]][e["offset"]:e["offset"] + e["size"]]
        out[name] = np.frombuffer(bytes(raw), dtype="<f4").reshape(e["shape"]).copy()
    return out


def load_checkpoint(prefix):
    """Checkpoint prefix -> weight dict keyed as clair_amd.weights.TENSOR_TABLE."""
    from clair_amd import weights
    tensors = read_tensors(prefix)
    names = weights.tf_variable_names()
    w = OrderedDict((k, np.zeros(shape, dtype=np.float32)) for k, shape in weights.TENSOR_TABLE.items())
    seen = set()
    for tf_name, (key, index) in names.items():
        if tf_name not in tensors:
            raise KeyError("variable %s not found in checkpoint %s (has: %s ...)"
                           % (tf_name, prefix, ", ".join(list(tensors)[:4])))
        if index is None:
            w[key][...] = tensors[tf_name].reshape(w[key].shape)
        else:
            w[key][index] = tensors[tf_name].reshape(w[key][index].shape)
        seen.add(key)
    weights.check_weights(w)
    return w


# ---- minimal writer (tests / converting an .npz container back into the reference's format) -----
_CRC_TABLE = None


def _crc32c(data):
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tbl = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tbl.append(c)
        _CRC_TABLE = tbl
    crc = 0xFFFFFFFF
    for b in data:
        crc = _CRC_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked_crc(data):
    crc = _crc32c(data)
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


def _build_block(items, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (key, value) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        prev = key
    for r in restarts or [0]:
        out += struct.pack("<I", r)
    out += struct.pack("<I", max(1, len(restarts)))
    return bytes(out)


def _shape_proto(shape):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    return dims


def write_checkpoint(prefix, tensors, entries_per_block=64):
    """Write name -> float32 ndarray as prefix.index + prefix.data-00000-of-00001."""
    names = sorted(tensors)
    blob, items = bytearray(), [(b"", b"\x08\x01\x1a\x02\x08\x01")]   # header: num_shards = 1, version.producer = 1
    for name in names:
        a = np.ascontiguousarray(tensors[name], dtype="<f4")
        raw = a.tobytes()
        shape = _shape_proto(a.shape)
        entry = (b"\x08" + _put_varint(DT_FLOAT) + b"\x12" + _put_varint(len(shape)) + shape
                 + b"\x20" + _put_varint(len(blob)) + b"\x28" + _put_varint(len(raw))
                 + b"\x35" + struct.pack("<I", _masked_crc(raw) if len(raw) < (1 << 16) else 0))
        items.append((name.encode(), entry))
        blob += raw
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(blob))
    out, index_items = bytearray(), []

    def emit(block):
        off = len(out)
        out.extend(block + b"\x00" + struct.pack("<I", _masked_crc(block + b"\x00")))
        return _put_varint(off) + _put_varint(len(block))

    for i in range(0, len(items), entries_per_block):
        chunk = items[i:i + entries_per_block]
        index_items.append((chunk[-1][0], emit(_build_block(chunk))))
    meta_handle = emit(_build_block([]))
    index_handle = emit(_build_block(index_items, restart_interval=1))
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    return prefix


def export_checkpoint(prefix, w):
    """Write a weight dict (TENSOR_TABLE keys) under the reference's TF variable names."""
    from clair_amd import weights
    tensors = {}
    for tf_name, (key, index) in weights.tf_variable_names().items():
        tensors[tf_name] = w[key] if index is None else w[key][index]
    return write_checkpoint(prefix, tensors)
