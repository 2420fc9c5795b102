"""Reader (and minimal writer) for TensorFlow "tensor bundle" checkpoints, without TensorFlow.

The reference saves/restores its weights with ``tf.train.Saver`` (/root/reference/clair/model.py:712,
1016-1020); a checkpoint is ``prefix.index`` + ``prefix.data-00000-of-00001`` (README.md:231,
clair/callVarBam.py:72 checks ``prefix.meta``).  The published Clair models (README.md:94-110) are
only downloadable, so this reader cannot be tested against a real file here: the format below is the
published one (LevelDB table + BundleEntryProto).  It is exercised (tests/test_weights.py) by a round trip through
the writer below AND by a small bundle assembled byte by byte from the format description by an independent script
(tools/make_tf_bundle_fixture.py -> tests/golden/tf_bundle_small.*: snappy-compressed and plain blocks, block and
tensor CRCs, prefix-compressed keys over several restart points, optimizer slots and non-float variables to skip,
a variable saved as /part_N pieces).  The variable names come from clair_amd/weights.py:tf_variable_names() and are
[TF-recall]: when a real checkpoint names them differently, load_checkpoint lists what the file holds and a JSON
override (prefix + ".names.json" or $CLAIR_AMD_TF_NAMES) maps expected -> actual names without touching code.

.index   LevelDB SSTable; blocks plain (type 0) or snappy (type 1), each followed by type byte + masked CRC32C:
         key "" -> BundleHeaderProto, key <variable name> ->
         BundleEntryProto {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset, 5: size, 6: crc32c (masked)}
.data-*  raw little-endian tensor bytes at [offset, offset+size)
"""
import os
import struct
from collections import OrderedDict

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DT_FLOAT = 1
VERIFY_TENSOR_CRC_BELOW = 1 << 26      # tensors up to 64 MiB have their CRC32C checked on load (pure-Python CRC: ~10 MB/s)


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
        elif num == 6:
            entry["crc32c"] = val
        elif num == 7:
            entry["slices"] = entry.get("slices", 0) + 1
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


def snappy_decompress(buf):
    """Raw snappy block format (format_description.txt): varint uncompressed length, then literal / copy elements."""
    want, pos = _get_varint(buf, 0)
    out = bytearray()
    n = len(buf)
    while pos < n:
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            if pos + ln > n:
                raise ValueError("snappy: literal runs past the end of the block")
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln, off = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln, off = (tag >> 2) + 1, int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:
            ln, off = (tag >> 2) + 1, int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("snappy: copy offset %d outside the %d bytes produced so far" % (off, len(out)))
        for _ in range(ln):            # byte by byte: copies may overlap their own output
            out.append(out[-off])
    if len(out) != want:
        raise ValueError("snappy: block decompressed to %d bytes, header says %d" % (len(out), want))
    return bytes(out)


def _read_block(data, offset, size):
    """Block contents at [offset, offset+size): trailer = 1 type byte (0 plain, 1 snappy) + masked CRC32C of contents + type."""
    if offset + size + 5 > len(data):
        raise ValueError("table block [%d, +%d) runs past the end of the file" % (offset, size))
    raw, kind = data[offset:offset + size], data[offset + size]
    stored = struct.unpack_from("<I", data, offset + size + 1)[0]
    if stored != _masked_crc(data[offset:offset + size + 1]):
        raise ValueError("table block at %d: CRC32C mismatch (corrupt index file)" % offset)
    if kind == 0:
        return raw
    if kind == 1:
        return snappy_decompress(raw)
    raise ValueError("table block at %d has unknown compression type %d" % (offset, kind))


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


def read_tensors(prefix, verify_crc=True):
    """All float32 tensors of a checkpoint prefix -> OrderedDict name -> ndarray (optimizer slots and counters included;
    non-float variables such as global_step are skipped)."""
    entries = read_index(prefix + ".index")
    shards = {}
    out = OrderedDict()
    for name, e in entries.items():
        if e["dtype"] != DT_FLOAT:
            continue
        if e["shard_id"] not in shards:
            n_shards = 1 + max(x["shard_id"] for x in entries.values())
            shards[e["shard_id"]] = np.memmap("%s.data-%05d-of-%05d" % (prefix, e["shard_id"], n_shards), dtype=np.uint8, mode="r")
        if e.get("slices"):
            raise ValueError("variable %s is stored as TensorSlice pieces (a partitioned variable saved through SaveSlice); "
                             "this reader handles whole tensors and /part_N variables only" % name)
        raw = bytes(shards[e["shard_id"]][e["offset"]:e["offset"] + e["size"]])
        if len(raw) != e["size"] or e["size"] != 4 * int(np.prod(e["shape"], dtype=np.int64)):
            raise ValueError("variable %s: %d bytes on disk for shape %s" % (name, len(raw), (e["shape"],)))
        if e.get("crc32c") and e["size"] <= VERIFY_TENSOR_CRC_BELOW and verify_crc and e["crc32c"] != _masked_crc(raw):
            raise ValueError("variable %s: CRC32C mismatch (corrupt data file)" % name)
        out[name] = np.frombuffer(raw, dtype="<f4").reshape(e["shape"]).copy()
    # a variable created under a partitioner is saved piecewise as <name>/part_0 .. part_k (split along axis 0): join them
    parts = {}
    for name in list(out):
        head, sep, tail = name.rpartition("/part_")
        if sep and tail.isdigit() and head not in out:
            parts.setdefault(head, {})[int(tail)] = name
    for head, pieces in parts.items():
        if sorted(pieces) == list(range(len(pieces))):
            out[head] = np.concatenate([out.pop(pieces[i]) for i in range(len(pieces))], axis=0)
    return out


def _name_overrides(prefix):
    """{"rename": {expected: actual}, "rename_prefix": {expected_prefix: actual_prefix}} from prefix.names.json or
    $CLAIR_AMD_TF_NAMES -- for a checkpoint whose variable names differ from the table in clair_amd/weights.py."""
    import json
    path = os.environ.get("CLAIR_AMD_TF_NAMES") or (prefix + ".names.json")
    if not os.path.isfile(path):
        return {}, {}, None
    with open(path) as f:
        doc = json.load(f)
    unknown = set(doc) - {"rename", "rename_prefix"}
    if unknown:
        raise ValueError("%s: unknown keys %s (expected \"rename\" and / or \"rename_prefix\")" % (path, sorted(unknown)))
    return dict(doc.get("rename", {})), dict(doc.get("rename_prefix", {})), path


def candidate_names(tf_name, rename, rename_prefix):
    """Names under which `tf_name` may be stored, most specific first: explicit override, prefix override, the table's
    own name, and for the LSTM variables the same canonical name inside the CudnnLSTM layer's own scope ("cudnn_lstm"),
    where a GPU-trained graph may have created its saveable [TF-recall]."""
    out = []
    if tf_name in rename:
        out.append(rename[tf_name])
    for old, new in sorted(rename_prefix.items(), key=lambda kv: -len(kv[0])):
        if tf_name.startswith(old):
            out.append(new + tf_name[len(old):])
    out.append(tf_name)
    if tf_name.startswith("LSTM"):
        head, _, tail = tf_name.partition("/")
        out.append("%s/cudnn_lstm/%s" % (head, tail))
    seen, uniq = set(), []
    for n in out:
        if n not in seen:
            seen.add(n)
            uniq.append(n)
    return uniq


def cudnn_opaque_to_canonical(opaque, input_size, units):
    """One bidirectional CudnnLSTM layer's flat parameter buffer -> {(direction, "kernel"|"bias"): array} in the canonical
    CudnnCompatibleLSTMCell form (kernel [input + units, 4 units], bias [4 units], gate columns i | c~ | f | o).

    A GPU-trained Clair graph holds each LSTM layer as tf.contrib.cudnn_rnn.CudnnLSTM (/root/reference/clair/model.py:281-296), whose
    only variable is the flat "opaque_kernel".  tf.train.Saver normally stores it through the layer's CudnnLSTMSaveable in the
    canonical per-direction form (the names candidate_names() tries inside "cudnn_lstm/"); a checkpoint written without the
    saveable carries the buffer itself.  Layout [TF-recall: cuDNN's documented canonical order, as cudnn_rnn_ops converts it]: all
    weight matrices first -- per direction (fw, bw): W_i, W_f, W_c, W_o [units, input_size] then R_i, R_f, R_c, R_o [units, units],
    row-major -- then all biases -- per direction: bW_i, bW_f, bW_c, bW_o, bR_i, bR_f, bR_c, bR_o [units].  The canonical kernel
    is [W^T ; R^T] with the gate blocks reordered i, c, f, o; the canonical bias is bW + bR (forget bias 0 in both forms)."""
    opaque = np.asarray(opaque, dtype=np.float32).ravel()
    per_dir_w = 4 * units * (input_size + units)
    if opaque.size != 2 * per_dir_w + 2 * 8 * units:
        raise ValueError("opaque CudnnLSTM buffer has %d floats; a bidirectional layer with input %d and %d units has %d"
                         % (opaque.size, input_size, units, 2 * per_dir_w + 16 * units))
    out, order = {}, (0, 2, 1, 3)          # canonical block j takes cuDNN gate order[j]: i, c, f, o <- i, f, c, o
    for d, name in enumerate(("fw", "bw")):
        wbase = d * per_dir_w
        W = opaque[wbase:wbase + 4 * units * input_size].reshape(4, units, input_size)
        R = opaque[wbase + 4 * units * input_size:wbase + per_dir_w].reshape(4, units, units)
        kernel = np.empty((input_size + units, 4 * units), dtype=np.float32)
        for j, g in enumerate(order):
            kernel[:input_size, j * units:(j + 1) * units] = W[g].T
            kernel[input_size:, j * units:(j + 1) * units] = R[g].T
        b = opaque[2 * per_dir_w + d * 8 * units:2 * per_dir_w + (d + 1) * 8 * units].reshape(2, 4, units)
        out[(name, "kernel")] = kernel
        out[(name, "bias")] = np.concatenate([b[0, g] + b[1, g] for g in order])
    return out


def _expand_opaque_layers(tensors):
    """For every "<LSTMn>/.../opaque_kernel" of the right size that has no canonical twin in the file, add the canonical
    per-direction tensors under the names the table expects."""
    added = {}
    for name, a in tensors.items():
        if not name.endswith("opaque_kernel"):
            continue
        layer = name.split("/")[0]
        if layer not in ("LSTM1", "LSTM2"):
            continue
        input_size = 32 if layer == "LSTM1" else 256
        try:
            parts = cudnn_opaque_to_canonical(a, input_size, 128)
        except ValueError:
            continue
        for (d, kind), value in parts.items():
            canon = "%s/stack_bidirectional_rnn/cell_0/bidirectional_rnn/%s/cudnn_compatible_lstm_cell/%s" % (layer, d, kind)
            if canon not in tensors:
                added[canon] = value
    return added


def load_checkpoint(prefix):
    """Checkpoint prefix -> weight dict keyed as clair_amd.weights.TENSOR_TABLE."""
    from clair_amd import weights
    tensors = read_tensors(prefix)
    tensors.update(_expand_opaque_layers(tensors))
    rename, rename_prefix, override_path = _name_overrides(prefix)
    names = weights.tf_variable_names()
    w = OrderedDict((k, np.zeros(shape, dtype=np.float32)) for k, shape in weights.TENSOR_TABLE.items())
    model_vars = {n: a for n, a in tensors.items() if not (n.endswith("/Adam") or n.endswith("/Adam_1"))}

    def listing():
        rows = ["  %s %s" % (n, tuple(a.shape)) for n, a in sorted(model_vars.items())]
        more = "" if len(rows) <= 80 else "\n  ... %d more" % (len(rows) - 80)
        return ("float32 variables in %s (optimizer slots left out):\n%s%s\nTo map names, write %s.names.json "
                "{\"rename\": {expected: actual}, \"rename_prefix\": {expected_prefix: actual_prefix}} (or point "
                "$CLAIR_AMD_TF_NAMES at such a file)." % (prefix, "\n".join(rows[:80]), more, prefix))

    def shape_hints(want, tf_name):
        """Variables of the file that could BE the missing one: same shape (a [160,512] float variable is almost certainly an
        LSTM1 kernel), not already claimed by another expected name; the ready-made override line comes with them."""
        claimed = {n for t in names for n in candidate_names(t, rename, rename_prefix) if n in tensors and t != tf_name}
        hits = [n for n, a in sorted(model_vars.items()) if tuple(a.shape) == tuple(want) and n not in claimed]
        if not hits:
            return "no unclaimed variable of shape %s in the file" % (tuple(want),)
        return ("unclaimed variables of the same shape %s: %s\n  e.g. {\"rename\": {\"%s\": \"%s\"}}"
                % (tuple(want), ", ".join(hits[:6]) + (" ... (%d)" % len(hits) if len(hits) > 6 else ""), tf_name, hits[0]))

    for tf_name, (key, index) in names.items():
        tried = candidate_names(tf_name, rename, rename_prefix)
        found = next((n for n in tried if n in tensors), None)
        want = w[key].shape if index is None else w[key][index].shape
        if found is None:
            raise KeyError("variable %s not found in checkpoint %s (tried: %s%s)\n%s\n%s"
                           % (tf_name, prefix, ", ".join(tried), "; overrides from " + override_path if override_path else "",
                              shape_hints(want, tf_name), listing()))
        if int(np.prod(tensors[found].shape)) != int(np.prod(want)):
            raise ValueError("variable %s in checkpoint %s has shape %s, the graph needs %s\n%s\n%s"
                             % (found, prefix, tuple(tensors[found].shape), tuple(want), shape_hints(want, tf_name), listing()))
        if index is None:
            w[key][...] = tensors[found].reshape(want)
        else:
            w[key][index] = tensors[found].reshape(want)
    weights.check_weights(w)
    return w


# ---- minimal writer (tests / converting an .npz container back into the reference's format) -----
_CRC_TABLE = None


def _crc32c(data):
    global _CRC_TABLE
    if len(data) >= 4096:          # big ranges through the host library when it is built (1 GB/s vs 10 MB/s here)
        try:
            from clair_amd import _hostapi
            return _hostapi.crc32c(bytes(data))
        except Exception:
            pass
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
                 + b"\x35" + struct.pack("<I", _masked_crc(raw)))
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
