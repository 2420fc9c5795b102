"""A stand-in for the part of TensorFlow 1.13's Python API that tools/mint_tf_golden.py touches, in NumPy -- TEST infrastructure.

TensorFlow cannot be installed in the build image, so the script that is meant to pin this repository's oracle to the real
reference arithmetic (tools/mint_tf_golden.py) had never executed.  This module lets it execute end to end on the CPU: graph
construction in the script's own order, variable creation under TensorFlow's naming rules, the load of the recipe weights BY NAME,
`Session.run` of every fetched tensor, `tf.train.Saver().save` (a tensor bundle assembled by tools/make_tf_bundle_fixture.py's
writer -- not clair_amd/tf_bundle.py's own) and `tf.train.NewCheckpointReader`.  What it proves: the tool parses, asks TF for nothing
that does not exist in this subset, its name table matches the names TF's scoping rules produce, its weight slicing (L3/Unit_i,
L5_k) agrees with the engine's tensor ids, and the files it writes are what the three consuming tests read.  What it does NOT prove:
anything about TensorFlow's arithmetic -- every rule below is [TF-recall], the same recollection the oracle restates
(LSTMBlockCell gate order i, c~, f, o; forget_bias 0 under CudnnCompatibleLSTMCell; backward direction = reverse, run, reverse).
A file minted with this module is NEVER written into tests/golden/: parity stays "unpinned" until the real TensorFlow has run.

TF 1.13 rules restated (tensorflow/python/ops/rnn.py, tensorflow/contrib/rnn/python/ops/{rnn,lstm_ops}.py, tensorflow/python/layers/core.py):
  * tf.variable_scope(name) nests names with "/"; tf.layers.dense(name=N) creates N/kernel [in, units] and N/bias [units];
  * stack_bidirectional_dynamic_rnn: scope "stack_bidirectional_rnn", per layer i "cell_%d" % i, then bidirectional_dynamic_rnn's
    "bidirectional_rnn" with "fw" / "bw" (dynamic_rnn is handed that scope: no extra "rnn" level), then the cell's own layer name;
  * CudnnCompatibleLSTMCell(num_units) is an LSTMBlockCell(forget_bias=0, cell_clip=None, use_peephole=False) whose layer name is
    "cudnn_compatible_lstm_cell"; variables "kernel" [input + num_units, 4 num_units] and "bias" [4 num_units];
  * variables are listed by tf.global_variables() in creation order; Variable.name ends in ":0".
"""
import os
import sys
import types

import numpy as np

__version__ = "1.13.2-fake"

_STATE = {"scopes": [], "variables": []}
_MEMO = "__memo__"


def reset_default_graph():
    _STATE["scopes"], _STATE["variables"] = [], []


class _Scope(object):
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        _STATE["scopes"].append(self.name)
        return self

    def __exit__(self, *a):
        _STATE["scopes"].pop()


def variable_scope(name):
    return _Scope(name)


def _scoped(name):
    return "/".join(_STATE["scopes"] + [name])


class _DType(object):
    def __init__(self, name):
        self.name = name
        self.base_dtype = self


float32, int64 = _DType("float32"), _DType("int64")


class _Shape(object):
    def __init__(self, dims):
        self.dims = tuple(dims)

    def as_list(self):
        return list(self.dims)

    def __iter__(self):
        return iter(self.dims)

    def __len__(self):
        return len(self.dims)

    def __repr__(self):
        return repr(self.dims)


class Tensor(object):
    """A node: `fn(feed)` -> ndarray, with the static shape TF would know (None = batch)."""
    def __init__(self, fn, shape):
        self._fn, self.static = fn, tuple(shape)

    # one evaluation per Session.run: without it the 256 L3 units would each re-run both LSTM layers
    @property
    def fn(self):
        def cached(feed):
            memo = feed.setdefault(_MEMO, {})
            if id(self) not in memo:
                memo[id(self)] = self._fn(feed)
            return memo[id(self)]
        return cached

    @fn.setter
    def fn(self, f):
        self._fn = f

    def get_shape(self):
        return _Shape(self.static)

    def __ge__(self, other):
        return Tensor(lambda feed: self.fn(feed) >= np.float32(other), self.static)

    def __mul__(self, other):
        if isinstance(other, Tensor):
            return Tensor(lambda feed: (self.fn(feed) * other.fn(feed)).astype(np.float32), self.static)
        return Tensor(lambda feed: (self.fn(feed) * np.float32(other)).astype(np.float32), self.static)

    __rmul__ = __mul__


class Variable(Tensor):
    def __init__(self, name, shape, dtype=float32, value=None):
        self.name = name + ":0"
        self.shape = _Shape(shape)
        self.dtype = dtype
        self.value = np.zeros(shape, np.float32 if dtype is float32 else np.int64) if value is None else np.asarray(value)
        Tensor.__init__(self, lambda feed: self.value, shape)
        _STATE["variables"].append(self)

    def load(self, value, sess):
        value = np.asarray(value)
        assert tuple(value.shape) == tuple(self.shape.dims), (self.name, value.shape, self.shape)
        self.value = value.astype(self.value.dtype)


def global_variables():
    return list(_STATE["variables"])


def global_variables_initializer():
    return Tensor(lambda feed: None, ())


def get_variable(name, initializer=None):
    v = np.asarray(initializer)
    return Variable(_scoped(name), v.shape, float32 if v.dtype.kind == "f" else int64, v)


def placeholder(dtype, shape, name=None):
    t = Tensor(None, shape)
    t.fn = lambda feed: np.asarray(feed[t], dtype=np.float32)
    return t


class _Dim(object):
    def __init__(self, t, axis):
        self.t, self.axis = t, axis


class _ShapeOf(object):
    def __init__(self, t):
        self.t = t

    def __getitem__(self, axis):
        return _Dim(self.t, axis)


def shape(t):
    return _ShapeOf(t)


def reshape(t, new_shape):
    def fn(feed):
        return t.fn(feed).reshape([d.t.fn(feed).shape[d.axis] if isinstance(d, _Dim) else d for d in new_shape])
    return Tensor(fn, [None if isinstance(d, _Dim) else d for d in new_shape])


def transpose(t, perm=None):
    return Tensor(lambda feed: np.transpose(t.fn(feed), perm), [t.static[p] for p in perm])


def unstack(t, axis):
    n = t.static[axis]
    assert n is not None, "unstack along an unknown dimension"
    rest = [d for i, d in enumerate(t.static) if i != axis]
    return [Tensor((lambda i: lambda feed: np.take(t.fn(feed), i, axis=axis))(i), rest) for i in range(n)]


def stack(ts, axis):
    st = list(ts[0].static)
    st.insert(axis, len(ts))
    return Tensor(lambda feed: np.stack([u.fn(feed) for u in ts], axis=axis), st)


def where(cond, a, b):
    return Tensor(lambda feed: np.where(cond.fn(feed), a.fn(feed), b.fn(feed)).astype(np.float32), a.static)


def _elu(t):
    def fn(feed):
        x = t.fn(feed)
        return np.where(x > 0, x, np.expm1(np.minimum(x, 0).astype(np.float64)).astype(np.float32)).astype(np.float32)
    return Tensor(fn, t.static)


def _softmax(t):
    def fn(feed):
        x = t.fn(feed)
        e = np.exp(x - x.max(axis=-1, keepdims=True))
        return (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)
    return Tensor(fn, t.static)


def _dense(inputs, units, name, activation=None):
    with variable_scope(name):
        k = Variable(_scoped("kernel"), (inputs.static[-1], units))
        b = Variable(_scoped("bias"), (units,))
    lin = Tensor(lambda feed: (inputs.fn(feed) @ k.value + b.value).astype(np.float32), list(inputs.static[:-1]) + [units])
    return activation(lin) if activation else lin


class _CudnnCompatibleLSTMCell(object):
    layer_name = "cudnn_compatible_lstm_cell"

    def __init__(self, num_units):
        self.num_units = num_units


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(np.float32)


def _dynamic_rnn(cell, inp, reverse):
    """One direction over a time-major input; builds kernel / bias under the current scope + the cell's layer name."""
    h = cell.num_units
    with variable_scope(cell.layer_name):
        k = Variable(_scoped("kernel"), (inp.static[-1] + h, 4 * h))
        b = Variable(_scoped("bias"), (4 * h,))

    def fn(feed):
        x = inp.fn(feed)                       # [T, n, D]
        if reverse:
            x = x[::-1]
        n = x.shape[1]
        cs, hs = np.zeros((n, h), np.float32), np.zeros((n, h), np.float32)
        outs = []
        for t in range(x.shape[0]):
            z = (np.concatenate([x[t], hs], axis=1) @ k.value + b.value).astype(np.float32)
            i, ci, f, o = z[:, :h], z[:, h:2 * h], z[:, 2 * h:3 * h], z[:, 3 * h:]          # LSTMBlockCell: i, ci, f, o; forget_bias 0
            cs = (_sigmoid(f) * cs + _sigmoid(i) * np.tanh(ci)).astype(np.float32)
            hs = (_sigmoid(o) * np.tanh(cs)).astype(np.float32)
            outs.append(hs)
        y = np.stack(outs)
        return y[::-1] if reverse else y
    return Tensor(fn, [inp.static[0], inp.static[1], h])


def random_uniform_initializer(minval, maxval, seed=None):
    def init(count):
        return np.random.default_rng(seed).uniform(minval, maxval, count).astype(np.float32)
    return init


def _opaque_layout(input_size, units):
    """[TF-recall] cuDNN's canonical parameter order of ONE bidirectional LSTM layer, as index ranges into the flat buffer:
    {(direction, "W"|"R", gate): (offset, rows, cols)} and {(direction, "bW"|"bR", gate): offset}; gates in cuDNN's order i, f, c, o;
    all matrices (direction-major, W before R) come before all biases (direction-major, bW before bR)."""
    mats, biases, at = {}, {}, 0
    for d in ("fw", "bw"):
        for kind, cols in (("W", input_size), ("R", units)):
            for g in "ifco":
                mats[(d, kind, g)] = (at, units, cols)
                at += units * cols
    for d in ("fw", "bw"):
        for kind in ("bW", "bR"):
            for g in "ifco":
                biases[(d, kind, g)] = at
                at += units
    return mats, biases, at


def _opaque_to_canonical(opaque, input_size, units):
    """What CudnnLSTMSaveable stores: per direction the CudnnCompatibleLSTMCell kernel [input + units, 4 units] (gate columns in
    LSTMBlockCell's order i, c~, f, o) and bias [4 units] = bW + bR.  Written independently of clair_amd/tf_bundle.py's conversion."""
    mats, biases, total = _opaque_layout(input_size, units)
    assert opaque.size == total
    out = {}
    for d in ("fw", "bw"):
        cols, bias = [], []
        for g in "icfo":
            ow, r, c = mats[(d, "W", g)]
            orr, r2, c2 = mats[(d, "R", g)]
            cols.append(np.concatenate([opaque[ow:ow + r * c].reshape(r, c).T, opaque[orr:orr + r2 * c2].reshape(r2, c2).T], axis=0))
            bias.append(opaque[biases[(d, "bW", g)]:biases[(d, "bW", g)] + units] + opaque[biases[(d, "bR", g)]:biases[(d, "bR", g)] + units])
        out[(d, "kernel")] = np.concatenate(cols, axis=1).astype(np.float32)
        out[(d, "bias")] = np.concatenate(bias).astype(np.float32)
    return out


class _CudnnLSTM(object):
    """tf.contrib.cudnn_rnn.CudnnLSTM(num_layers=1, direction="bidirectional"): ONE variable, "<scope>/cudnn_lstm/opaque_kernel"."""

    def __init__(self, num_layers, num_units, direction="bidirectional", dtype=None, kernel_initializer=None, bias_initializer=None, seed=None):
        assert num_layers == 1 and direction == "bidirectional"
        self.units, self.k_init, self.b_init, self.var = num_units, kernel_initializer, bias_initializer, None

    def build(self, input_shape):
        self.input_size = list(input_shape)[-1]
        mats, biases, total = _opaque_layout(self.input_size, self.units)
        n_w = total - 16 * self.units
        value = np.concatenate([self.k_init(n_w), self.b_init(16 * self.units)]).astype(np.float32)
        with variable_scope("cudnn_lstm"):
            self.var = Variable(_scoped("opaque_kernel"), (total,), float32, value)
        self.var.cudnn = (self.input_size, self.units)
        self.var.scope = "/".join(_STATE["scopes"] + ["cudnn_lstm"])

    def __call__(self, inp):
        h, var = self.units, self.var

        def run(x, kernel, bias, reverse):
            if reverse:
                x = x[::-1]
            n = x.shape[1]
            cs, hs, outs = np.zeros((n, h), np.float32), np.zeros((n, h), np.float32), []
            for t in range(x.shape[0]):
                z = (np.concatenate([x[t], hs], axis=1) @ kernel + bias).astype(np.float32)
                i, ci, f, o = z[:, :h], z[:, h:2 * h], z[:, 2 * h:3 * h], z[:, 3 * h:]
                cs = (_sigmoid(f) * cs + _sigmoid(i) * np.tanh(ci)).astype(np.float32)
                hs = (_sigmoid(o) * np.tanh(cs)).astype(np.float32)
                outs.append(hs)
            y = np.stack(outs)
            return y[::-1] if reverse else y

        def fn(feed):
            c = _opaque_to_canonical(var.value, self.input_size, h)
            x = inp.fn(feed)
            return np.concatenate([run(x, c[("fw", "kernel")], c[("fw", "bias")], False), run(x, c[("bw", "kernel")], c[("bw", "bias")], True)], axis=2)
        return Tensor(fn, [inp.static[0], inp.static[1], 2 * h]), None


def _stack_bidirectional_dynamic_rnn(cells_fw, cells_bw, inputs, dtype=None, time_major=False):
    assert time_major and dtype is float32
    prev = inputs
    with variable_scope("stack_bidirectional_rnn"):
        for i, (cf, cb) in enumerate(zip(cells_fw, cells_bw)):
            with variable_scope("cell_%d" % i):
                with variable_scope("bidirectional_rnn"):
                    with variable_scope("fw"):
                        fw = _dynamic_rnn(cf, prev, False)
                    with variable_scope("bw"):
                        bw = _dynamic_rnn(cb, prev, True)
            prev = Tensor((lambda fw, bw: lambda feed: np.concatenate([fw.fn(feed), bw.fn(feed)], axis=2))(fw, bw),
                          [fw.static[0], fw.static[1], fw.static[2] + bw.static[2]])
    return prev, None, None


class ConfigProto(object):
    def __init__(self, **kw):
        self.kw = kw


class Session(object):
    def __init__(self, config=None):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass

    def run(self, fetches, feed_dict=None):
        feed = dict(feed_dict or {})
        feed.pop(_MEMO, None)

        def ev(f):
            return [ev(g) for g in f] if isinstance(f, (list, tuple)) else f.fn(feed)
        return ev(fetches)


# ---- tf.train: Saver writes a tensor bundle with tools/make_tf_bundle_fixture.py's building blocks (one plain data block) ----------------
def _fixture_writer():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_tf_bundle_fixture as w
    return w


class _Saver(object):
    def __init__(self, var_list=None):
        self.var_list = var_list

    def _tensors(self):
        """(name, variable dtype, array) of everything this Saver writes.  Default Saver: every global variable, a CudnnLSTM layer's opaque
        buffer through its saveable -- the canonical per-direction tensors under the layer's scope [TF-recall: CudnnLSTMSaveable].  A Saver
        over an explicit name -> variable dict writes those variables as they are (no saveable)."""
        if self.var_list is not None:
            return [(name, v.dtype, np.asarray(v.value)) for name, v in self.var_list.items()]
        out = []
        for v in global_variables():
            if getattr(v, "cudnn", None):
                for (d, kind), value in _opaque_to_canonical(v.value, *v.cudnn).items():
                    out.append(("%s/stack_bidirectional_rnn/cell_0/bidirectional_rnn/%s/cudnn_compatible_lstm_cell/%s" % (v.scope, d, kind), float32, value))
            else:
                out.append((v.name.split(":")[0], v.dtype, np.asarray(v.value)))
        return out

    def save(self, sess, prefix, write_meta_graph=True):
        import struct
        w = _fixture_writer()
        data, items = b"", [(b"", b"\x08\x01\x10\x00\x1a\x02\x08\x01")]
        for name, dtype, value in sorted(self._tensors(), key=lambda t: t[0].encode()):
            raw = np.ascontiguousarray(value, dtype="<f4" if dtype is float32 else "<i8").tobytes()
            items.append((name.encode(), w.entry(1 if dtype is float32 else 9, tuple(np.shape(value)), len(data), len(raw), w.masked(raw))))
            data += raw
        out = b""

        def emit(contents):
            nonlocal out
            handle = w.varint(len(out)) + w.varint(len(contents))
            out += contents + b"\x00" + struct.pack("<I", w.masked(contents + b"\x00"))
            return handle
        index_items = []
        for at in range(0, len(items), 48):
            chunk = items[at:at + 48]
            index_items.append((chunk[-1][0], emit(w.block(chunk, 16))))
        meta = emit(w.block([], 16))
        index = emit(w.block(index_items, 1))
        footer = meta + index
        out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
        with open(prefix + ".data-00000-of-00001", "wb") as f:
            f.write(data)
        with open(prefix + ".index", "wb") as f:
            f.write(out)
        return prefix


class _CheckpointReader(object):
    def __init__(self, prefix):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        from clair_amd import tf_bundle
        self.entries = tf_bundle.read_index(prefix + ".index")

    def get_variable_to_shape_map(self):
        return {k: list(v["shape"]) for k, v in self.entries.items()}

    def get_variable_to_dtype_map(self):
        return {k: (float32 if v["dtype"] == 1 else int64) for k, v in self.entries.items()}


def _global_step():
    return Variable("global_step", (), int64, np.int64(0))


contrib = types.SimpleNamespace(
    rnn=types.SimpleNamespace(stack_bidirectional_dynamic_rnn=_stack_bidirectional_dynamic_rnn),
    cudnn_rnn=types.SimpleNamespace(CudnnCompatibleLSTMCell=_CudnnCompatibleLSTMCell, CudnnLSTM=_CudnnLSTM))
test = types.SimpleNamespace(is_gpu_available=lambda cuda_only=False: True)
layers = types.SimpleNamespace(dense=_dense)
nn = types.SimpleNamespace(elu=_elu, softmax=_softmax)
train = types.SimpleNamespace(Saver=_Saver, NewCheckpointReader=_CheckpointReader, get_or_create_global_step=_global_step)
