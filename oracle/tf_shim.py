"""oracle.tf_shim — a minimal lazy-graph stand-in for the TensorFlow-1.12 API surface that the
reference's four hot-path model files call.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Why it exists.  TensorFlow 1.12.3 (requirements.txt:4) cannot be installed here, so the reference's
model graphs could not be executed, and every training row of SURVEY §8 rested on a hand restatement
(oracle/train.py).  The graph-BUILDING code of the reference, however, is plain Python over ~40 TF
calls.  With this module registered as `tensorflow`, the reference's own, unmodified

    model/general_recommender/MF.py        (whole class: __init__, build_graph, train_model, predict)
    model/general_recommender/LightGCN.py  (incl. create_adj_mat, _create_lightgcn_embed, create_bpr_loss)
    model/general_recommender/NGCF.py      (incl. get_adj_mat, _split_A_hat, _create_ngcf_embed)
    model/general_recommender/MultiVAE.py  (q_graph, p_graph, _create_loss, train_model, predict)
    util/learner.py, util/tool.py          (losses, optimiser selection, l2_loss, inner_product, ...)

run as they are (oracle/ref_models.py loads them from /root/reference).  The graph is therefore the
reference's; the PRIMITIVES are torch-CPU ops and the derivatives come from torch.autograd — an
autograd that is independent of the hand-derived backward passes in oracle/train.py and in the HIP
kernels.  What remains restated [EXT] is each primitive's published definition (cited per op below)
and the optimiser update rules of TF 1.12 `python/training/*.py` + `core/kernels/training_ops.cc`.

The graph is lazy: every call returns a `Tensor` node; `Session.run(fetches, feed_dict)` evaluates
the nodes it needs (memoised per run), differentiates the loss of every fetched `minimize` op with
respect to the trainable variables it reaches, and applies the updates after all fetches were
computed (so a fetched loss is the pre-update loss, as in `sess.run((loss, optimizer))`).

Float width: `set_float("float32" | "float64")` BEFORE the graph is built; `tf.float32` then means
that width, which is how the same reference code yields its own fp64 error-bar twin.

Random inputs (`tf.nn.dropout`, `tf.random_normal`, `tf.random_uniform`) are data: a Session draws
them from its own numpy RandomState unless arrays were queued with `Session.inject(...)`; every
draw of a run is appended to `Session.draw_log` so that a golden file can carry it.
"""
import contextlib
import sys
import types

import numpy as np
import torch

# --------------------------------------------------------------------------------------- dtypes
_FLOAT = {"t": torch.float32}


def set_float(name):
    _FLOAT["t"] = {"float32": torch.float32, "float64": torch.float64}[name]


def float_dtype():
    return _FLOAT["t"]


class DType:
    def __init__(self, name, is_float):
        self.name, self.is_float = name, is_float
        self.base_dtype = self

    def torch(self):
        if self.name == "bool":
            return torch.bool
        return _FLOAT["t"] if self.is_float else torch.int64

    def __repr__(self):
        return "tf_shim." + self.name


float32 = DType("float32", True)
float64 = DType("float64", True)
int32 = DType("int32", False)
int64 = DType("int64", False)
bool = DType("bool", False)  # noqa: A001  (the reference writes tf.bool)


# --------------------------------------------------------------------------------------- graph
class _Graph:
    def __init__(self):
        self.variables = []
        self.counter = 0


_GRAPH = _Graph()
_RUN = {"session": None}


def reset_default_graph():
    _GRAPH.variables = []
    _GRAPH.counter = 0


def _to_torch(x):
    """constants: python scalars stay scalars (no dtype promotion, as a TF constant takes the other
    operand's dtype); arrays become tensors of the session float width / int64."""
    if isinstance(x, (int, float)):
        return x
    if isinstance(x, torch.Tensor):
        return x
    a = np.asarray(x)
    if a.ndim == 0:                                     # (np.ascontiguousarray would make it 1-d)
        if a.dtype.kind == "f":
            return torch.tensor(float(a), dtype=_FLOAT["t"])
        return torch.tensor(a.item())
    if a.dtype.kind == "f":
        return torch.from_numpy(np.ascontiguousarray(a)).to(_FLOAT["t"])
    if a.dtype.kind == "b":
        return torch.from_numpy(np.ascontiguousarray(a))
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.int64)


class Tensor:
    """A node of the lazy graph: `fn(*evaluated inputs)`."""

    def __init__(self, fn, inputs=(), kind="op", name=None):
        self.fn, self.kind, self.name = fn, kind, name
        self.inputs = [i if isinstance(i, Tensor) else _Const(i) for i in inputs]
        _GRAPH.counter += 1
        self.order = _GRAPH.counter

    # arithmetic the model files write with operators
    def __add__(self, o): return add(self, o)
    def __radd__(self, o): return add(o, self)
    def __sub__(self, o): return subtract(self, o)
    def __rsub__(self, o): return subtract(o, self)
    def __mul__(self, o): return multiply(self, o)
    def __rmul__(self, o): return multiply(o, self)
    def __truediv__(self, o): return div(self, o)
    def __rtruediv__(self, o): return div(o, self)
    def __neg__(self): return Tensor(lambda a: -a, [self])
    def __pow__(self, p): return pow(self, p)
    def __getitem__(self, key): return Tensor(lambda a: a[key], [self])
    __hash__ = object.__hash__

    def __repr__(self):
        return "<tf_shim.Tensor %s #%d>" % (self.name or self.kind, self.order)


class _Const(Tensor):
    def __init__(self, value):
        self.value = _to_torch(value)
        self.fn, self.kind, self.name, self.inputs = None, "const", None, []
        self.order = 0


class _Placeholder(Tensor):
    def __init__(self, dtype, shape, name, default=None):
        Tensor.__init__(self, None, [], kind="placeholder", name=name)
        self.dtype, self.shape, self.default = dtype, shape, default


class Variable(Tensor):
    """tf.Variable: holds its value (a torch tensor) between runs."""

    def __init__(self, initial_value, trainable=True, name=None, dtype=None, **_):
        Tensor.__init__(self, None, [], kind="variable", name=name)
        if isinstance(initial_value, Tensor):
            with torch.no_grad():
                initial_value = _evaluate([initial_value], {})[0]
        self.value = _to_torch(np.asarray(initial_value) if not isinstance(initial_value, torch.Tensor)
                               else initial_value).clone()
        if self.value.dtype.is_floating_point:
            self.value = self.value.to(_FLOAT["t"])
        self.trainable = trainable
        _GRAPH.variables.append(self)

    def load(self, array):
        """overwrite the value (golden scripts install fixed initial tables)"""
        t = _to_torch(np.asarray(array))
        assert tuple(t.shape) == tuple(self.value.shape), (t.shape, self.value.shape)
        self.value = t.to(self.value.dtype).clone()

    def numpy(self):
        return self.value.detach().numpy().copy()


class _Op:
    """a fetched side effect (assign / minimize): evaluated for its effect, run() returns None"""
    kind = "sideeffect"


def _evaluate(nodes, feeds, leaves=None):
    """values of `nodes` (memoised, iterative post-order so that deep graphs do not recurse)"""
    memo = {}
    leaves = leaves or {}
    out = []
    for root in nodes:
        stack = [root]
        while stack:
            n = stack[-1]
            k = id(n)
            if k in memo:
                stack.pop()
                continue
            if n.kind == "const":
                memo[k] = n.value
            elif n.kind == "variable":
                memo[k] = leaves.get(k, n.value)
            elif n.kind == "placeholder":
                if k in feeds:
                    v = feeds[k]
                    t = _to_torch(np.asarray(v))
                    memo[k] = t.to(n.dtype.torch()) if isinstance(t, torch.Tensor) else \
                        torch.tensor(t, dtype=n.dtype.torch())
                elif n.default is not None:
                    memo[k] = torch.tensor(n.default, dtype=n.dtype.torch()) \
                        if isinstance(n.default, (int, float)) else _to_torch(n.default)
                else:
                    raise ValueError("placeholder %r was not fed" % (n.name,))
            else:
                pending = [i for i in n.inputs if id(i) not in memo]
                if pending:
                    stack.extend(reversed(pending))     # first input first
                    continue
                memo[k] = n.fn(*[memo[id(i)] for i in n.inputs])
            stack.pop()
        out.append(memo[id(root)])
    return out


def _ancestors(root):
    seen, order, stack = set(), [], [root]
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        order.append(n)
        stack.extend(n.inputs)
    return order


# --------------------------------------------------------------------------------------- API: graph inputs
def placeholder(dtype, shape=None, name=None):
    return _Placeholder(dtype, shape, name)


def placeholder_with_default(input, shape=None, name=None):  # noqa: A002
    dt = float32 if isinstance(input, float) else int32
    return _Placeholder(dt, shape, name, default=input)


def constant(value, dtype=None, **_):
    return _Const(value)


def zeros(shape, dtype=None, **_):
    return Tensor(lambda: torch.zeros(*[int(s) for s in shape], dtype=_FLOAT["t"]), [])


def identity(x, name=None):
    return Tensor(lambda a: a, [x])


@contextlib.contextmanager
def name_scope(name=None, *a, **k):
    yield name


variable_scope = name_scope


def set_random_seed(seed):
    _RUN["graph_seed"] = seed


def global_variables_initializer():
    op = _Op()
    op.tensors = []
    op.apply = lambda sess: None            # variables are initialised where they are created
    op.prepare = lambda sess, values: None
    return op


def trainable_variables():
    return [v for v in _GRAPH.variables if v.trainable]


def shape(x, name=None):
    return Tensor(lambda a: tuple(a.shape), [x])


def cast(x, dtype, name=None):
    return Tensor(lambda a: a.to(dtype.torch()), [x])


# --------------------------------------------------------------------------------------- API: math
# Each op names the TF definition it follows [EXT: tensorflow r1.12].
def _bin(f):
    def op(a, b, name=None):
        return Tensor(f, [a, b])
    return op


add = _bin(lambda a, b: a + b)
subtract = _bin(lambda a, b: a - b)
multiply = _bin(lambda a, b: a * b)
div = _bin(lambda a, b: a / b)                      # tf.div on floats = true division
divide = div
maximum = _bin(lambda a, b: torch.maximum(a, b) if isinstance(b, torch.Tensor)
               else torch.clamp(a, min=b))


def square(x, name=None):
    return Tensor(lambda a: a * a, [x])


def sqrt(x, name=None):
    return Tensor(torch.sqrt, [x])


def exp(x, name=None):
    return Tensor(torch.exp, [x])


def log(x, name=None):
    return Tensor(torch.log, [x])


def pow(x, y, name=None):  # noqa: A001
    return Tensor(lambda a, b: torch.pow(a, b), [x, y])


def floor(x, name=None):
    return Tensor(torch.floor, [x])


def sigmoid(x, name=None):
    return Tensor(torch.sigmoid, [x])


def tanh(x, name=None):
    return Tensor(torch.tanh, [x])


def log_sigmoid(x, name=None):
    """math_ops.log_sigmoid: -softplus(-x)"""
    return Tensor(lambda a: -torch.nn.functional.softplus(-a), [x])


def _axes(axis):
    if axis is None:
        return None
    return tuple(axis) if isinstance(axis, (list, tuple)) else (axis,)


def reduce_sum(x, axis=None, keepdims=False, name=None, keep_dims=None, reduction_indices=None):
    ax = _axes(axis if axis is not None else reduction_indices)
    kd = True if (keepdims or keep_dims) else False
    return Tensor(lambda a: torch.sum(a) if ax is None else torch.sum(a, dim=ax, keepdim=kd), [x])


def reduce_mean(x, axis=None, keepdims=False, name=None, keep_dims=None):
    ax = _axes(axis)
    kd = True if (keepdims or keep_dims) else False
    return Tensor(lambda a: torch.mean(a) if ax is None else torch.mean(a, dim=ax, keepdim=kd), [x])


def add_n(inputs, name=None):
    def f(*xs):
        acc = xs[0]
        for t in xs[1:]:
            acc = acc + t
        return acc
    return Tensor(f, list(inputs))


def concat(values, axis, name=None):
    return Tensor(lambda *xs: torch.cat(xs, dim=axis), list(values))


def stack(values, axis=0, name=None):
    return Tensor(lambda *xs: torch.stack(xs, dim=axis), list(values))


def split(value, num_or_size_splits, axis=0, name=None):
    sizes = list(num_or_size_splits)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    return [Tensor(lambda a, lo=int(offs[i]), n=int(sizes[i]): a.narrow(axis, lo, n), [value])
            for i in range(len(sizes))]


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    def f(x, y):
        if transpose_a:
            x = x.t()
        if transpose_b:
            y = y.t()
        return x @ y
    return Tensor(f, [a, b])


def squeeze(x, axis=None, name=None):
    return Tensor(lambda a: a.squeeze() if axis is None else a.squeeze(axis), [x])


def expand_dims(x, axis, name=None):
    return Tensor(lambda a: a.unsqueeze(axis), [x])


def transpose(x, perm=None, name=None):
    return Tensor(lambda a: a.t() if perm is None else a.permute(*perm), [x])


# --------------------------------------------------------------------------------------- API: sparse
class SparseTensor:
    """tf.SparseTensor(indices [nnz,2], values, dense_shape); rows in the order given (the
    reference passes scipy COO = row-major)."""

    def __init__(self, indices, values, dense_shape):
        idx = np.asarray(indices)
        self.indices = idx.reshape(-1, 2).astype(np.int64)
        self.values = np.asarray(values)
        self.dense_shape = tuple(int(s) for s in dense_shape)
        self._cache = {}

    def torch(self):
        dt = _FLOAT["t"]
        if dt not in self._cache:
            i = torch.from_numpy(np.ascontiguousarray(self.indices.T))
            v = torch.from_numpy(np.ascontiguousarray(self.values)).to(dt)
            self._cache[dt] = torch.sparse_coo_tensor(i, v, self.dense_shape).coalesce().to_sparse_csr()
        return self._cache[dt]


class _DynSparse:
    """A SparseTensor whose VALUES are a graph tensor: what tf.sparse_retain / `sparse * scalar` produce in
    NGCF.py:352-362 (`_dropout_sparse`: node dropout of the adjacency).  A dropped entry keeps its slot with
    value 0: in sparse_tensor_dense_matmul it adds an exact +0, which is what removing it does [EXT:
    sparse_ops.sparse_retain keeps the entries whose mask is True; SparseTensor.__mul__ scales the values]."""

    def __init__(self, indices, values, dense_shape):
        self.indices, self.values, self.dense_shape = indices, values, tuple(int(x) for x in dense_shape)

    def __mul__(self, o):
        return _DynSparse(self.indices, multiply(self.values, o), self.dense_shape)
    __rmul__ = __mul__


def _sparse_mul(self, o):
    base = Tensor(lambda: torch.from_numpy(np.ascontiguousarray(self.values)).to(_FLOAT["t"]), [])
    return _DynSparse(self.indices, multiply(base, o), self.dense_shape)


SparseTensor.__mul__ = _sparse_mul
SparseTensor.__rmul__ = _sparse_mul


def sparse_retain(sp_input, to_retain, name=None):
    vals = sp_input.values
    if not isinstance(vals, Tensor):
        host = np.ascontiguousarray(sp_input.values)
        vals = Tensor(lambda: torch.from_numpy(host).to(_FLOAT["t"]), [])
    kept = Tensor(lambda v, m: v * m.to(v.dtype), [vals, to_retain])
    return _DynSparse(sp_input.indices, kept, sp_input.dense_shape)


def sparse_tensor_dense_matmul(sp_a, b, name=None, **_):
    """sparse_ops.sparse_tensor_dense_matmul: out[r] = sum over the stored entries of row r of
    value * b[col]; the derivative with respect to b is left to torch.autograd."""
    if isinstance(sp_a, _DynSparse):
        rows = torch.from_numpy(np.ascontiguousarray(sp_a.indices[:, 0]))
        cols = torch.from_numpy(np.ascontiguousarray(sp_a.indices[:, 1]))
        crow = torch.zeros(sp_a.dense_shape[0] + 1, dtype=torch.int64)
        crow[1:] = torch.cumsum(torch.bincount(rows, minlength=sp_a.dense_shape[0]), 0)
        assert (rows[1:] >= rows[:-1]).all().item(), "row-major entries expected (scipy COO of a CSR slice)"

        def f(v, x):
            a = torch.sparse_csr_tensor(crow, cols, v.detach(), size=sp_a.dense_shape)   # the adjacency is a constant
            return torch.sparse.mm(a, x)
        return Tensor(f, [sp_a.values, b])
    return Tensor(lambda x: torch.sparse.mm(sp_a.torch(), x), [b])


# --------------------------------------------------------------------------------------- API: random inputs
def random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):  # noqa: A002
    def f(shp):
        a = _RUN["session"].draw("random_normal", tuple(int(s) for s in shp), mean=mean, stddev=stddev)
        return _to_torch(a)
    return Tensor(f, [shape if isinstance(shape, Tensor) else _Const(np.asarray(shape, np.int64))])


def random_uniform(shape, minval=0.0, maxval=1.0, dtype=None, seed=None, name=None):  # noqa: A002
    def f(shp):
        a = _RUN["session"].draw("random_uniform", tuple(int(s) for s in shp), minval=minval, maxval=maxval)
        return _to_torch(a)
    return Tensor(f, [shape if isinstance(shape, Tensor) else _Const(np.asarray(shape, np.int64))])


# --------------------------------------------------------------------------------------- API: tf.nn
def _embedding_lookup(params, ids, name=None, **_):
    """array_ops.gather on axis 0; kind="gather" is what makes a variable's gradient an
    IndexedSlices (sparse optimiser application) when every consumer is one"""
    return Tensor(lambda p, i: p.index_select(0, i.reshape(-1)).reshape(tuple(i.shape) + tuple(p.shape[1:])),
                  [params, ids], kind="gather")


def _softplus(x, name=None):
    return Tensor(torch.nn.functional.softplus, [x])


def _l2_loss(t, name=None):
    """nn.l2_loss: sum(t ** 2) / 2"""
    return Tensor(lambda a: torch.sum(a * a) / 2, [t])


def _leaky_relu(features, alpha=0.2, name=None):
    """nn_ops.leaky_relu (r1.12): maximum(alpha * features, features)"""
    return Tensor(lambda a: torch.maximum(a * alpha, a), [features])


def _dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
    """nn_ops.dropout (r1.12): binary = floor(keep_prob + uniform[0,1)); div(x, keep_prob) * binary.
    The uniform draw is an input: the session hands out the {0,1} mask."""
    def f(a, kp):
        k = float(kp)
        m = _RUN["session"].draw("dropout", tuple(a.shape), keep_prob=k)
        return (a / kp) * _to_torch(m).to(a.dtype)
    return Tensor(f, [x, keep_prob])


def _l2_normalize(x, axis=None, epsilon=1e-12, name=None, dim=None):
    """nn_impl.l2_normalize: x * rsqrt(maximum(sum(x², axis, keepdims), epsilon))"""
    ax = axis if axis is not None else dim

    def f(a):
        ss = torch.sum(a * a, dim=ax, keepdim=True)
        return a * torch.rsqrt(torch.clamp(ss, min=epsilon))
    return Tensor(f, [x])


def _log_softmax(logits, axis=-1, name=None):
    return Tensor(lambda a: torch.log_softmax(a, dim=axis), [logits])


def _softmax(logits, axis=-1, name=None):
    return Tensor(lambda a: torch.softmax(a, dim=axis), [logits])


def _sigmoid_cross_entropy_with_logits(labels=None, logits=None, name=None):
    """nn_impl: max(x, 0) - x * z + log(1 + exp(-|x|))"""
    return Tensor(lambda z, x: torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-torch.abs(x))),
                  [labels, logits])


nn = types.SimpleNamespace(
    embedding_lookup=_embedding_lookup, softplus=_softplus, l2_loss=_l2_loss, leaky_relu=_leaky_relu,
    dropout=_dropout, l2_normalize=_l2_normalize, log_softmax=_log_softmax, softmax=_softmax,
    sigmoid=sigmoid, tanh=tanh, relu=lambda x, name=None: Tensor(torch.relu, [x]),
    elu=lambda x, name=None: Tensor(torch.nn.functional.elu, [x]),
    selu=lambda x, name=None: Tensor(torch.selu, [x]),
    sigmoid_cross_entropy_with_logits=_sigmoid_cross_entropy_with_logits)


def _losses_sigmoid_cross_entropy(multi_class_labels, logits, weights=1.0, **_):
    """losses_impl.sigmoid_cross_entropy with unit weights: SUM_BY_NONZERO_WEIGHTS = mean"""
    per = _sigmoid_cross_entropy_with_logits(labels=multi_class_labels, logits=logits)
    return Tensor(lambda a: torch.sum(a) / a.numel(), [per])


losses = types.SimpleNamespace(sigmoid_cross_entropy=_losses_sigmoid_cross_entropy)


# --------------------------------------------------------------------------------------- initialisers
class _Init:
    """callable(shape) -> fp32 array, drawn from one module-level stream (golden scripts overwrite
    the variables with fixed tables afterwards; only scale and shape matter here)"""
    rng = np.random.RandomState(2017)

    def __init__(self, draw):
        self.draw = draw

    def __call__(self, shape, dtype=None, partition_info=None):
        return self.draw(tuple(int(s) for s in shape)).astype(np.float32)


def _fans(shape):
    fan_in = shape[-2] if len(shape) > 1 else shape[-1]
    fan_out = shape[-1]
    return float(fan_in), float(fan_out)


def _variance_scaling(factor=2.0, mode="FAN_IN", uniform=False, seed=None, dtype=None):
    """contrib.layers.variance_scaling_initializer [EXT]"""
    def draw(shape):
        fi, fo = _fans(shape)
        n = {"FAN_IN": fi, "FAN_OUT": fo, "FAN_AVG": (fi + fo) / 2.0}[mode]
        if uniform:
            lim = np.sqrt(3.0 * factor / n)
            return _Init.rng.uniform(-lim, lim, shape)
        std = np.sqrt(1.3 * factor / n)
        return np.clip(_Init.rng.randn(*shape), -2, 2) * std
    return _Init(draw)


def _xavier(uniform=True, seed=None, dtype=None):
    return _variance_scaling(factor=1.0, mode="FAN_AVG", uniform=uniform)


def truncated_normal_initializer(mean=0.0, stddev=1.0, seed=None, dtype=None):
    return _Init(lambda s: mean + stddev * np.clip(_Init.rng.randn(*s), -2, 2))


def random_normal_initializer(mean=0.0, stddev=1.0, seed=None, dtype=None):
    return _Init(lambda s: mean + stddev * _Init.rng.randn(*s))


def random_uniform_initializer(minval=0.0, maxval=None, seed=None, dtype=None):
    return _Init(lambda s: _Init.rng.uniform(minval, maxval, s))


def _l2_regularizer(scale, scope=None):
    """contrib.layers.l2_regularizer: scale * nn.l2_loss(w); a scale of 0 disables it (returns None)"""
    if float(scale) == 0.0:
        return lambda _: None
    return lambda w: multiply(float(scale), _l2_loss(w))


def _apply_regularization(regularizer, weights_list=None):
    """contrib.layers.apply_regularization: add_n of the penalties, None -> constant 0"""
    pen = [regularizer(w) for w in weights_list]
    pen = [p if p is not None else _Const(0.0) for p in pen]
    return add_n(pen)


contrib = types.ModuleType("tensorflow.contrib")
contrib.layers = types.ModuleType("tensorflow.contrib.layers")
contrib.layers.xavier_initializer = _xavier
contrib.layers.variance_scaling_initializer = _variance_scaling
contrib.layers.l2_regularizer = _l2_regularizer
contrib.layers.apply_regularization = _apply_regularization


# --------------------------------------------------------------------------------------- assign
def assign(ref, value, name=None):
    op = _Op()
    node = value if isinstance(value, Tensor) else _Const(value)
    op.tensors = [node]

    def prepare(sess, values):
        op._new = values[0].detach().clone()

    def apply(sess):
        ref.value = op._new.to(ref.value.dtype)
    op.prepare, op.apply = prepare, apply
    return op


# --------------------------------------------------------------------------------------- optimisers
class _Optimizer:
    """compute_gradients + apply_gradients of python/training/optimizer.py, reduced to what
    `.minimize(loss)` does: d loss / d (every trainable variable the loss reaches); a variable whose
    every consumer is a gather receives IndexedSlices -> the sparse update (duplicate indices summed
    first, `_apply_sparse_duplicate_indices`), any other variable the dense update."""

    def minimize(self, loss, global_step=None, var_list=None, name=None):
        return _Minimize(self, loss)

    def slots(self, var):
        raise NotImplementedError

    def begin(self, dt):
        pass

    def finish(self):
        pass


class _Minimize(_Op):
    def __init__(self, opt, loss):
        self.opt, self.loss = opt, loss
        anc = _ancestors(loss)
        consumers = {}
        for n in anc:
            for i in n.inputs:
                if i.kind == "variable":
                    consumers.setdefault(id(i), []).append(n)
        self.vars = [v for v in _GRAPH.variables if v.trainable and id(v) in consumers]
        self.gathers = {id(v): sorted(consumers[id(v)], key=lambda n: n.order) for v in self.vars}
        self.sparse = {id(v): all(c.kind == "gather" and c.inputs[0] is v for c in consumers[id(v)])
                       for v in self.vars}
        self.state = {id(v): opt.slots(v) for v in self.vars}
        self.tensors = [loss]
        self.last_grads = None


class _Adam(_Optimizer):
    """python/training/adam.py + training_ops.cc ApplyAdam [EXT]"""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, **_):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.b1p = self.b2p = None

    def slots(self, var):
        return {"m": torch.zeros_like(var.value), "v": torch.zeros_like(var.value)}

    def begin(self, dt):
        t = lambda x: torch.tensor(x, dtype=dt)  # noqa: E731
        if self.b1p is None:
            self.b1p, self.b2p = t(self.b1), t(self.b2)      # beta powers start at beta (adam.py:_create_slots)
        self._c = (t(self.lr), t(self.b1), t(self.b2), t(self.eps), t(1.0))

    def update(self, var, st, g, sparse, rows):
        lr, b1, b2, eps, one = self._c
        m, v = st["m"], st["v"]
        alpha = lr * torch.sqrt(one - self.b2p) / (one - self.b1p)
        if sparse:      # _apply_sparse_shared: every row decays; the summed slices are scatter-added
            m.mul_(b1).add_(g * (one - b1))
            v.mul_(b2).add_((g * g) * (one - b2))
            var.value = var.value - alpha * m / (torch.sqrt(v) + eps)
        else:           # ApplyAdam
            m.add_((g - m) * (one - b1))
            v.add_((g * g - v) * (one - b2))
            var.value = var.value - (m * alpha) / (torch.sqrt(v) + eps)

    def finish(self):
        self.b1p = self.b1p * self._c[1]
        self.b2p = self.b2p * self._c[2]


class _GradientDescent(_Optimizer):
    def __init__(self, learning_rate, **_):
        self.lr = learning_rate

    def slots(self, var):
        return {}

    def begin(self, dt):
        self._lr = torch.tensor(self.lr, dtype=dt)

    def update(self, var, st, g, sparse, rows):
        if sparse:
            var.value[rows] = var.value[rows] - self._lr * g[rows]
        else:
            var.value = var.value - self._lr * g


class _Adagrad(_Optimizer):
    """adagrad.py + training_ops.cc (Sparse)ApplyAdagrad: accum += g²; var -= lr * g * rsqrt(accum)"""

    def __init__(self, learning_rate, initial_accumulator_value=0.1, **_):
        self.lr, self.init = learning_rate, initial_accumulator_value

    def slots(self, var):
        return {"accum": torch.full_like(var.value, self.init)}

    def begin(self, dt):
        self._lr = torch.tensor(self.lr, dtype=dt)

    def update(self, var, st, g, sparse, rows):
        a = st["accum"]
        r = rows if sparse else slice(None)
        a[r] = a[r] + g[r] * g[r]
        var.value[r] = var.value[r] - (self._lr * g[r]) * (1.0 / torch.sqrt(a[r]))


class _RMSProp(_Optimizer):
    """rmsprop.py (decay 0.9, momentum 0, epsilon 1e-10) + training_ops.cc"""

    def __init__(self, learning_rate, decay=0.9, momentum=0.0, epsilon=1e-10, **_):
        self.lr, self.rho, self.mom, self.eps = learning_rate, decay, momentum, epsilon

    def slots(self, var):
        return {"ms": torch.ones_like(var.value), "mom": torch.zeros_like(var.value)}

    def begin(self, dt):
        self._c = [torch.tensor(x, dtype=dt) for x in (self.lr, self.rho, self.mom, self.eps, 1.0)]

    def update(self, var, st, g, sparse, rows):
        lr, rho, mom, eps, one = self._c
        ms, mo = st["ms"], st["mom"]
        if sparse:      # SparseApplyRMSProp
            gr = g[rows]
            ms[rows] = ms[rows] * rho + (gr * gr) * (one - rho)
            mo[rows] = mo[rows] * mom + (1.0 / torch.sqrt(ms[rows] + eps)) * lr * gr
            var.value[rows] = var.value[rows] - mo[rows]
        else:           # ApplyRMSProp
            ms.add_((g * g - ms) * (one - rho))
            mo.copy_(mo * mom + (g * lr) / torch.sqrt(eps + ms))
            var.value = var.value - mo


class _Momentum(_Optimizer):
    def __init__(self, learning_rate, momentum, **_):
        self.lr, self.mom = learning_rate, momentum

    def slots(self, var):
        return {"accum": torch.zeros_like(var.value)}

    def begin(self, dt):
        self._c = [torch.tensor(x, dtype=dt) for x in (self.lr, self.mom)]

    def update(self, var, st, g, sparse, rows):
        lr, mom = self._c
        a = st["accum"]
        r = rows if sparse else slice(None)
        a[r] = a[r] * mom + g[r]
        var.value[r] = var.value[r] - a[r] * lr


train = types.SimpleNamespace(AdamOptimizer=_Adam, GradientDescentOptimizer=_GradientDescent,
                              AdagradOptimizer=_Adagrad, RMSPropOptimizer=_RMSProp,
                              MomentumOptimizer=_Momentum)


# --------------------------------------------------------------------------------------- session
class ConfigProto:
    def __init__(self, **_):
        self.gpu_options = types.SimpleNamespace()


class Session:
    def __init__(self, config=None, seed=0, **_):
        self.rng = np.random.RandomState(seed)
        self.draw_log = []          # (kind, array) of every random input, in evaluation order
        self._queue = []
        self.grad_log = None        # {variable name: dense gradient} of the last minimize, when asked
        self.keep_grads = False

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def inject(self, arrays):
        """queue random inputs (taken in evaluation order instead of fresh draws)"""
        self._queue.extend(arrays)

    def draw(self, kind, shape, **p):
        if self._queue:
            a = np.asarray(self._queue.pop(0))
            assert tuple(a.shape) == tuple(shape), (kind, a.shape, shape)
        elif kind == "dropout":
            a = (self.rng.random_sample(shape) < p["keep_prob"]).astype(np.float32)
        elif kind == "random_normal":
            a = (p["mean"] + p["stddev"] * self.rng.randn(*shape)).astype(np.float32)
        else:
            a = self.rng.uniform(p["minval"], p["maxval"], shape).astype(np.float32)
        self.draw_log.append((kind, a))
        return a

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        flat = [fetches] if single else list(fetches)
        feeds = {id(k): v for k, v in (feed_dict or {}).items()}
        minimizers = [f for f in flat if isinstance(f, _Minimize)]
        others = [f for f in flat if isinstance(f, _Op) and not isinstance(f, _Minimize)]
        leaves = {}
        for mz in minimizers:
            for v in mz.vars:
                leaves.setdefault(id(v), v.value.detach().clone().requires_grad_(True))
        nodes = []
        for f in flat:
            nodes.extend(f.tensors if isinstance(f, _Op) and hasattr(f, "tensors") else
                         ([] if isinstance(f, _Op) else [f]))
        prev = _RUN["session"]
        _RUN["session"] = self
        try:
            with (torch.enable_grad() if minimizers else torch.no_grad()):
                # gather ids first (a sparse update needs the touched rows)
                id_nodes = [c.inputs[1] for mz in minimizers for v in mz.vars if mz.sparse[id(v)]
                            for c in mz.gathers[id(v)]]
                values = _evaluate(nodes + id_nodes, feeds, leaves)
                by_node = {id(n): val for n, val in zip(nodes + id_nodes, values)}
                grads = {}
                for mz in minimizers:
                    gl = torch.autograd.grad(by_node[id(mz.loss)], [leaves[id(v)] for v in mz.vars],
                                             allow_unused=True, retain_graph=True)
                    grads[id(mz)] = gl
        finally:
            _RUN["session"] = prev
        with torch.no_grad():
            for op in others:
                op.prepare(self, [by_node[id(n)] for n in op.tensors])
            for mz in minimizers:
                dt = mz.vars[0].value.dtype
                mz.opt.begin(dt)
                if self.keep_grads:
                    self.grad_log = {}
                for v, g in zip(mz.vars, grads[id(mz)]):
                    if g is None:
                        continue
                    rows = None
                    if mz.sparse[id(v)]:
                        ids = torch.cat([by_node[id(c.inputs[1])].reshape(-1) for c in mz.gathers[id(v)]])
                        rows = torch.unique(ids)
                    if self.keep_grads:
                        self.grad_log[v.name] = g.detach().numpy().copy()
                    mz.opt.update(v, mz.state[id(v)], g.detach(), mz.sparse[id(v)], rows)
                mz.opt.finish()
            for op in others:
                op.apply(self)
        out = []
        for f in flat:
            if isinstance(f, _Op):
                out.append(None)
            else:
                val = by_node[id(f)]
                if isinstance(val, torch.Tensor):
                    a = val.detach().numpy()
                    out.append(a.copy() if a.ndim else a[()])
                else:
                    out.append(val)
        return out[0] if single else out

    def close(self):
        pass


# --------------------------------------------------------------------------------------- install
def install():
    """register this module as `tensorflow` (+ tensorflow.contrib[.layers]); returns what to hand
    to `uninstall` to put the previous entries back"""
    me = sys.modules[__name__]
    saved = {k: sys.modules.get(k) for k in ("tensorflow", "tensorflow.contrib", "tensorflow.contrib.layers")}
    sys.modules["tensorflow"] = me
    sys.modules["tensorflow.contrib"] = contrib
    sys.modules["tensorflow.contrib.layers"] = contrib.layers
    return saved


def uninstall(saved):
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
