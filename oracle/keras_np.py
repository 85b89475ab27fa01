"""A numpy-backed stand-in for the slice of Keras the reference touches.

TEST INFRASTRUCTURE (see oracle/__init__.py).  TensorFlow/Keras are not
installable in the build container, so the reference cannot run as shipped.
This module implements -- on top of oracle/mlp_np.py, i.e. with exactly the
same arithmetic as the oracle -- the duck-typed surface listed in SURVEY.md
8b ("Model-object surface the agents touch"), so that tests/ref_shims can
expose it as ``tensorflow.keras`` and execute the reference's agent classes
and training loop *verbatim* from /root/reference.  That pins the oracle's
orchestration (message/rollback/consensus bookkeeping, RNG call order) to the
reference source; it cannot pin TensorFlow's own kernel numerics.
"""
import numpy as np
from . import mlp_np as M

F32 = np.float32
_shuffle_stream = None


def set_shuffle_stream(stream):
    """Mini-batch fits draw their per-epoch permutations from this stream
    (oracle.rpbcac_oracle.ShuffleStream) -- the oracle-defined shuffle."""
    global _shuffle_stream
    _shuffle_stream = stream


class Tensor(np.ndarray):
    """ndarray with a ``.numpy()`` method, standing in for an eager tf.Tensor."""

    def numpy(self):
        a = np.asarray(self)
        return a[()] if a.ndim == 0 else a

    def __getitem__(self, idx):
        r = super().__getitem__(idx)
        return r if isinstance(r, np.ndarray) else np.asarray(r).view(Tensor)


def as_tensor(x, dtype=None):
    arr = np.asarray(x, dtype=dtype)
    return arr.view(Tensor)


# ---- layers -----------------------------------------------------------------
class LeakyReLU:
    def __init__(self, alpha=0.3):
        self.alpha = alpha
        assert abs(alpha - 0.1) < 1e-12, "only the reference's alpha=0.1 is restated"


class _Sym:
    """Symbolic handle returned by ``layer.output`` / ``model.inputs``."""

    def __init__(self, layer=None):
        self.layer = layer


class Input(_Sym):
    def __init__(self, shape=None):
        super().__init__(None)
        self.shape = tuple(shape)


class Flatten:
    trainable = True

    def get_weights(self):
        return []

    def set_weights(self, w):
        assert len(w) == 0


class Dense:
    def __init__(self, units, activation=None):
        self.units, self.activation = units, activation
        self.trainable = True
        self.kernel = self.bias = None
        self.output = _Sym(self)

    def build(self, fan_in, rng):
        self.kernel = M.glorot_uniform(rng, fan_in, self.units)
        self.bias = np.zeros(self.units, F32)

    def get_weights(self):
        return [self.kernel.copy(), self.bias.copy()]

    def set_weights(self, w):
        k, b = w
        assert np.shape(k) == self.kernel.shape and np.shape(b) == self.bias.shape
        self.kernel = np.array(k, dtype=F32)
        self.bias = np.array(b, dtype=F32)


class _LayersNS:
    Dense, Flatten, LeakyReLU = Dense, Flatten, LeakyReLU


layers = _LayersNS()
_init_rng = np.random.default_rng(0)


def set_init_seed(seed):
    """Stand-in for tf.random.set_seed: seeds the Glorot initialiser."""
    global _init_rng
    _init_rng = np.random.default_rng(seed)


# ---- optimizers / losses ------------------------------------------------------
class SGD:
    def __init__(self, learning_rate=0.01):
        self.learning_rate = learning_rate


class Adam:
    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self.state = None


class MeanSquaredError:
    pass


class SparseCategoricalCrossentropy:
    pass


class _OptNS:
    SGD, Adam = SGD, Adam


class _LossNS:
    MeanSquaredError, SparseCategoricalCrossentropy = MeanSquaredError, SparseCategoricalCrossentropy


optimizers = _OptNS()
losses = _LossNS()


class History:
    def __init__(self, loss):
        self.history = {"loss": [float(l) for l in loss]}


# ---- models -------------------------------------------------------------------
class Model:
    """Functional sub-model ``Model(model.inputs, model.layers[-2].output)``:
    shares the Dense layer objects of the parent up to the given output."""

    def __init__(self, inputs=None, outputs=None, _layers=None):
        if _layers is None:
            src = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
            parent = src.parent_layers
            stop = parent.index(outputs.layer)
            _layers = parent[:stop + 1]
            self._input_shape = src.shape
        self.layers = list(_layers)
        self.optimizer = self.loss = None
        self._trainable_at_compile = None

    # -- structure
    @property
    def _dense(self):
        return [l for l in self.layers if isinstance(l, Dense)]

    @property
    def trainable(self):
        return all(l.trainable for l in self.layers)

    @trainable.setter
    def trainable(self, flag):
        for l in self.layers:
            l.trainable = bool(flag)

    @property
    def output_shape(self):
        return (None, self._dense[-1].units)

    def get_weights(self):
        return [w for l in self.layers for w in l.get_weights()]

    def set_weights(self, weights):
        weights = list(weights)
        assert len(weights) == 2 * len(self._dense)
        for k, l in enumerate(self._dense):
            l.set_weights(weights[2 * k:2 * k + 2])

    # -- forward
    def _params(self):
        return [a for l in self._dense for a in (l.kernel, l.bias)]

    def _flat_in(self, x):
        x = np.asarray(x, dtype=F32)
        return x.reshape(x.shape[0], -1)

    def _is_softmax(self):
        return self._dense[-1].activation == "softmax"

    def __call__(self, x):
        x = self._flat_in(x)
        dn = self._dense
        if len(dn) == 3:
            out = M.forward(self._params(), x)
            if self._is_softmax():
                out = M.softmax(out)
        elif len(dn) == 2:                       # feature sub-model
            out = M.features(self._params(), x)
        else:
            raise NotImplementedError
        return as_tensor(out)

    def predict(self, x):
        return np.asarray(self(x))

    # -- training
    def compile(self, optimizer=None, loss=None):
        self.optimizer, self.loss = optimizer, loss
        # Keras snapshots `trainable` at compile time.
        self._trainable_at_compile = [l.trainable for l in self._dense]
        if isinstance(optimizer, Adam) and optimizer.state is None:
            optimizer.state = M.AdamState(self._params(), optimizer.learning_rate, optimizer.beta_1,
                                          optimizer.beta_2, optimizer.epsilon)

    def _train_step(self, x, y, w):
        dn = self._dense
        params = self._params()
        logits, cache = M.forward(params, x, want_cache=True)
        if isinstance(self.loss, MeanSquaredError):
            loss, dout = M.mse_loss_and_dout(logits, y, sample_weight=w)
        else:
            loss, dout = M.sparse_ce_loss_and_dlogits(logits, y, sample_weight=w)
        tr = self._trainable_at_compile
        hidden_trainable = tr[0] and tr[1]
        assert tr[0] == tr[1] and tr[2], "only (all) or (head-only) trainable patterns occur in the reference"
        grads = M.backward(params, cache, dout, hidden_trainable=hidden_trainable)
        if isinstance(self.optimizer, Adam):
            M.adam_apply(params, grads, self.optimizer.state)
        else:
            M.sgd_apply(params, grads, self.optimizer.learning_rate)
        for k, l in enumerate(dn):                # params were updated in place; keep layer refs
            l.kernel, l.bias = params[2 * k], params[2 * k + 1]
        return loss

    def train_on_batch(self, x, y, sample_weight=None):
        x = self._flat_in(x)
        w = None if sample_weight is None else np.asarray(sample_weight, dtype=F32).reshape(x.shape[0])
        return float(self._train_step(x, np.asarray(y), w))

    def fit(self, x, y, sample_weight=None, batch_size=32, epochs=1, verbose=0):
        x = self._flat_in(x)
        y = np.asarray(y)
        B = x.shape[0]
        bs = min(int(batch_size), B)
        w = None if sample_weight is None else np.asarray(sample_weight, dtype=F32).reshape(B)
        perms = None
        if bs < B:
            assert _shuffle_stream is not None, "mini-batch fit needs keras_np.set_shuffle_stream(...)"
            perms = _shuffle_stream.perms(epochs, B)
        hist = []
        for e in range(epochs):
            order = np.arange(B) if perms is None else perms[e]
            tot, cnt = 0.0, 0
            for lo in range(0, B, bs):
                idx = order[lo:lo + bs]
                if perms is None and bs == B:
                    xb, yb, wb = x, y, w
                else:
                    xb, yb, wb = x[idx], y[idx], (None if w is None else w[idx])
                loss = self._train_step(xb, yb, wb)
                tot += float(loss) * len(idx)
                cnt += len(idx)
            hist.append(F32(tot / cnt))
        return History(hist)


class Sequential(Model):
    def __init__(self, layer_list):
        inp = layer_list[0]
        assert isinstance(inp, Input)
        body = list(layer_list[1:])
        fan_in = int(np.prod(inp.shape))
        for l in body:
            if isinstance(l, Dense):
                l.build(fan_in, _init_rng)
                fan_in = l.units
        super().__init__(_layers=body)
        self._input_shape = inp.shape
        handle = Input(inp.shape)
        handle.parent_layers = self.layers
        self.inputs = [handle]
