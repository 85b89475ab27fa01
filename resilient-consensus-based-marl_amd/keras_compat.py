"""Minimal Keras-shaped model objects for the drop-in API.

The reference builds its networks with ``tensorflow.keras`` (main.py:59-82) and
the agent classes touch a small duck-typed surface of those models (SURVEY.md
8b): ``output_shape``, ``__call__``, ``predict``, ``get_weights/set_weights``,
``layers[-1].get_weights/set_weights``, ``inputs``, ``layers[-2].output``,
``compile``, ``trainable``.  TensorFlow is not a dependency of this package, so
``main.py`` builds its three MLPs from these classes instead; real Keras models
with the same architecture work too (the agents only use the duck type and
read/write weights through ``get_weights/set_weights``).

Forward passes run on the GPU through the C-ABI (single.RowOps); weights are
NumPy fp32 arrays in Keras order [W1[in,h], b1[h], W2[h,h], b2[h], W3[h,out], b3[out]].
"""
import numpy as np

from . import single

_rng = np.random.default_rng(0)


def set_seed(seed):
    """Seeds the Glorot initialiser (the role of tf.random.set_seed at main.py:47)."""
    global _rng
    _rng = np.random.default_rng(int(seed))


class LeakyReLU:
    def __init__(self, alpha=0.3):
        self.alpha = float(alpha)


class Input:
    def __init__(self, shape=None):
        self.shape = tuple(shape)


class Flatten:
    def get_weights(self):
        return []

    def set_weights(self, w):
        assert len(w) == 0


class Dense:
    """Keras Dense: y = act(x @ kernel + bias); Glorot-uniform kernel, zero bias."""

    def __init__(self, units, activation=None):
        self.units, self.activation = int(units), activation
        self.kernel = self.bias = None
        self.output = self                      # `model.layers[-2].output` is only used as a handle

    def build(self, fan_in):
        lim = np.sqrt(6.0 / (fan_in + self.units))
        self.kernel = _rng.uniform(-lim, lim, size=(fan_in, self.units)).astype(np.float32)
        self.bias = np.zeros(self.units, np.float32)

    def get_weights(self):
        return [self.kernel.copy(), self.bias.copy()]

    def set_weights(self, w):
        k, b = np.asarray(w[0], np.float32), np.asarray(w[1], np.float32)
        assert k.shape == self.kernel.shape and b.shape == self.bias.shape, (k.shape, self.kernel.shape)
        self.kernel, self.bias = k.copy(), b.copy()


class Sequential:
    """Input -> Flatten -> Dense(20, LeakyReLU(0.1)) -> Dense(20, LeakyReLU(0.1)) -> Dense(out[, softmax])."""

    def __init__(self, layer_list):
        inp = [l for l in layer_list if isinstance(l, Input)]
        self.layers = [l for l in layer_list if not isinstance(l, Input)]
        dense = [l for l in self.layers if isinstance(l, Dense)]
        if len(inp) != 1 or len(dense) != 3:
            raise ValueError("expected Input, Flatten and three Dense layers (main.py:59-82)")
        # two equally wide LeakyReLU(0.1) hidden layers.  20 units (main.py:59-82) run everywhere; other widths are for
        # critics trained through train_RPBCAC (dense-GEMM path of the engine) -- the per-agent method views
        # (model(x), fit, ...) are compiled for 20 units and raise RcmarlError(UNSUPPORTED) otherwise
        for l in dense[:2]:
            if not isinstance(l.activation, LeakyReLU) or abs(l.activation.alpha - 0.1) > 1e-12 or l.units != dense[0].units:
                raise ValueError("hidden layers must be two Dense(h, LeakyReLU(alpha=0.1)) of equal width h")
        if dense[2].activation not in (None, "softmax", "linear"):
            raise ValueError("output activation must be None or 'softmax'")
        self.inputs = inp
        self.input_shape = (None,) + inp[0].shape
        fan = int(np.prod(inp[0].shape))
        for l in dense:
            l.build(fan)
            fan = l.units
        self._dense = dense
        self.trainable = True
        self.optimizer = self.loss = None

    @property
    def output_shape(self):
        return (None, self._dense[2].units)

    def get_weights(self):
        return [w for l in self._dense for w in l.get_weights()]

    def set_weights(self, weights):
        assert len(weights) == 6, "expected [W1,b1,W2,b2,W3,b3]"
        for k, l in enumerate(self._dense):
            l.set_weights(weights[2 * k:2 * k + 2])

    def compile(self, optimizer=None, loss=None):
        self.optimizer, self.loss = optimizer, loss

    def __call__(self, x):
        w = self.get_weights()
        if self._dense[2].activation == "softmax":
            return single.get_ops().policy(w, x)
        return single.get_ops().value(w, x)

    def predict(self, x):
        return self(x)


class FeatureModel:
    """``keras.Model(model.inputs, model.layers[-2].output)``: shares the hidden layers (agents/...:39-40)."""

    def __init__(self, inputs, outputs):
        self._model = None
        self.inputs, self.outputs = inputs, outputs
        self.trainable = True

    def bind(self, model):
        self._model = model
        return self

    def get_weights(self):
        return self._model.get_weights()[:4]

    def set_weights(self, w):
        assert len(w) == 4
        self._model.set_weights(list(w) + self._model.get_weights()[4:])


def Model(inputs, outputs):
    return FeatureModel(inputs, outputs)


class _NS:
    pass


layers = _NS()
layers.Dense, layers.Flatten, layers.LeakyReLU, layers.Input = Dense, Flatten, LeakyReLU, Input
