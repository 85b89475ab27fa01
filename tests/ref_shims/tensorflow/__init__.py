"""Numpy-backed `tensorflow` stub (TEST INFRASTRUCTURE).

Lets tests execute the reference's Python sources verbatim from
/root/reference in a container without TensorFlow.  Only the ops the
reference calls are provided (agents/resilient_CAC_agents.py:50-56,
training/train_agents.py:89-98).  Semantics: tf.sort -> np.sort,
tf.clip_by_value(x, lo, hi) -> max(min(x, hi), lo), tf.reduce_mean -> fp32
ordered mean.  Never imported by the product package.
"""
import numpy as np
from . import keras  # noqa: F401  (tf.keras -> oracle.keras_np)
from oracle.keras_np import as_tensor

float32 = np.float32


def convert_to_tensor(x, dtype=None):
    return as_tensor(np.asarray(x, dtype=dtype))


def concat(values, axis):
    return as_tensor(np.concatenate([np.asarray(v) for v in values], axis=axis))


def zeros(shape, dtype=np.float32):
    return as_tensor(np.zeros(shape, dtype=dtype))


def sort(x, axis=-1):
    return as_tensor(np.sort(np.asarray(x), axis=axis))


def clip_by_value(x, lo, hi):
    return as_tensor(np.maximum(np.minimum(np.asarray(x), np.asarray(hi)), np.asarray(lo)))


def reduce_mean(x, axis=None):
    x = np.asarray(x)
    return as_tensor(np.mean(x, axis=axis, dtype=x.dtype))


class _Math:
    @staticmethod
    def minimum(a, b):
        return as_tensor(np.minimum(np.asarray(a), np.asarray(b)))

    @staticmethod
    def maximum(a, b):
        return as_tensor(np.maximum(np.asarray(a), np.asarray(b)))

    @staticmethod
    def reduce_sum(x, axis=None):
        x = np.asarray(x)
        return as_tensor(np.sum(x, axis=axis, dtype=x.dtype))

    @staticmethod
    def square(x):
        return as_tensor(np.square(np.asarray(x)))


math = _Math()


class _Random:
    @staticmethod
    def set_seed(seed):
        keras.set_init_seed(seed)


random = _Random()


class _Logger:
    def setLevel(self, level):
        pass


def get_logger():
    return _Logger()
