"""`tensorflow.keras` stub -> oracle.keras_np (TEST INFRASTRUCTURE)."""
from oracle.keras_np import (Input, Model, Sequential, layers, optimizers, losses,  # noqa: F401
                             set_shuffle_stream, set_init_seed)
