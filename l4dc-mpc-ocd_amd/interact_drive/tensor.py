"""A numpy stand-in for the tf.Tensor / tf.Variable values the reference API hands around."""
import numpy as np


class Tensor(np.ndarray):
    """float32 ndarray with ``.numpy()`` and ``.assign()`` like tf.Tensor / tf.Variable."""

    def __new__(cls, value, dtype=np.float32):
        return np.array(value, dtype=dtype).view(cls)

    def numpy(self):
        a = np.asarray(self)
        return a[()] if a.ndim == 0 else a

    def assign(self, value):
        self[...] = np.asarray(value, dtype=self.dtype)
        return self


def constant(value, dtype=np.float32) -> Tensor:
    return Tensor(value, dtype=dtype)
