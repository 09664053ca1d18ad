"""Parameter transforms (host-side fp64 scalar math).

Mirrors gpflowSlim/transforms.py: Identity :40-57, Log1pe :117-181, Exp :80-114, positive :377.
Hyper-parameters are stored unconstrained and read through ``forward`` on every use, exactly as
the reference does (params.py:142-145,164-166), so e.g. ``variance=1.0`` round-trips through
softplus(.)+1e-6 before it reaches the device.
"""
import numpy as np

from ._settings import settings


class Transform(object):
    def forward(self, x):
        raise NotImplementedError

    def backward(self, y):
        raise NotImplementedError

    # the reference's graph-mode name; values here are eager numpy
    def forward_tensor(self, x):
        return self.forward(x)

    def log_jacobian_tensor(self, x):
        raise NotImplementedError


class Identity(Transform):
    """transforms.py:40-57"""

    def forward(self, x):
        return np.asarray(x, dtype=settings.float_type)

    def backward(self, y):
        return np.asarray(y, dtype=settings.float_type)

    def log_jacobian_tensor(self, x):
        return 0.0

    def __str__(self):
        return '(none)'


class Exp(Transform):
    """transforms.py:80-114: y = exp(x) + lower"""

    def __init__(self, lower=1e-6):
        self._lower = lower

    def forward(self, x):
        return np.exp(np.asarray(x, dtype=settings.float_type)) + self._lower

    def backward(self, y):
        return np.log(np.asarray(y, dtype=settings.float_type) - self._lower)

    def log_jacobian_tensor(self, x):
        return float(np.sum(x))

    def __str__(self):
        return '+ve'


class Log1pe(Transform):
    """transforms.py:117-181: y = softplus(x) + lower"""

    def __init__(self, lower=1e-6):
        self._lower = lower

    def forward(self, x):
        # tf.nn.softplus (transforms.py:145-146); logaddexp is the overflow-safe form
        return np.logaddexp(0.0, np.asarray(x, dtype=settings.float_type)) + self._lower

    def backward(self, y):
        # transforms.py:177-178
        ys = np.maximum(np.asarray(y, dtype=settings.float_type) - self._lower,
                        np.finfo(settings.float_type).eps)
        return ys + np.log(-np.expm1(-ys))

    def log_jacobian_tensor(self, x):
        # transforms.py:148-149
        return float(-np.sum(np.logaddexp(0.0, -np.asarray(x, dtype=settings.float_type))))

    def __str__(self):
        return '+ve'


positive = Log1pe()      # transforms.py:377
