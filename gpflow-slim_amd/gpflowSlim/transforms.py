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

    def forward_grad(self, x):
        """d forward(x) / d x, elementwise (chain rule from constrained to unconstrained gradients)."""
        raise NotImplementedError


class Identity(Transform):
    """transforms.py:40-57"""

    def forward(self, x):
        return np.asarray(x, dtype=settings.float_type)

    def backward(self, y):
        return np.asarray(y, dtype=settings.float_type)

    def log_jacobian_tensor(self, x):
        return 0.0

    def forward_grad(self, x):
        return np.ones_like(np.asarray(x, dtype=settings.float_type))

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

    def forward_grad(self, x):
        return np.exp(np.asarray(x, dtype=settings.float_type))

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

    def forward_grad(self, x):
        # d softplus / dx = sigmoid(x)
        x = np.asarray(x, dtype=settings.float_type)
        return np.exp(-np.logaddexp(0.0, -x))

    def __str__(self):
        return '+ve'


class LowerTriangular(Transform):
    """transforms.py:294-372: free vector [num_matrices, N(N+1)/2] <-> lower-triangular [N, N, num_matrices]"""

    def __init__(self, N, num_matrices=1, squeeze=False):
        self.num_matrices = num_matrices
        self.squeeze = squeeze
        self.N = N

    def forward(self, x):
        x = np.asarray(x, dtype=settings.float_type)
        xr = np.reshape(x, (self.num_matrices, -1))
        L = xr.shape[1]
        matsize = int((L * 8 + 1) ** 0.5 * 0.5 - 0.5)
        if matsize * (matsize + 1) // 2 != L:
            raise ValueError("The free state must be a triangle number.")
        var = np.zeros((matsize, matsize, self.num_matrices), settings.float_type)
        flat = self._tril_flat(matsize)             # (cached: np.tril_indices is a third of this call at N = 4096)
        vf = var.reshape(matsize * matsize, self.num_matrices)
        for i in range(self.num_matrices):
            vf[flat, i] = xr[i, :]
        return var.squeeze() if self.squeeze else var

    def _tril_flat(self, n):
        cache = getattr(self, "_flat_cache", None)
        if cache is None or cache[0] != n:
            rows, cols = np.tril_indices(n, 0)
            cache = self._flat_cache = (n, rows * n + cols)
        return cache[1]

    def backward(self, y):
        y = np.asarray(y, dtype=settings.float_type)
        N = int(np.sqrt(y.size / self.num_matrices))
        reshaped = np.reshape(y, (N * N, self.num_matrices))
        return reshaped[self._tril_flat(N)].T

    def log_jacobian_tensor(self, x):
        return 0.0

    def __str__(self):
        return "LoTri->vec"


positive = Log1pe()      # transforms.py:377
