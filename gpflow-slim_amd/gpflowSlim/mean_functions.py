"""Mean functions: Zero (default), Constant, Linear.

Mirrors gpflowSlim/mean_functions.py:24-100.  Host-side [N, R] arrays; the device only ever
sees the residual Y - m(X).
"""
import numpy as np

from ._settings import settings
from .params import Parameter


class MeanFunction(object):
    def __init__(self, name='MeanFunction'):
        self._parameters = []
        self._name = name

    def __call__(self, X):
        raise NotImplementedError("Implement the __call__ method for this mean function")

    @property
    def parameters(self):
        return self._parameters

    @property
    def name(self):
        return self._name


class Zero(MeanFunction):
    """mean_functions.py:57-59: zeros [N, 1] (broadcasts against Y [N, R])"""

    def __call__(self, X):
        return np.zeros((np.shape(X)[0], 1), dtype=settings.float_type)


class Constant(MeanFunction):
    """mean_functions.py:84-100: y_i = c"""

    def __init__(self, c=None, name='Constant'):
        MeanFunction.__init__(self, name)
        c = np.zeros(1) if c is None else c
        self.c = Parameter(c, name='c')
        self._parameters = self._parameters + [self.c]

    def __call__(self, X):
        return np.tile(np.reshape(self.c.value, (1, -1)), (np.shape(X)[0], 1))


class Linear(MeanFunction):
    """mean_functions.py:62-81: y_i = A x_i + b"""

    def __init__(self, A=None, b=None, name='Linear'):
        MeanFunction.__init__(self, name)
        A = np.ones((1, 1)) if A is None else A
        b = np.zeros(1) if b is None else b
        self.A = Parameter(np.atleast_2d(A), name='A')
        self.b = Parameter(b, name='b')
        self._parameters = self._parameters + [self.A, self.b]

    def __call__(self, X):
        return np.matmul(np.asarray(X, dtype=settings.float_type), self.A.value) + self.b.value
