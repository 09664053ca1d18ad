"""Log-densities on the exact-GP path.

Mirrors gpflowSlim/densities.py: gaussian :24-25, multivariate_normal :73-95.  The triangular
solve of multivariate_normal runs on the GPU (csrc: recursive trsm); the scalar reductions of
an [N, R] host array stay on the host.
"""
import numpy as np

from . import _backend as be


def gaussian(x, mu, var):
    """densities.py:24-25"""
    return -0.5 * (np.log(2 * np.pi) + np.log(var) + np.square(mu - x) / var)


def multivariate_normal(x, mu, L):
    """densities.py:73-95.  L is the Cholesky factor of the covariance; x, mu vectors or [N, R]
    matrices (columns independent)."""
    x = np.asarray(x, dtype=np.float64)
    d = x - mu
    alpha = be.get_handle().trsm_lower(L, d, trans=False)
    num_col = 1 if x.ndim == 1 else x.shape[1]
    num_dims = x.shape[0]
    ret = -0.5 * num_dims * num_col * np.log(2 * np.pi)
    ret += -num_col * np.sum(np.log(np.diag(L)))
    ret += -0.5 * np.sum(np.square(alpha))
    return ret
