"""GP conditionals on the GPU.

Mirrors gpflowSlim/conditionals.py: conditional :24-66, feature_conditional :69-77,
base_conditional :80-121.  ``conditional`` / ``feature_conditional`` build Kmm (+jitter), Kmn and
Knn on the device and never move them over PCIe; ``base_conditional`` accepts host matrices.
"""
import numpy as np

from . import _backend as be
from ._settings import settings


def conditional(Xnew, X, kern, f, *, full_cov=False, q_sqrt=None, white=False):
    """conditionals.py:24-66"""
    Xnew = np.asarray(Xnew, dtype=settings.float_type)
    X = np.asarray(X, dtype=settings.float_type)
    prog = kern._program(X.shape[1])
    return be.get_handle().conditional(prog, X, Xnew, f, settings.numerics.jitter_level,
                                       q_sqrt=q_sqrt, white=white, full_cov=full_cov)


def feature_conditional(Xnew, feat, kern, f, *, full_cov=False, q_sqrt=None, white=False):
    """conditionals.py:69-77 with features.InducingPoints.Kuu/Kuf (features.py:74-81)"""
    return conditional(Xnew, feat.Z, kern, f, full_cov=full_cov, q_sqrt=q_sqrt, white=white)


def base_conditional(Kmn, Kmm, Knn, f, *, full_cov=False, q_sqrt=None, white=False):
    """conditionals.py:80-121"""
    return be.get_handle().base_conditional(Kmn, Kmm, Knn, f, q_sqrt=q_sqrt, white=white, full_cov=full_cov)
