"""Inducing features: InducingPoints and the conditional dispatch.

Mirrors gpflowSlim/features.py:26-81 (InducingFeature / InducingPoints), :153-174 (dispatch),
:177-193 (inducingpoint_wrapper).
"""
from functools import singledispatch

import numpy as np

from . import conditionals
from ._settings import settings
from .params import Parameter


class InducingFeature(object):
    def __len__(self):
        raise NotImplementedError()


class InducingPoints(InducingFeature):
    def __init__(self, Z):
        self._Z = Parameter(np.asarray(Z, dtype=settings.float_type))

    @property
    def Z(self):
        return self._Z.value

    def __len__(self):
        return self.Z.shape[0]

    def Kuu(self, kern, jitter=0.0):
        """features.py:74-77"""
        Kzz = kern.K(self.Z)
        Kzz += jitter * np.eye(len(self), dtype=settings.dtypes.float_type)
        return Kzz

    def Kuf(self, kern, Xnew):
        """features.py:79-81"""
        return kern.K(self.Z, Xnew)


@singledispatch
def conditional(feat, kern, Xnew, f, *, full_cov=False, q_sqrt=None, white=False):
    raise NotImplementedError("No implementation for {} found".format(type(feat).__name__))


@conditional.register(InducingPoints)
def default_feature_conditional(feat, kern, Xnew, f, *, full_cov=False, q_sqrt=None, white=False):
    """features.py:162-174"""
    return conditionals.feature_conditional(Xnew, feat, kern, f, full_cov=full_cov, q_sqrt=q_sqrt, white=white)


def inducingpoint_wrapper(feat, Z):
    """features.py:177-193"""
    if feat is not None and Z is not None:
        raise ValueError("Cannot pass both an InducingFeature instance and Z values")
    elif feat is None and Z is None:
        raise ValueError("You must pass either an InducingFeature instance or Z values")
    elif Z is not None:
        feat = InducingPoints(Z)
    elif isinstance(feat, np.ndarray):
        feat = InducingPoints(feat)
    else:
        assert isinstance(feat, InducingFeature)
    return feat
