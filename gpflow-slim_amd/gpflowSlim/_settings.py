"""Minimal settings object: the two knobs the exact-GP path reads.

Mirrors gpflowSlim/_settings.py:25-35,83-89 and gpflowSlim/gpflowrc:6-11, except that the
default float type is float64 (the reference ships float32 and switches to float64 through a
user gpflowrc; the MI355X path computes in fp64 only).
"""
import contextlib
import copy

import numpy as np


class _Section(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __repr__(self):
        return "_Section(%r)" % (self.__dict__,)


class _Settings(object):
    def __init__(self):
        self.dtypes = _Section(float_type=np.float64, int_type=np.int32)
        self.numerics = _Section(jitter_level=1e-6, ekern_quadrature="warn")

    # reference spellings (gpflowSlim/_settings.py:83-95)
    @property
    def float_type(self):
        return self.dtypes.float_type

    @property
    def np_float(self):
        return self.dtypes.float_type

    @property
    def int_type(self):
        return self.dtypes.int_type

    @property
    def jitter(self):
        return self.numerics.jitter_level

    def set_jitter(self, value):
        """gpflowSlim/_settings.py:52-53"""
        self.numerics.jitter_level = float(value)

    def get_settings(self):
        """gpflowSlim/_settings.py:55-60: a mutable copy for temp_settings."""
        return copy.deepcopy(self)

    @contextlib.contextmanager
    def temp_settings(self, tmp):
        """gpflowSlim/_settings.py:46-47,62-71"""
        saved_d, saved_n = self.dtypes, self.numerics
        self.dtypes, self.numerics = tmp.dtypes, tmp.numerics
        try:
            yield
        finally:
            self.dtypes, self.numerics = saved_d, saved_n


settings = _Settings()
