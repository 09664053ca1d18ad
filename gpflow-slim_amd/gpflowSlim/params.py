"""Parameter: an unconstrained fp64 host array read through its transform.

Mirrors gpflowSlim/params.py:131-194 without the TensorFlow variable: ``vf_val`` is a numpy
array, ``value`` applies ``transform.forward`` on every read (params.py:164-166), and
``assign`` / ``assign_unconstrained`` are what an optimiser writes to.
"""
import numpy as np

from ._settings import settings
from .transforms import Identity


class Parameter(object):
    def __init__(self, value, transform=None, prior=None, trainable=True, dtype=None, name='Param'):
        self.instance_name = name
        if transform is None:
            transform = Identity()
        self.prior = prior
        self.transform = transform
        self.trainable = trainable
        # params.py:142-145 (the dtype argument is ignored there as well)
        self.vf_val = np.array(self.transform.backward(value), dtype=settings.float_type)

    @property
    def name(self):
        return self.instance_name

    @property
    def shape(self):
        return self.vf_val.shape

    @property
    def dtype(self):
        return self.vf_val.dtype

    @property
    def size(self):
        return int(self.vf_val.size)

    @property
    def value(self):
        return self.transform.forward(self.vf_val)

    @property
    def unconstrained_tensor(self):
        return self.vf_val

    @property
    def constrained_tensor(self):
        return self.value

    def assign(self, value):
        """Set the constrained value."""
        self.vf_val = np.array(self.transform.backward(value), dtype=settings.float_type)

    def assign_unconstrained(self, x):
        self.vf_val = np.array(x, dtype=settings.float_type).reshape(self.vf_val.shape)

    def _build_prior(self, unconstrained_tensor, constrained_tensor):
        """params.py:176-194: log p(constrained) + log|d constrained / d unconstrained|"""
        if self.prior is None:
            return 0.0
        log_jacobian = self.transform.log_jacobian_tensor(unconstrained_tensor)
        logp_var = self.prior.logp(constrained_tensor)
        return float(np.squeeze(np.sum(logp_var) + log_jacobian))
