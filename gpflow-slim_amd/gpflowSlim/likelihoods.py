"""Gaussian likelihood (the only one on the exact-GP path).

Mirrors gpflowSlim/likelihoods.py:158-188.
"""
import numpy as np

from . import densities
from . import transforms
from ._settings import settings
from .params import Parameter


class Likelihood(object):
    def __init__(self, name='likelihood'):
        self._name = name
        self._parameters = []

    @property
    def name(self):
        return self._name

    @property
    def parameters(self):
        return self._parameters


class Gaussian(Likelihood):
    def __init__(self, var=1.0, min_var=None):
        super().__init__()
        trans = transforms.positive if min_var is None else transforms.Log1pe(min_var)   # :162
        self._variance = Parameter(var, transform=trans, dtype=settings.float_type, name='variance')
        self._parameters = self._parameters + [self._variance]

    @property
    def variance(self):
        return self._variance.value

    def logp(self, F, Y):
        return densities.gaussian(F, Y, self.variance)

    def conditional_mean(self, F):
        return np.array(F, copy=True)

    def conditional_variance(self, F):
        return np.full(np.shape(F), np.squeeze(self.variance))

    def predict_mean_and_var(self, Fmu, Fvar):
        """likelihoods.py:180-181"""
        return np.array(Fmu, copy=True), Fvar + self.variance

    def predict_density(self, Fmu, Fvar, Y):
        """likelihoods.py:183-184"""
        return densities.gaussian(Fmu, Y, Fvar + self.variance)

    def variational_expectations(self, Fmu, Fvar, Y):
        """likelihoods.py:186-188"""
        return -0.5 * np.log(2 * np.pi) - 0.5 * np.log(self.variance) \
               - 0.5 * (np.square(Y - Fmu) + Fvar) / self.variance
