"""Sparse variational GP (Hensman et al. 2015) on top of the device conditional.

Mirrors gpflowSlim/models/svgp.py:30-130: ELBO = sum of variational expectations (rescaled for
mini-batches) - KL[q(u) || p(u)].  With the Gaussian likelihood the whole bound is one device call
(gps_svgp_elbo: Kuu potrf + Kuf trsm of BASELINE config 5, the q_sqrt products, the reduction of the
expectations and the KL on the same factor); any other likelihood object with a
``variational_expectations`` goes through conditional() + gauss_kl() like the reference.
"""
import numpy as np

from .. import _backend as be
from .. import features
from .. import kullback_leiblers
from .. import likelihoods
from .. import transforms
from .._settings import settings
from ..params import Parameter
from .model import GPModel


class SVGP(GPModel):
    def __init__(self, X, Y, kern, likelihood, feat=None, mean_function=None, num_latent=None, q_diag=False,
                 whiten=True, minibatch_size=None, Z=None, num_data=None, **kwargs):
        X = np.ascontiguousarray(X, dtype=settings.float_type)
        Y = np.ascontiguousarray(Y, dtype=settings.float_type)
        GPModel.__init__(self, X, Y, kern, likelihood, mean_function, **kwargs)
        self.num_data = num_data or X.shape[0]
        self.q_diag, self.whiten = q_diag, whiten
        self.feature = features.inducingpoint_wrapper(feat, Z)
        self.num_latent = num_latent or Y.shape[1]
        num_inducing = len(self.feature)
        self._q_mu = Parameter(np.zeros((num_inducing, self.num_latent), dtype=settings.float_type), name='q_mu')
        if self.q_diag:
            self._q_sqrt = Parameter(np.ones((num_inducing, self.num_latent), dtype=settings.float_type),
                                     transforms.positive, name='q_sqrt')
        else:
            q_sqrt = np.array([np.eye(num_inducing, dtype=settings.float_type)
                               for _ in range(self.num_latent)]).swapaxes(0, 2)
            self._q_sqrt = Parameter(q_sqrt, transform=transforms.LowerTriangular(num_inducing, self.num_latent),
                                     name='q_sqrt')
        self._parameters = self._parameters + [self._q_mu, self._q_sqrt]

    @property
    def q_mu(self):
        return self._q_mu.value

    @property
    def q_sqrt(self):
        return self._q_sqrt.value

    def build_prior_KL(self):
        """models/svgp.py:101-106"""
        if self.whiten:
            K = None
        else:
            K = self.feature.Kuu(self.kern, jitter=settings.numerics.jitter_level)
        return kullback_leiblers.gauss_kl(self.q_mu, self.q_sqrt, K)

    def _build_likelihood(self):
        """models/svgp.py:108-125"""
        scale = float(self.num_data) / float(self.X.shape[0])
        if type(self.likelihood) is likelihoods.Gaussian:
            yres = self.Y - self.mean_function(self.X)
            yres = np.ascontiguousarray(np.broadcast_to(yres, self.Y.shape))
            elbo, _, _ = be.get_handle().svgp_elbo(self.kern._program(self.X.shape[1]), self.feature.Z, self.X, yres,
                                                   self.q_mu, self.q_sqrt, settings.numerics.jitter_level,
                                                   float(np.squeeze(self.likelihood.variance)), white=self.whiten,
                                                   scale=scale)
            return elbo
        KL = self.build_prior_KL()
        fmean, fvar = self._build_predict(self.X, full_cov=False)
        var_exp = self.likelihood.variational_expectations(fmean, fvar, self.Y)
        return float(np.sum(var_exp) * scale - KL)

    def _build_predict(self, Xnew, full_cov=False):
        """models/svgp.py:127-130"""
        mu, var = features.conditional(self.feature, self.kern, Xnew, self.q_mu, q_sqrt=self.q_sqrt,
                                       full_cov=full_cov, white=self.whiten)
        return mu + self.mean_function(Xnew), var
