"""Sparse GP regression with the collapsed (Titsias 2009) bound.

Mirrors gpflowSlim/models/sgpr.py:85-189 (SGPR.__init__, _build_likelihood, _build_predict).  All
O(M^2 N) work -- Kuu potrf, (L^-1 Kuf), A A^T, the second potrf and the solves -- runs in gps_sgpr on
the GPU; Kuf ([M, N], 32 GB at M = 4096, N = 10^6) never leaves HBM.  GPRFITC and the upper bound
(sgpr.py:30-82, 192-326) are not mirrored.
"""
import numpy as np

from .. import features
from .. import likelihoods
from .. import _backend as be
from .._settings import settings
from .model import GPModel


class SGPR(GPModel):
    def __init__(self, X, Y, kern, feat=None, mean_function=None, Z=None, obs_var=0.1, num_data=None,
                 num_latent=None, **kwargs):
        X = np.ascontiguousarray(X, dtype=settings.float_type)
        Y = np.ascontiguousarray(Y, dtype=settings.float_type)
        likelihood = likelihoods.Gaussian(var=obs_var)
        GPModel.__init__(self, X, Y, kern, likelihood, mean_function, **kwargs)
        self.feature = features.inducingpoint_wrapper(feat, Z)
        self.num_data = X.shape[0] if num_data is None else num_data
        self.num_latent = Y.shape[1] if num_latent is None else num_latent
        self._parameters = self._parameters + [self.feature._Z]

    def _call(self, Xnew=None, full_cov=False, want_bound=True):
        err = np.ascontiguousarray(self.Y - self.mean_function(self.X))
        prog = self.kern._program(self.X.shape[1])
        return be.get_handle().sgpr(prog, self.feature.Z, self.X, err, settings.numerics.jitter_level,
                                    float(np.squeeze(self.likelihood.variance)), Xnew=Xnew, full_cov=full_cov,
                                    want_bound=want_bound)

    def _build_likelihood(self):
        """models/sgpr.py:121-153"""
        bound, _, _ = self._call()
        return bound

    def _build_predict(self, Xnew, full_cov=False):
        """models/sgpr.py:155-189"""
        Xnew = np.ascontiguousarray(Xnew, dtype=settings.float_type)
        _, mean, var = self._call(Xnew=Xnew, full_cov=full_cov, want_bound=False)
        R = self.Y.shape[1]
        if full_cov:
            var = np.tile(var[:, :, None], [1, 1, R])
        else:
            var = np.tile(var[:, None], [1, R])
        return mean + self.mean_function(Xnew), var
