"""Sparse GP regression with the collapsed (Titsias 2009) bound.

Mirrors gpflowSlim/models/sgpr.py:85-189 (SGPR.__init__, _build_likelihood, _build_predict).  All
O(M^2 N) work -- Kuu potrf, (L^-1 Kuf), A A^T, the second potrf and the solves -- runs in gps_sgpr on
the GPU; Kuf ([M, N], 32 GB at M = 4096, N = 10^6) never leaves HBM.  GPRFITC (sgpr.py:190-326) runs through
gps_fitc (the same pipeline with per-point weights 1/nu_i), and SGPRUpperMixin.compute_upper_bound
(sgpr.py:30-82) is assembled from the terms of two such device evaluations.
"""
import numpy as np

from .. import features
from .. import likelihoods
from .. import _backend as be
from .._settings import settings
from .model import GPModel


class SGPRUpperMixin(object):
    """Upper bound for the GP regression marginal likelihood (models/sgpr.py:30-82, Titsias 2014)."""

    def compute_upper_bound(self):
        """models/sgpr.py:55-82.  With A = L^-1 Kuf:  chol(Kuu + s^-2 Kuf Kuf^T) = L chol(I + s^-2 A A^T), so
        logdet = -sum log diag(LB) and |v|^2 = |LB_c^-1 A Y|^2 / s_c^4 -- both are terms of gps_sgpr, evaluated once
        at the likelihood variance and once at the corrected noise (with Y, not Y - m(X), as the reference does)."""
        h = be.get_handle()
        prog = self.kern._program(self.X.shape[1])
        Y = np.ascontiguousarray(self.Y)
        num_data = float(self.Y.shape[0])
        s2 = float(np.squeeze(self.likelihood.variance))
        jit = settings.numerics.jitter_level
        h.sgpr(prog, self.feature.Z, self.X, Y, jit, s2)
        t = h.sparse_last_terms()
        c = num_data * t["kdiag"] - t["tr_AAT"]                       # trace bound, sgpr.py:67
        corrected_noise = s2 + c
        const = -0.5 * num_data * np.log(2 * np.pi * s2)
        logdet = -t["sum_log_diag_LB"]
        h.sgpr(prog, self.feature.Z, self.X, Y, jit, corrected_noise)
        tc = h.sparse_last_terms()
        quad = -0.5 / corrected_noise * np.sum(Y ** 2.0) + 0.5 * tc["sum_c2"]
        return const + logdet + quad


class SGPR(GPModel, SGPRUpperMixin):
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

    def _call(self, Xnew=None, full_cov=False, want_bound=True, X=None, Y=None, handle=None):
        # (X, Y, handle: one rank's shard of the data points on its own handle -- gpflowSlim.distributed_sparse)
        X = self.X if X is None else X
        Y = self.Y if Y is None else Y
        err = np.ascontiguousarray(Y - self.mean_function(X))
        prog = self.kern._program(X.shape[1])
        return (handle or be.get_handle()).sgpr(prog, self.feature.Z, X, err, settings.numerics.jitter_level,
                                                float(np.squeeze(self.likelihood.variance)), Xnew=Xnew, full_cov=full_cov,
                                                want_bound=want_bound)

    def _build_likelihood(self):
        """models/sgpr.py:121-153"""
        bound, _, _ = self._call()
        return bound

    _fitc_gradient = False

    def compute_log_likelihood_and_gradients(self):
        """The collapsed bound and d bound / d(unconstrained parameter) for every parameter of the model (kernel, noise,
        mean function, inducing inputs -- Z is a Parameter of the feature, features.py:65): what TF autodiff through
        models/sgpr.py:121-153 yields.  Returns (bound, [(Parameter, gradient shaped like its unconstrained value), ...])."""
        d_all = self.X.shape[1]
        prog = self.kern._program(d_all)
        layout = self.kern._grad_layout(d_all)
        err = np.ascontiguousarray(self.Y - self.mean_function(self.X))
        zparam = getattr(self.feature, "_Z", None)
        want_z = zparam is not None and any(p is zparam for p in self.parameters)
        bound, slots, gnoise, g_mean, g_Z = be.get_handle().sgpr_grad(
            prog, self.feature.Z, self.X, err, settings.numerics.jitter_level, float(np.squeeze(self.likelihood.variance)),
            want_grad_Z=want_z, fitc=self._fitc_gradient)
        if len(layout) != len(slots):
            raise RuntimeError("gradient slot layout mismatch: %d vs %d" % (len(layout), len(slots)))
        grads = {id(p): np.zeros_like(np.atleast_1d(p.vf_val), dtype=settings.float_type) for p in self.parameters}
        for (param, idx), g in zip(layout, slots):
            if param is None:
                continue
            if idx is None:
                grads[id(param)] += g
            else:
                grads[id(param)].reshape(-1)[idx] += g
        grads[id(self.likelihood._variance)] += gnoise
        from ..mean_functions import Constant as _MConst, Linear as _MLin
        mf = self.mean_function

        def _fit(g, like):
            like = np.atleast_1d(like)
            return g.reshape(like.shape) if g.size == like.size else np.full(like.shape, np.sum(g))

        if isinstance(mf, _MConst):
            grads[id(mf.c)] = grads[id(mf.c)] + _fit(np.sum(g_mean, axis=0), mf.c.vf_val)
        elif isinstance(mf, _MLin):
            grads[id(mf.A)] = grads[id(mf.A)] + _fit(self.X.T @ g_mean, mf.A.vf_val)
            grads[id(mf.b)] = grads[id(mf.b)] + _fit(np.sum(g_mean, axis=0), mf.b.vf_val)
        out = []
        for p in self.parameters:
            if p is zparam:
                out.append((p, (g_Z * np.atleast_1d(p.transform.forward_grad(p.vf_val))).reshape(p.vf_val.shape)))
            else:
                g = grads[id(p)].reshape(np.atleast_1d(p.vf_val).shape) * np.atleast_1d(p.transform.forward_grad(p.vf_val))
                out.append((p, g.reshape(p.vf_val.shape)))
        return bound, out

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


class GPRFITC(GPModel, SGPRUpperMixin):
    """GP regression with the FITC approximation (models/sgpr.py:190-326, Snelson & Ghahramani 2006)."""

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

    def _call(self, Xnew=None, full_cov=False, want_bound=True, X=None, Y=None, handle=None):
        X = self.X if X is None else X
        Y = self.Y if Y is None else Y
        err = np.ascontiguousarray(Y - self.mean_function(X))
        prog = self.kern._program(X.shape[1])
        return (handle or be.get_handle()).sgpr(prog, self.feature.Z, X, err, settings.numerics.jitter_level,
                                                float(np.squeeze(self.likelihood.variance)), Xnew=Xnew, full_cov=full_cov,
                                                want_bound=want_bound, fitc=True)

    def _build_likelihood(self):
        """models/sgpr.py:252-291"""
        bound, _, _ = self._call()
        return bound

    # the FITC log-likelihood and its gradient (gps_fitc_grad): same assembly as SGPR's
    _fitc_gradient = True
    compute_log_likelihood_and_gradients = SGPR.compute_log_likelihood_and_gradients

    def _build_predict(self, Xnew, full_cov=False):
        """models/sgpr.py:293-318"""
        Xnew = np.ascontiguousarray(Xnew, dtype=settings.float_type)
        _, mean, var = self._call(Xnew=Xnew, full_cov=full_cov, want_bound=False)
        R = self.num_latent
        if full_cov:
            var = np.tile(var[:, :, None], [1, 1, R])
        else:
            var = np.tile(var[:, None], [1, R])
        return mean + self.mean_function(Xnew), var

    @property
    def Z(self):
        raise NotImplementedError("Inducing points are now in `model.feature.Z`.")
