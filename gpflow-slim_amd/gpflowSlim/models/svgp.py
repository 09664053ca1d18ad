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
                 whiten=True, minibatch_size=None, Z=None, num_data=None, train_inducing=False, **kwargs):
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
        # The reference keeps Z (a Parameter of the feature, features.py:65) out of `model.parameters`, but its example
        # minimises the objective over every TF variable (examples/svgp.py:161), Z included.  train_inducing=True puts Z
        # into `parameters`, so that `optimize()` moves it too.
        self.train_inducing = bool(train_inducing)
        if self.train_inducing and getattr(self.feature, "_Z", None) is not None:
            self._parameters = self._parameters + [self.feature._Z]

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

    def _bound_on(self, X, Y, scale, handle=None):
        """The Gaussian-likelihood bound on the data points (X, Y) with the given mini-batch scale: one device call.
        (The whole data set for _build_likelihood; one rank's shard for gpflowSlim.distributed_sparse.)"""
        yres = np.ascontiguousarray(np.broadcast_to(Y - self.mean_function(X), Y.shape))
        elbo, _, _ = (handle or be.get_handle()).svgp_elbo(
            self.kern._program(X.shape[1]), self.feature.Z, X, yres, self.q_mu, self.q_sqrt,
            settings.numerics.jitter_level, float(np.squeeze(self.likelihood.variance)), white=self.whiten, scale=scale)
        return elbo

    def _build_likelihood(self):
        """models/svgp.py:108-125"""
        scale = float(self.num_data) / float(self.X.shape[0])
        if type(self.likelihood) is likelihoods.Gaussian:
            return self._bound_on(self.X, self.Y, scale)
        KL = self.build_prior_KL()
        fmean, fvar = self._build_predict(self.X, full_cov=False)
        var_exp = self.likelihood.variational_expectations(fmean, fvar, self.Y)
        return float(np.sum(var_exp) * scale - KL)

    def compute_log_likelihood_and_gradients(self):
        """The bound and d bound / d(unconstrained parameter) for every parameter of the model -- what
        `tf.gradients(objective, variables)` yields in the reference (examples/svgp.py:159-161) up to the sign of
        `objective`.  Gaussian likelihood, whitened or not; the inducing inputs move only with train_inducing=True.
        Returns (bound, [(Parameter, gradient array shaped like Parameter.unconstrained_tensor), ...])."""
        return self._bound_and_gradients_on(self.X, self.Y, float(self.num_data) / float(self.X.shape[0]))

    def _bound_and_gradients_on(self, X, Y, scale, handle=None):
        """compute_log_likelihood_and_gradients on the data points (X, Y) with the given scale (see _bound_on); every
        output is linear in the per-point terms, which is what lets ranks add their shards' results up."""
        if type(self.likelihood) is not likelihoods.Gaussian:
            raise NotImplementedError("analytic gradients of the SVGP bound need the Gaussian likelihood")
        d_all = X.shape[1]
        prog = self.kern._program(d_all)
        layout = self.kern._grad_layout(d_all)
        yres = np.ascontiguousarray(np.broadcast_to(Y - self.mean_function(X), Y.shape))
        zparam = getattr(self.feature, "_Z", None)
        want_z = zparam is not None and any(p is zparam for p in self.parameters)
        res = (handle or be.get_handle()).svgp_elbo_grad(
            prog, self.feature.Z, X, yres, self.q_mu, self.q_sqrt, settings.numerics.jitter_level,
            float(np.squeeze(self.likelihood.variance)), white=self.whiten, scale=scale, want_grad_Z=want_z)
        elbo, slots, gnoise, g_qmu, g_qsqrt, g_mean = res[:6]
        g_Z = res[6] if want_z else None
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
            grads[id(mf.A)] = grads[id(mf.A)] + _fit(X.T @ g_mean, mf.A.vf_val)
            grads[id(mf.b)] = grads[id(mf.b)] + _fit(np.sum(g_mean, axis=0), mf.b.vf_val)
        out = []
        for p in self.parameters:
            if p is self._q_mu:
                out.append((p, g_qmu.reshape(p.vf_val.shape)))
            elif p is self._q_sqrt:
                if self.q_diag:
                    out.append((p, (g_qsqrt * p.transform.forward_grad(p.vf_val)).reshape(p.vf_val.shape)))
                else:       # LowerTriangular: the free vector holds the lower-triangular entries, row-major per latent
                    n_ind = g_qsqrt.shape[0]
                    rows, cols = np.tril_indices(n_ind, 0)
                    out.append((p, np.stack([g_qsqrt[rows, cols, q] for q in range(g_qsqrt.shape[2])]).reshape(p.vf_val.shape)))
            elif p is zparam:
                out.append((p, (g_Z * np.atleast_1d(p.transform.forward_grad(p.vf_val))).reshape(p.vf_val.shape)))
            else:
                g = grads[id(p)].reshape(np.atleast_1d(p.vf_val).shape) * np.atleast_1d(p.transform.forward_grad(p.vf_val))
                out.append((p, g.reshape(p.vf_val.shape)))
        return elbo, out

    def _build_predict(self, Xnew, full_cov=False):
        """models/svgp.py:127-130"""
        mu, var = features.conditional(self.feature, self.kern, Xnew, self.q_mu, q_sqrt=self.q_sqrt,
                                       full_cov=full_cov, white=self.whiten)
        return mu + self.mean_function(Xnew), var
