"""Exact Gaussian-process regression on one MI355X.

Mirrors gpflowSlim/models/gpr.py:28-132 (exact branch :69-72, :119-131).  X is uploaded once
and stays in HBM; every ``likelihood_tensor`` / ``objective`` read rebuilds K, re-factors and
re-solves on the device, as the reference's ``sess.run(objective)`` does; ``predict_f``
re-factors too (models/gpr.py:119-121) unless ``reuse_factor`` is switched on.
"""
import numpy as np

from .. import likelihoods
from .. import _backend as be
from .._settings import settings
from .model import GPModel


class _FactorKey(object):
    """What the resident factor was computed from: the hyper-parameter bytes and the X / Y array OBJECTS (held, so
    that their addresses cannot be recycled while the key is alive)."""

    def __init__(self, params, X, Y):
        self.params, self.X, self.Y = params, X, Y

    def __eq__(self, other):
        return isinstance(other, _FactorKey) and self.params == other.params and self.X is other.X and self.Y is other.Y

    def __ne__(self, other):
        return not self.__eq__(other)


class GPR(GPModel):
    def __init__(self, X, Y, kern, mean_function=None, obs_var=0.1, num_latent=None, min_var=None, **kwargs):
        """X [N, D], Y [N, R]; kern, mean_function as in the reference (models/gpr.py:41-53)."""
        X = np.ascontiguousarray(X, dtype=settings.float_type)
        Y = np.ascontiguousarray(Y, dtype=settings.float_type)
        if X.ndim != 2 or Y.ndim != 2 or X.shape[0] != Y.shape[0]:
            raise ValueError("GPR needs X [N, D] and Y [N, R]")
        likelihood = likelihoods.Gaussian(var=obs_var, min_var=min_var)
        GPModel.__init__(self, X, Y, kern, likelihood, mean_function, **kwargs)
        self.num_latent = Y.shape[1] if num_latent is None else num_latent
        # not in the reference: opt-in reuse of the resident factor by predict_f
        self.reuse_factor = False

    # ---- device plumbing -------------------------------------------------------------------
    @property
    def _factor_key(self):
        """What the factor resident on the device was computed from (kept on the HANDLE: several models may share
        one handle, and even one X array; the factor belongs to whoever evaluated last)."""
        return be.get_handle().factor_key

    @_factor_key.setter
    def _factor_key(self, key):
        be.get_handle().factor_key = key

    def _handle(self):
        h = be.get_handle()
        # The handle keeps a reference to the array it uploaded, so "is" cannot be fooled by a new array that the
        # allocator placed at a freed model's address (ids and data pointers are reused; object identity of a live
        # object is not).  X is treated as immutable, like the reference's tensor; assign a new array to change it.
        if h.resident_token is not self.X:
            h.gpr_set_data(self.X, self.X)
            self._factor_key = None
        return h

    def _state_key(self):
        parts = [p.vf_val.tobytes() for p in self.parameters]
        return _FactorKey(b"|".join(parts), self.X, self.Y)

    def _resid(self):
        return np.ascontiguousarray(self.Y - self.mean_function(self.X))

    # ---- reference API ---------------------------------------------------------------------
    def _build_likelihood(self):
        """models/gpr.py:69-72 + densities.py:73-95, fused on the device."""
        h = self._handle()
        prog = self.kern._program(self.X.shape[1])
        self._factor_key = None
        lml = h.gpr_lml(prog, float(np.squeeze(self.likelihood.variance)), self._resid())
        self._factor_key = self._state_key()
        return lml

    def compute_log_likelihood_and_gradients(self):
        """LML and d LML / d(unconstrained parameter) for every parameter of the model -- what
        `tf.gradients(objective, variables)` yields in the reference (examples/gpr.py:53-54) up to the
        sign of `objective = -LML`.  Returns (lml, [(Parameter, gradient array shaped like
        Parameter.unconstrained_tensor), ...]) in `self.parameters` order.  Priors are not included."""
        h = self._handle()
        d_all = self.X.shape[1]
        prog = self.kern._program(d_all)
        layout = self.kern._grad_layout(d_all)
        resid = self._resid()
        self._factor_key = None
        lml, slots, gnoise, kinv_resid = h.gpr_lml_grad(prog, float(np.squeeze(self.likelihood.variance)), resid)
        self._factor_key = self._state_key()
        if len(layout) != len(slots):
            raise RuntimeError("gradient slot layout mismatch: %d vs %d" % (len(layout), len(slots)))
        grads = {id(p): np.zeros_like(np.atleast_1d(p.vf_val), dtype=settings.float_type) for p in self.parameters}
        for (param, idx), g in zip(layout, slots):
            if param is None:
                continue
            if idx is None:
                grads[id(param)] += g
            else:
                grads[id(param)].reshape(-1)[idx] += g          # index into the flattened parameter
        grads[id(self.likelihood._variance)] += gnoise
        # mean function: d LML / d m(X) = K_y^-1 (Y - m) ; Zero has no parameters
        from ..mean_functions import Constant as _MConst, Linear as _MLin
        mf = self.mean_function

        def _fit(g, like):           # sum a per-output gradient into a parameter that broadcasts over outputs
            like = np.atleast_1d(like)
            return g.reshape(like.shape) if g.size == like.size else np.full(like.shape, np.sum(g))

        if isinstance(mf, _MConst):
            grads[id(mf.c)] = grads[id(mf.c)] + _fit(np.sum(kinv_resid, axis=0), mf.c.vf_val)
        elif isinstance(mf, _MLin):
            grads[id(mf.A)] = grads[id(mf.A)] + _fit(self.X.T @ kinv_resid, mf.A.vf_val)
            grads[id(mf.b)] = grads[id(mf.b)] + _fit(np.sum(kinv_resid, axis=0), mf.b.vf_val)
        out = []
        for p in self.parameters:
            g = grads[id(p)].reshape(np.atleast_1d(p.vf_val).shape) * np.atleast_1d(p.transform.forward_grad(p.vf_val))
            out.append((p, g.reshape(p.vf_val.shape)))
        return lml, out

    def _build_predict(self, Xnew, full_cov=False):
        """models/gpr.py:119-131"""
        Xnew = np.ascontiguousarray(Xnew, dtype=settings.float_type)
        h = self._handle()
        prog = self.kern._program(self.X.shape[1])
        key = self._state_key()
        warm = bool(self.reuse_factor) and self._factor_key is not None and self._factor_key == key
        self._factor_key = None
        mean, var = h.gpr_predict(prog, float(np.squeeze(self.likelihood.variance)), self._resid(), Xnew,
                                  full_cov=full_cov, refactor=not warm)
        self._factor_key = key
        fmean = mean + self.mean_function(Xnew)
        R = self.Y.shape[1]
        if full_cov:
            fvar = np.tile(var[:, :, None], [1, 1, R])             # models/gpr.py:127-128
        else:
            fvar = np.tile(np.reshape(var, (-1, 1)), [1, R])       # models/gpr.py:131
        return fmean, fvar
