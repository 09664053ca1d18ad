"""Model / GPModel shells.

Mirrors gpflowSlim/models/model.py:29-166.  ``objective`` / ``likelihood_tensor`` are floats
(eager), not graph tensors.  ``optimize()`` (:172-196: eager L-BFGS over TF autodiff gradients, with an
Adam fall-back) drives the same loop with the analytic gradients of
``compute_log_likelihood_and_gradients`` (models that have them: GPR) -- L-BFGS-B from scipy, or Adam.
"""
import numpy as np

from ..mean_functions import Zero
from .._settings import settings


class Model(object):
    def __init__(self, name='model'):
        self._name = name
        self._parameters = []

    @property
    def name(self):
        return self._name

    def compute_log_prior(self):
        return self.prior_tensor

    def compute_log_likelihood(self):
        return self.likelihood_tensor

    @property
    def parameters(self):
        return self._parameters

    @property
    def likelihood_tensor(self):
        return self._build_likelihood()

    @property
    def prior_tensor(self):
        """models/model.py:57-65"""
        priors = [p._build_prior(p.unconstrained_tensor, p.constrained_tensor) for p in self.parameters]
        if not priors:
            return 0.0
        return float(np.sum(priors))

    @property
    def objective(self):
        """models/model.py:67-73"""
        return -(self.likelihood_tensor + self.prior_tensor)

    def _build_likelihood(self):
        raise NotImplementedError

    # ---- fitting (models/model.py:172-196, LBFGS.py) ----------------------------------------------------------
    def _pack(self):
        return np.concatenate([np.atleast_1d(p.vf_val).ravel() for p in self.parameters]) if self.parameters else np.zeros(0)

    def _unpack(self, x):
        k = 0
        for p in self.parameters:
            shape = np.shape(p.vf_val)
            size = int(np.prod(shape)) if shape else 1
            p.assign_unconstrained(np.asarray(x[k:k + size], dtype=settings.float_type).reshape(shape))
            k += size

    def _objective_and_grad(self, x):
        """objective = -(LML + priors) and its gradient w.r.t. the packed unconstrained parameters."""
        if not hasattr(self, "compute_log_likelihood_and_gradients"):
            raise NotImplementedError("%s has no analytic gradients" % type(self).__name__)
        self._unpack(x)
        lml, grads = self.compute_log_likelihood_and_gradients()
        g = np.concatenate([np.atleast_1d(gi).ravel() if getattr(p, "trainable", True) else np.zeros(np.size(gi))
                            for p, gi in grads])
        lp = 0.0
        if any(p.prior is not None for p in self.parameters):
            # log-priors (params.py:176-194: log p(constrained) + log |d constrained / d unconstrained|) are host functions of
            # one parameter each, whatever object the caller supplied as `prior`: central differences over that parameter's own
            # unconstrained entries
            gp, lp = self._prior_and_grad([p for p, _ in grads])
            g = g + gp
        return -float(lml + lp), -g

    def _prior_and_grad(self, params):
        from ..params import Parameter
        total, pieces = 0.0, []
        for p in params:
            u0 = np.array(p.vf_val, dtype=np.float64)
            gpiece = np.zeros(u0.size)
            if p.prior is not None:
                def lp_at(u, p=p):
                    return p._build_prior(u, p.transform.forward(u))
                total += lp_at(u0)
                if getattr(p, "trainable", True):
                    flat = u0.ravel()
                    for i in range(flat.size):
                        hh = 1e-6 * max(1.0, abs(flat[i]))
                        up, um = flat.copy(), flat.copy()
                        up[i] += hh; um[i] -= hh
                        gpiece[i] = (lp_at(up.reshape(u0.shape)) - lp_at(um.reshape(u0.shape))) / (2 * hh)
            pieces.append(gpiece)
        return np.concatenate(pieces) if pieces else np.zeros(0), total

    def optimize(self, max_iter=1000, method="L-BFGS-B", learning_rate=1e-2, callback=None, tol=None):
        """Minimise ``objective`` over the unconstrained parameters.  method: 'L-BFGS-B' (scipy, 20 corrections
        like the reference's LBFGS(nCorrection=20)) or 'adam'.  A non-positive-definite step (Cholesky failure)
        is treated as +inf by the line search.  Returns the final objective."""
        from .._backend import NotPositiveDefiniteError
        x0 = self._pack()

        def fg(x):
            try:
                f, g = self._objective_and_grad(x)
            except NotPositiveDefiniteError:
                return 1e300, np.zeros_like(x)
            if not np.isfinite(f) or not np.all(np.isfinite(g)):
                return 1e300, np.zeros_like(x)
            return f, g

        if method.lower() == "adam":
            x = x0.copy()
            m1 = np.zeros_like(x); m2 = np.zeros_like(x)
            b1, b2, eps = 0.9, 0.999, 1e-8
            best_f, best_x = np.inf, x.copy()
            for it in range(1, max_iter + 1):
                f, g = fg(x)
                if f < best_f:
                    best_f, best_x = f, x.copy()
                m1 = b1 * m1 + (1 - b1) * g
                m2 = b2 * m2 + (1 - b2) * g * g
                x = x - learning_rate * (m1 / (1 - b1 ** it)) / (np.sqrt(m2 / (1 - b2 ** it)) + eps)
                if callback is not None:
                    callback(it, f)
            f, _ = fg(x)
            if f > best_f:
                x, f = best_x, best_f
            self._unpack(x)
            return f
        from scipy.optimize import minimize
        it = [0]

        def cb(xk):
            it[0] += 1
            if callback is not None:
                callback(it[0], None)

        res = minimize(fg, x0, jac=True, method="L-BFGS-B", callback=cb,
                       options={"maxiter": max_iter, "maxcor": 20, **({"ftol": tol} if tol is not None else {})})
        self._unpack(res.x)
        return float(res.fun)


class GPModel(Model):
    def __init__(self, X, Y, kern, likelihood, mean_function, name='GPModel'):
        super(GPModel, self).__init__(name=name)
        self.mean_function = mean_function or Zero()
        self.kern = kern
        self.likelihood = likelihood
        self.X, self.Y = X, Y
        self._parameters = self.mean_function.parameters + self.kern.parameters + self.likelihood.parameters

    def predict_f(self, Xnew):
        """models/model.py:121-126"""
        return self._build_predict(Xnew)

    def predict_f_full_cov(self, Xnew):
        """models/model.py:128-133"""
        return self._build_predict(Xnew, full_cov=True)

    def predict_f_samples(self, Xnew, num_samples):
        """models/model.py:135-148: samples from the posterior over f(Xnew); one Cholesky (on the GPU)
        of the jittered posterior covariance per latent function.  Returns [num_samples, N*, R]."""
        from .. import _backend as be
        mu, var = self._build_predict(Xnew, full_cov=True)
        jitter = np.eye(mu.shape[0], dtype=settings.float_type) * settings.numerics.jitter_level
        samples = []
        for i in range(self.num_latent):
            L = be.get_handle().potrf(var[:, :, i] + jitter)
            V = np.random.standard_normal((L.shape[0], num_samples)).astype(settings.float_type)
            samples.append(mu[:, i:i + 1] + np.matmul(L, V))
        return np.transpose(np.stack(samples))

    def predict_y(self, Xnew):
        """models/model.py:150-155"""
        pred_f_mean, pred_f_var = self._build_predict(Xnew)
        return self.likelihood.predict_mean_and_var(pred_f_mean, pred_f_var)

    def predict_density(self, Xnew, Ynew):
        """models/model.py:157-166"""
        pred_f_mean, pred_f_var = self._build_predict(Xnew)
        return self.likelihood.predict_density(pred_f_mean, pred_f_var, Ynew)

    def _build_predict(self, *args, **kwargs):
        raise NotImplementedError
