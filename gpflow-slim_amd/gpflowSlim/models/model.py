"""Model / GPModel shells.

Mirrors gpflowSlim/models/model.py:29-166.  ``objective`` / ``likelihood_tensor`` are floats
(eager), not graph tensors.  ``optimize()`` (eager L-BFGS through TF autodiff, :172-196) is not
mirrored; the gradients it would consume come from ``GPR.compute_log_likelihood_and_gradients``.
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
