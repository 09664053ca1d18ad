"""KL[q || p] between Gaussians for the SVGP objective.

Mirrors gpflowSlim/kullback_leiblers.py:26-105.  The Cholesky of the prior covariance and the
triangular solves run on the GPU (gps_potrf / gps_trsm_lower); the scalar reductions stay on the host.
"""
import numpy as np

from . import _backend as be
from ._settings import settings


def gauss_kl(q_mu, q_sqrt, K=None):
    """kullback_leiblers.py:26-105.  q_mu [M, N]; q_sqrt [M, N] (diagonal) or [M, M, N] (lower
    triangular square roots); K [M, M] or None (p = N(0, I))."""
    q_mu = np.asarray(q_mu, dtype=settings.float_type)
    q_sqrt = np.asarray(q_sqrt, dtype=settings.float_type)
    h = be.get_handle()
    if K is None:
        white = True
        alpha = q_mu
    else:
        white = False
        Lp = h.potrf(K)
        alpha = h.trsm_lower(Lp, q_mu, trans=False)

    if q_sqrt.ndim == 2:
        diag = True
        num_latent = q_sqrt.shape[1]
        NM = q_sqrt.size
        Lq = Lq_diag = q_sqrt
    elif q_sqrt.ndim == 3:
        diag = False
        num_latent = q_sqrt.shape[2]
        NM = q_sqrt.shape[1] * q_sqrt.shape[2]
        Lq = np.tril(np.transpose(q_sqrt, (2, 0, 1)))                 # force lower triangle
        Lq_diag = np.diagonal(Lq, axis1=1, axis2=2)
    else:
        raise ValueError("Bad dimension for q_sqrt: {}".format(q_sqrt.ndim))

    mahalanobis = np.sum(np.square(alpha))
    constant = -float(NM)
    logdet_qcov = np.sum(np.log(np.square(Lq_diag)))

    if white:
        trace = np.sum(np.square(Lq))
    else:
        if diag:
            M = Lp.shape[0]
            Lp_inv = h.trsm_lower(Lp, np.eye(M, dtype=settings.float_type), trans=False)
            K_inv = h.trsm_lower(Lp, Lp_inv, trans=True)
            trace = np.sum(np.diag(K_inv)[:, None] * np.square(q_sqrt))
        else:
            trace = 0.0
            for i in range(num_latent):
                LpiLq = h.trsm_lower(Lp, Lq[i], trans=False)
                trace += np.sum(np.square(LpiLq))

    twoKL = mahalanobis + constant - logdet_qcov + trace
    if not white:
        sum_log_sqdiag_Lp = np.sum(np.log(np.square(np.diag(Lp))))
        twoKL += num_latent * sum_log_sqdiag_Lp
    return 0.5 * twoKL
