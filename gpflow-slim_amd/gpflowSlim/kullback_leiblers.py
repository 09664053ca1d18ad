"""KL[q || p] between Gaussians (the prior term of the SVGP bound), evaluated on the GPU.

API of gpflowSlim/kullback_leiblers.py:26-105; the arithmetic is ``gps_gauss_kl`` (csrc/gps_cond.hip): one Cholesky of K,
the Mahalanobis term and log|K| from one fused reduction over the factor, tr(K^-1 S_q) through row sums of squares of
Lp^-T (diagonal q_sqrt) or of (Lp^-1 L_q)^T (full q_sqrt) -- K^-1 is never formed and nothing but the inputs and one
scalar crosses PCIe.  ``models.SVGP`` does not come through here for its bound: ``gps_svgp_elbo`` shares the factor of
Kuu between the conditional and this KL.
"""
from . import _backend as be


def gauss_kl(q_mu, q_sqrt, K=None):
    """KL[N(q_mu, q_sqrt q_sqrt^T) || N(0, K)] summed over the columns of q_mu [M, L].

    q_sqrt: [M, L] (diagonal square roots) or [M, M, L] (lower-triangular square roots, upper parts ignored);
    K: [M, M] positive definite, or None for p = N(0, I)."""
    return be.get_handle().gauss_kl(q_mu, q_sqrt, K)
