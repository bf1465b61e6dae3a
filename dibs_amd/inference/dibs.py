"""Host-side base class with the small post-processing helpers of the reference's ``DiBS`` base
(dibs/inference/dibs.py:84-99 particle_to_g_lim, :168-184 edge_probs, :661-692 callback).  The gradient
estimators of that class live in the HIP kernels (dibs_amd/csrc), not here."""
import numpy as np

from ..graph_utils import elwise_acyclic_constr_nograd


class DiBS:
    def __init__(self, *, x, interv_mask, alpha_linear, beta_linear, tau, n_grad_mc_samples, n_acyclicity_mc_samples,
                 grad_estimator_z, score_function_baseline, latent_prior_std, verbose):
        self.x = np.ascontiguousarray(x, np.float32)
        self.interv_mask = interv_mask
        self.n_vars = self.x.shape[-1]
        self.alpha_linear = alpha_linear
        self.beta_linear = beta_linear
        self.alpha = lambda t: (alpha_linear * t)
        self.beta = lambda t: (beta_linear * t)
        self.tau = tau
        self.n_grad_mc_samples = n_grad_mc_samples
        self.n_acyclicity_mc_samples = n_acyclicity_mc_samples
        self.grad_estimator_z = grad_estimator_z
        self.score_function_baseline = score_function_baseline
        self.latent_prior_std = latent_prior_std
        self.verbose = verbose

    @staticmethod
    def _scores(z):
        z = np.asarray(z, np.float32)
        return np.einsum("...ik,...jk->...ij", z[..., 0], z[..., 1])

    @staticmethod
    def _zero_diag(a):
        a = np.array(a)
        d = a.shape[-1]
        a[..., np.arange(d), np.arange(d)] = 0
        return a

    def particle_to_g_lim(self, z):
        """G for alpha -> infinity: (U V^T > 0), zero diagonal; int32 [..., d, d]."""
        return self._zero_diag((self._scores(z) > 0).astype(np.int32))

    def edge_probs(self, z, t):
        s = self._scores(z).astype(np.float64)
        return self._zero_diag((1.0 / (1.0 + np.exp(-self.alpha(t) * s))).astype(np.float32))

    def visualize_callback(self, ipython=False, save_path=None):
        """Text-only progress callback (the reference's plots need matplotlib/IPython: out of scope)."""
        def callback(**kwargs):
            zs, t = kwargs["zs"], kwargs["t"]
            gs = self.particle_to_g_lim(zs)
            n_cyc = int((elwise_acyclic_constr_nograd(gs, self.n_vars) > 0).sum())
            print(f"iteration {t:6d} | alpha {self.alpha(t):6.1f} | beta {self.beta(t):6.1f} | #cyclic {n_cyc:3d}")
        return callback
