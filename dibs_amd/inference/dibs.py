"""Base class with the public methods of the reference's ``DiBS`` base (dibs/inference/dibs.py:84-321, 557-658).

Two kinds of methods:
  * small per-particle helpers of the latent graph model p(G | Z) (graph samples from given noise, edge log-probabilities, the
    latent log-probability and its gradient in closed form, the soft graph prior, the acyclicity value): host numpy, O(d^2 k) each --
    the reference evaluates them eagerly on whatever arrays the caller passes;
  * everything that evaluates a likelihood or a Monte-Carlo gradient estimator (``eltwise_log_joint_prob``,
    ``eltwise_grad_z_likelihood``, ``eltwise_grad_theta_likelihood``, ``eltwise_grad_latent_prior``): the HIP engine through the C ABI
    (dibs_engine_eval_gradients / dibs_score_graphs), with the PRNG keys the caller passes, exactly as one SVGD step uses them.
There is no CPU fallback for the second kind."""
import numpy as np

from .. import random
from ..graph_utils import acyclic_constr_nograd, elwise_acyclic_constr_nograd


def _log_sigmoid(x):
    x = np.asarray(x, np.float32)
    return (np.minimum(x, np.float32(0)) - np.log1p(np.exp(-np.abs(x)))).astype(np.float32)


def _sigmoid(x):
    x = np.asarray(x, np.float64)
    return (1.0 / (1.0 + np.exp(-x))).astype(np.float32)


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

    # ---- generative graph model p(G | Z): host helpers (dibs.py:102-247) -------------------------
    def sample_g(self, p, subk, n_samples):
        """Bernoulli(p) graphs ``[n_samples, d, d]`` (int32, zero diagonal) from the key ``subk`` (dibs.py:102-119)."""
        p = np.asarray(p, np.float32)
        d = p.shape[-1]
        return self._zero_diag(random.bernoulli(subk, p, (n_samples, d, d)).astype(np.int32))

    def particle_to_soft_graph(self, z, eps, t):
        """Gumbel-softmax sample sigmoid(tau (eps + alpha(t) U V^T)), zero diagonal (dibs.py:121-140); eps ~ Logistic(0, 1)."""
        s = self._scores(z)
        return self._zero_diag(_sigmoid(np.float32(self.tau) * (np.asarray(eps, np.float32) + np.float32(self.alpha(t)) * s)))

    def particle_to_hard_graph(self, z, eps, t):
        """Gumbel-max sample ((eps + alpha(t) U V^T) > 0) as float32, zero diagonal (dibs.py:143-166)."""
        s = self._scores(z)
        return self._zero_diag(((np.asarray(eps, np.float32) + np.float32(self.alpha(t)) * s) > 0).astype(np.float32))

    def edge_log_probs(self, z, t):
        """(log sigmoid(alpha s), log sigmoid(-alpha s)), both with zero diagonal (dibs.py:187-204)."""
        a = np.float32(self.alpha(t)) * self._scores(z)
        return self._zero_diag(_log_sigmoid(a)), self._zero_diag(_log_sigmoid(-a))

    def latent_log_prob(self, single_g, single_z, t):
        """log p(G | Z) = sum_{i != j} g log p + (1 - g) log(1 - p) (dibs.py:208-229)."""
        log_p, log_1_p = self.edge_log_probs(single_z, t)
        g = np.asarray(single_g, np.float32)
        return np.float32(np.sum(g * log_p + (np.float32(1) - g) * log_1_p))

    def eltwise_grad_latent_log_prob(self, gs, single_z, t):
        """grad_Z log p(G | Z) for a batch of graphs ``[n, d, d]`` and one Z -> ``[n, d, k, 2]`` (dibs.py:232-247).  Closed form of the
        reference's autodiff: d/ds_ij = alpha (g_ij - p_ij) off the diagonal, then dU = dS V, dV = dS^T U."""
        z = np.asarray(single_z, np.float32)
        u, v = z[..., 0], z[..., 1]
        a = np.float32(self.alpha(t))
        ds = self._zero_diag(a * (np.asarray(gs, np.float32) - self.edge_probs(z, t)[None]))
        return np.stack([ds @ v, np.swapaxes(ds, -1, -2) @ u], axis=-1).astype(np.float32)

    # ---- latent prior p(Z): host helpers (dibs.py:557-623) ----------------------------------------
    def constraint_gumbel(self, single_z, single_eps, t):
        """h(G~(Z, eps)) = tr((I + G~/d)^d) - d for one Gumbel-soft graph (dibs.py:557-573)."""
        n_vars = np.asarray(single_z).shape[0]
        return acyclic_constr_nograd(self.particle_to_soft_graph(single_z, single_eps, t), n_vars)

    def log_graph_prior_particle(self, single_z, t):
        """log p(G_alpha(Z)): the graph prior on the matrix of edge probabilities (dibs.py:604-623)."""
        return self.log_graph_prior(soft_g=self.edge_probs(single_z, t))

    def log_graph_prior(self, *, soft_g):
        return self.graph_model.unnormalized_log_prob_soft(soft_g=soft_g)

    def visualize_callback(self, ipython=False, save_path=None):
        """Text-only progress callback (the reference's plots need matplotlib/IPython: out of scope)."""
        def callback(**kwargs):
            zs, t = kwargs["zs"], kwargs["t"]
            gs = self.particle_to_g_lim(zs)
            n_cyc = int((elwise_acyclic_constr_nograd(gs, self.n_vars) > 0).sum())
            print(f"iteration {t:6d} | alpha {self.alpha(t):6.1f} | beta {self.beta(t):6.1f} | #cyclic {n_cyc:3d}")
        return callback
