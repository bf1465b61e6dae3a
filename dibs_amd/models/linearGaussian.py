"""Linear-Gaussian likelihood models.  Same names / kwargs as dibs/models/linearGaussian.py; the SVGD engine
recognises the classes by ``_dibs_likelihood`` and lowers their hyper-parameters into ``dibs_config``.
Scoring of hard graphs goes through the device (``dibs_score_graphs``)."""
import numpy as np

from .. import random
from ..graph_utils import topological_order


class BGe:
    """BGe marginal likelihood (Geiger & Heckerman; Kuipers et al. 2014), diagonal T.
    Reference: linearGaussian.py:8-170."""
    _dibs_likelihood = "bge"

    def __init__(self, *, n_vars, mean_obs=None, alpha_mu=None, alpha_lambd=None):
        self.n_vars = n_vars
        self.mean_obs = np.zeros(n_vars, np.float32) if mean_obs is None else np.asarray(mean_obs, np.float32)
        self.alpha_mu = alpha_mu or 1.0
        self.alpha_lambd = alpha_lambd or (self.n_vars + 2)
        assert self.alpha_lambd > self.n_vars + 1
        self.no_interv_targets = np.zeros(self.n_vars, bool)

    def get_theta_shape(self, *, n_vars):
        raise NotImplementedError("Not available for BGe score; use `LinearGaussian` model instead.")

    def sample_parameters(self, *, key, n_vars, n_particles=0, batch_size=0):
        raise NotImplementedError("Not available for BGe score; use `LinearGaussian` model instead.")

    def sample_obs(self, *, key, n_samples, g, theta, toporder=None, interv=None):
        raise NotImplementedError("Not available for BGe score; use `LinearGaussian` model instead.")

    def _config_kwargs(self):
        return dict(likelihood="bge", bge_alpha_mu=self.alpha_mu, bge_alpha_lambd=self.alpha_lambd)

    def interventional_log_marginal_prob(self, g, _, x, interv_targets, rng=None):
        """log p(D | G) of one hard graph, evaluated on the device."""
        from ..inference.scoring import score_graphs
        return float(score_graphs(self, np.asarray(g)[None], None, x, interv_targets)[0])

    def log_marginal_likelihood(self, *, g, x, interv_targets):
        return self.interventional_log_marginal_prob(g, None, x, interv_targets)


class LinearGaussian:
    """Linear SEM with Gaussian edge weights and additive Gaussian noise.  Reference: linearGaussian.py:173-338."""
    _dibs_likelihood = "lingauss"

    def __init__(self, *, n_vars, obs_noise=0.1, mean_edge=0.0, sig_edge=1.0, min_edge=0.5):
        self.n_vars = n_vars
        self.obs_noise = obs_noise
        self.mean_edge = mean_edge
        self.sig_edge = sig_edge
        self.min_edge = min_edge
        self.no_interv_targets = np.zeros(self.n_vars, bool)

    def _config_kwargs(self):
        return dict(likelihood="lingauss", lin_obs_noise=self.obs_noise, lin_mean_edge=self.mean_edge,
                    lin_sig_edge=self.sig_edge, lin_min_edge=self.min_edge)

    def get_theta_shape(self, *, n_vars):
        return np.array((n_vars, n_vars))

    def sample_parameters(self, *, key, n_vars, n_particles=0, batch_size=0):
        shape = tuple(s for s in (batch_size, n_particles, n_vars, n_vars) if s != 0)
        theta = np.float32(self.mean_edge) + np.float32(self.sig_edge) * random.normal(key, shape)
        return (theta + np.sign(theta) * np.float32(self.min_edge)).astype(np.float32)

    def sample_obs(self, *, key, n_samples, g, theta, toporder=None, interv=None):
        """Ancestral sampling x_j = x_pa . theta[pa, j] + N(0, obs_noise); ``interv`` = {node: clamp value}."""
        interv = interv or {}
        g = np.asarray(g)
        d = g.shape[0]
        toporder = topological_order(g) if toporder is None else toporder
        key, subk = random.split(key)
        noise = np.float32(np.sqrt(self.obs_noise)) * random.normal(subk, (n_samples, d))
        x = np.zeros((n_samples, d), np.float32)
        for j in toporder:
            if j in interv:
                x[:, j] = interv[j]
                continue
            pa = np.where(g[:, j])[0]
            x[:, j] = (x[:, pa] @ np.asarray(theta)[pa, j] if pa.size else 0.0) + noise[:, j]
        return x

    # host-side (numpy, float64) evaluation helpers with the reference's names (linearGaussian.py:278-316); the SVGD path and
    # `interventional_log_joint_prob` run on the device
    def log_prob_parameters(self, *, theta, g):
        th, g = np.asarray(theta, np.float64), np.asarray(g, np.float64)
        z = (th - self.mean_edge) / self.sig_edge
        return float(np.sum(g * (-0.5 * z * z - np.log(self.sig_edge) - 0.5 * np.log(2.0 * np.pi))))

    def log_likelihood(self, *, x, theta, g, interv_targets):
        x, it = np.asarray(x, np.float64), np.asarray(interv_targets)
        assert x.shape == it.shape
        mean = x @ (np.asarray(g, np.float64) * np.asarray(theta, np.float64))
        ll = -0.5 * (x - mean) ** 2 / self.obs_noise - 0.5 * np.log(2.0 * np.pi * self.obs_noise)
        return float(np.sum(np.where(it != 0, 0.0, ll)))

    def interventional_log_joint_prob(self, g, theta, x, interv_targets, rng=None):
        from ..inference.scoring import score_graphs
        return float(score_graphs(self, np.asarray(g)[None], np.asarray(theta)[None], x, interv_targets)[0])
