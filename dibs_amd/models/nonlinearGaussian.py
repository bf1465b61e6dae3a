"""Nonlinear (per-node MLP) Gaussian likelihood.  Same name / kwargs as
dibs/models/nonlinearGaussian.py:84-326.  Parameters are a nested list
``[(W1[d,d,H1], b1[d,H1]), (), ..., (WL[d,H,1], bL[d,1])]`` like the reference's stax pytree."""
import numpy as np

from .. import random
from ..graph_utils import topological_order

_ACT = {"relu": lambda v: np.maximum(v, 0), "tanh": np.tanh, "sigmoid": lambda v: 1 / (1 + np.exp(-v)),
        "leakyrelu": lambda v: np.where(v > 0, v, 0.01 * v)}


class DenseNonlinearGaussian:
    _dibs_likelihood = "densenn"

    def __init__(self, *, n_vars, hidden_layers, obs_noise=0.1, sig_param=1.0, activation="relu", bias=True):
        if activation not in _ACT:
            raise KeyError(f"Invalid activation function `{activation}`")
        self.n_vars = n_vars
        self.obs_noise = obs_noise
        self.sig_param = sig_param
        self.hidden_layers = tuple(hidden_layers)
        self.activation = activation
        self.bias = bias
        self.no_interv_targets = np.zeros(self.n_vars, bool)

    def _config_kwargs(self):
        return dict(likelihood="densenn", nn_hidden=self.hidden_layers, nn_activation=self.activation,
                    nn_bias=self.bias, nn_obs_noise=self.obs_noise, nn_sig_param=self.sig_param)

    def _sizes(self, n_vars):
        return [n_vars] + list(self.hidden_layers) + [1]

    def get_theta_shape(self, *, n_vars):
        s = self._sizes(n_vars)
        out = []
        for li in range(len(s) - 1):
            out.append(((n_vars, s[li], s[li + 1]), (n_vars, s[li + 1])) if self.bias else ((n_vars, s[li], s[li + 1]),))
            if li < len(s) - 2:
                out.append(())
        return out

    def flat_to_tree(self, flat, n_vars):
        """[..., P] engine rows -> nested stax-like structure with the same leading dims."""
        flat = np.asarray(flat)
        lead = flat.shape[:-1]
        s = self._sizes(n_vars)
        out, off = [], 0
        for li in range(len(s) - 1):
            nw = n_vars * s[li] * s[li + 1]
            w = flat[..., off:off + nw].reshape(lead + (n_vars, s[li], s[li + 1]))
            off += nw
            if self.bias:
                nb = n_vars * s[li + 1]
                b = flat[..., off:off + nb].reshape(lead + (n_vars, s[li + 1]))
                off += nb
                out.append((w, b))
            else:
                out.append((w,))
            if li < len(s) - 2:
                out.append(())
        return out

    def tree_to_flat(self, theta):
        from ..utils.tree import tree_leaves
        leaves = tree_leaves(theta)
        lead = leaves[0].shape[:-3]
        return np.concatenate([np.asarray(l, np.float32).reshape(lead + (-1,)) for l in leaves], axis=-1)

    def sample_parameters(self, *, key, n_vars, n_particles=0, batch_size=0):
        """One key per (batch, particle, node); stax.serial / Dense key discipline (reference :155-186)."""
        shape = [s for s in (batch_size, n_particles, n_vars) if s != 0]
        n = int(np.prod(shape))
        subkeys = random.split(key, n)
        s = self._sizes(n_vars)
        nl = len(s) - 1
        Ws = [np.zeros((n, s[l], s[l + 1]), np.float32) for l in range(nl)]
        Bs = [np.zeros((n, s[l + 1]), np.float32) for l in range(nl)]
        for q in range(n):
            rng = subkeys[q]
            li = 0
            for si in range(2 * nl - 1):
                rng, layer_rng = random.split(rng)
                if si & 1:
                    continue
                if self.bias:
                    k1, k2 = random.split(layer_rng)
                    Ws[li][q] = random.normal(k1, (s[li], s[li + 1])) * np.float32(self.sig_param)
                    Bs[li][q] = random.normal(k2, (s[li + 1],)) * np.float32(self.sig_param)
                else:
                    Ws[li][q] = random.normal(layer_rng, (s[li], s[li + 1])) * np.float32(self.sig_param)
                li += 1
        out = []
        for l in range(nl):
            w = Ws[l].reshape(tuple(shape) + (s[l], s[l + 1]))
            out.append((w, Bs[l].reshape(tuple(shape) + (s[l + 1],))) if self.bias else (w,))
            if l < nl - 1:
                out.append(())
        return out

    def _forward_node(self, theta, j, xin):
        act = _ACT[self.activation]
        h = xin
        layers = [t for t in theta if len(t)]
        for li, lay in enumerate(layers):
            h = h @ np.asarray(lay[0])[j]
            if self.bias:
                h = h + np.asarray(lay[1])[j]
            if li < len(layers) - 1:
                h = act(h)
        return h[:, 0]

    def sample_obs(self, *, key, n_samples, g, theta, toporder=None, interv=None):
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
            if g[:, j].sum() > 0:
                x[:, j] = self._forward_node(theta, j, x * g[:, j][None]) + noise[:, j]
            else:
                x[:, j] = noise[:, j]
        return x

    # host-side (numpy, float64) evaluation helpers with the reference's names (nonlinearGaussian.py:248-318)
    def log_prob_parameters(self, *, theta, g):
        g = np.asarray(g, np.float64)
        sp = self.sig_param
        logn = lambda a: -0.5 * (np.asarray(a, np.float64) / sp) ** 2 - np.log(sp) - 0.5 * np.log(2.0 * np.pi)
        layers = [t for t in theta if len(t)]
        lp = 0.0
        for li, lay in enumerate(layers):
            for leaf_i, leaf in enumerate(lay):
                lw = logn(leaf)
                if li == 0 and leaf_i == 0:  # first-layer weights [d, d, H]: masked by g^T (nonlinearGaussian.py:264-269)
                    lw = lw * g.T[:, :, None]
                lp += float(lw.sum())
        return lp

    def log_likelihood(self, *, x, theta, g, interv_targets):
        x, it, g = np.asarray(x, np.float64), np.asarray(interv_targets), np.asarray(g, np.float64)
        assert x.shape == it.shape
        means = np.stack([self._forward_node(theta, j, x * g[:, j][None]) for j in range(g.shape[0])], axis=1)
        ll = -0.5 * (x - means) ** 2 / self.obs_noise - 0.5 * np.log(2.0 * np.pi * self.obs_noise)
        return float(np.sum(np.where(it != 0, 0.0, ll)))

    def interventional_log_joint_prob(self, g, theta, x, interv_targets, rng=None):
        from ..inference.scoring import score_graphs
        flat = self.tree_to_flat(theta)
        return float(score_graphs(self, np.asarray(g)[None], flat[None], x, interv_targets)[0])
