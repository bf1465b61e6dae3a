"""Graph priors p(G).  Same class names / constructor arguments as dibs/models/graph.py of the reference;
graphs are adjacency matrices (no igraph).  The SVGD engine needs only the *kind* and hyper-parameters of
the prior (its soft log-prob gradient is evaluated inside k_zgrad); the methods below serve data
generation and host-side inspection."""
import numpy as np

from .. import random
from ..graph_utils import mat_is_dag


class ErdosReniDAGDistribution:
    """Randomly oriented Erdos-Renyi DAGs, p(G) ~ p^e (1-p)^(C(d,2)-e)   (reference: graph.py:10-108)."""
    _dibs_prior = "er"

    def __init__(self, n_vars, n_edges_per_node=2):
        self.n_vars = n_vars
        self.n_edges_per_node = n_edges_per_node
        self.n_edges = n_edges_per_node * n_vars
        self.p = self.n_edges / ((self.n_vars * (self.n_vars - 1)) / 2)

    def sample_G(self, key, return_mat=False):
        key, subk = random.split(key)
        mat = random.bernoulli(subk, self.p, (self.n_vars, self.n_vars)).astype(np.int32)
        dag = np.tril(mat, k=-1)
        key, subk = random.split(key)
        perm = random.permutation(subk, self.n_vars)
        P = np.eye(self.n_vars, dtype=np.int32)[perm]
        return P.T @ dag @ P

    def unnormalized_log_prob_single(self, *, g, j):
        n_parents = int(np.asarray(g)[:, j].sum())
        return n_parents * np.log(self.p) + (self.n_vars - n_parents - 1) * np.log(1 - self.p)

    def unnormalized_log_prob(self, *, g):
        n_pairs = self.n_vars * (self.n_vars - 1) / 2.0
        e = float(np.asarray(g).sum())
        return e * np.log(self.p) + (n_pairs - e) * np.log(1 - self.p)

    def unnormalized_log_prob_soft(self, *, soft_g):
        return self.unnormalized_log_prob(g=soft_g)


class ScaleFreeDAGDistribution:
    """Randomly oriented scale-free DAGs, p(G) ~ prod_j (1 + indeg(j))^-3   (reference: graph.py:111-196)."""
    _dibs_prior = "sf"

    def __init__(self, n_vars, verbose=False, n_edges_per_node=2):
        self.n_vars = n_vars
        self.n_edges_per_node = n_edges_per_node
        self.verbose = verbose

    def sample_G(self, key, return_mat=False):
        """Barabasi-Albert preferential attachment (m = n_edges_per_node), edges new -> old, then a random
        relabelling.  (The reference delegates to igraph.Graph.Barabasi; this is an own generator.)"""
        d, m = self.n_vars, self.n_edges_per_node
        # one subkey for the attachment draws, one for the relabelling (both Threefry streams of dibs_amd.random)
        k_attach, k_perm = random.split(random.as_key(key))
        mat = np.zeros((d, d), np.int32)
        deg = np.zeros(d)
        for v in range(1, d):
            k = min(m, v)
            w = deg[:v] + 1.0
            # k targets without replacement, probability ~ degree + 1: Gumbel top-k on the uniform stream of this vertex
            k_attach, sub = random.split(k_attach)
            u01 = np.asarray(random.uniform(sub, (v,)), np.float64)
            gumbel = -np.log(-np.log(np.clip(u01, 1e-12, 1.0 - 1e-12)))
            targets = np.argsort(-(np.log(w / w.sum()) + gumbel), kind="stable")[:k]
            for u in targets:
                mat[v, u] = 1
                deg[u] += 1
                deg[v] += 1
        perm = random.permutation(k_perm, d)
        P = np.eye(d, dtype=np.int32)[perm]
        return P.T @ mat @ P

    def unnormalized_log_prob_single(self, *, g, j):
        return -3 * np.log(1 + int(np.asarray(g)[:, j].sum()))

    def unnormalized_log_prob(self, *, g):
        return float(sum(self.unnormalized_log_prob_single(g=g, j=j) for j in range(self.n_vars)))

    def unnormalized_log_prob_soft(self, *, soft_g):
        return float(np.sum(-3 * np.log(1 + np.asarray(soft_g).sum(0))))


class UniformDAGDistributionRejection:
    """Uniform over DAGs by rejection (d <= 5 in practice)   (reference: graph.py:199-276)."""
    _dibs_prior = "uniform"

    def __init__(self, n_vars):
        self.n_vars = n_vars

    def sample_G(self, key, return_mat=False):
        while True:
            key, subk = random.split(key)
            mat = random.bernoulli(subk, 0.5, (self.n_vars, self.n_vars)).astype(np.int32)
            np.fill_diagonal(mat, 0)
            if mat_is_dag(mat):
                return mat

    def unnormalized_log_prob_single(self, *, g, j):
        return 0.0

    def unnormalized_log_prob(self, *, g):
        return 0.0

    def unnormalized_log_prob_soft(self, *, soft_g):
        return 0.0
