"""Known-answer tests (CPU): the oracle's model formulas against INDEPENDENT ground truth -- closed forms evaluated with scipy, published
properties of the scores, hand-computed values.  The reference ships no golden vectors and cannot run here (no jax), so the
floating-point part of the oracle cannot be pinned against reference output; these tests pin each formula it restates against the
mathematics the reference implements instead.  A misreading of a reference formula shared by the oracle and the kernels (which the
oracle-vs-kernel parity tests cannot see) fails here.

  BGe:  Geiger & Heckerman (2002) / Kuipers, Moffa & Heckerman (2014, with the corrected R of the supplement), as implemented in
        dibs/models/linearGaussian.py:63-170
  LinearGaussian:  dibs/models/linearGaussian.py:278-338;   acyclicity:  dibs/graph_utils.py:8-28 (Yu et al. 2019)
  priors:  dibs/models/graph.py:93-108, 182-196;   kernels:  dibs/kernel.py:20-30, 52-71;   RMSprop:  jax.example_libraries.optimizers"""
import itertools
import math

import numpy as np
import pytest
import torch
from scipy import special, stats

from oracle import dibs_oracle as O


def _xt(a):
    return torch.as_tensor(np.asarray(a, np.float64))


def _data(n, d, seed):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(d, d))
    return rng.normal(size=(n, d)) @ a + rng.normal(size=d)   # correlated columns, non-zero means


def _bge(g, x, alpha_mu=1.0, alpha_lambd=None):
    hp = O.BGeParams(alpha_mu=alpha_mu, alpha_lambd=alpha_lambd)
    return float(O.bge_log_marginal(_xt(g), _xt(x), torch.zeros(x.shape, dtype=torch.float64), hp))


def _complete_dag(order):
    d = len(order)
    g = np.zeros((d, d))
    for a in range(d):
        for b in range(a + 1, d):
            g[order[a], order[b]] = 1.0
    return g


@pytest.mark.parametrize("d,n,alpha_mu,alpha_lambd", [(1, 7, 1.0, None), (2, 5, 1.0, None), (3, 20, 0.5, 7.0), (5, 40, 2.0, 9.5)])
def test_bge_complete_dag_is_the_normal_wishart_evidence(d, n, alpha_mu, alpha_lambd):
    """For a complete DAG (any variable order) the node scores telescope to the marginal likelihood of N observations under the
    normal-Wishart prior with nu = mean_obs = 0, T = t I  (Geiger & Heckerman 2002, eq. 18 with Kuipers' constants):
        log p(D) = -N d / 2 log pi + d / 2 log(a_mu / (a_mu + N)) + log Gamma_d((a_l + N) / 2) - log Gamma_d(a_l / 2)
                   + a_l / 2 log|T| - (a_l + N) / 2 log|R|,
        R = T + S_N + N a_mu / (N + a_mu) xbar xbar^T,   t = a_mu (a_l - d - 1) / (a_mu + 1)."""
    x = _data(n, d, seed=d)
    al = d + 2 if alpha_lambd is None else alpha_lambd
    t = alpha_mu * (al - d - 1) / (alpha_mu + 1)
    xbar = x.mean(0, keepdims=True)
    R = t * np.eye(d) + (x - xbar).T @ (x - xbar) + n * alpha_mu / (n + alpha_mu) * xbar.T @ xbar
    want = (-0.5 * n * d * math.log(math.pi) + 0.5 * d * math.log(alpha_mu / (alpha_mu + n))
            + special.multigammaln(0.5 * (al + n), d) - special.multigammaln(0.5 * al, d)
            + 0.5 * al * d * math.log(t) - 0.5 * (al + n) * np.linalg.slogdet(R)[1])
    for order in itertools.islice(itertools.permutations(range(d)), 6):
        got = _bge(_complete_dag(order), x, alpha_mu, alpha_lambd)
        assert abs(got - want) < 1e-9 * max(1.0, abs(want)), (order, got, want)


def test_bge_is_score_equivalent():
    """The "e" of BGe: Markov-equivalent DAGs have the same marginal likelihood.  Chain, reversed chain and fork over (0, 1, 2) are
    equivalent; the collider 0 -> 1 <- 2 is not.  Covered-edge reversal in a 4-node graph as a second case."""
    x = _data(30, 3, seed=1)
    e = lambda *edges: (lambda g: [g.__setitem__(ed, 1.0) for ed in edges] and g)(np.zeros((3, 3)))
    chain, rev, fork, coll = e((0, 1), (1, 2)), e((2, 1), (1, 0)), e((1, 0), (1, 2)), e((0, 1), (2, 1))
    s = [_bge(g, x) for g in (chain, rev, fork, coll)]
    assert abs(s[0] - s[1]) < 1e-9 and abs(s[0] - s[2]) < 1e-9
    assert abs(s[0] - s[3]) > 1e-3
    x4 = _data(25, 4, seed=2)
    g1 = np.zeros((4, 4)); g1[0, 1] = g1[0, 2] = g1[1, 2] = g1[2, 3] = 1     # 1 -> 2 is covered (pa(2) = pa(1) + {1})
    g2 = g1.copy(); g2[1, 2] = 0; g2[2, 1] = 1
    assert abs(_bge(g1, x4) - _bge(g2, x4)) < 1e-9


def test_bge_empty_graph_is_a_product_of_student_t_evidences():
    """Empty graph: every node is an independent one-dimensional normal-gamma model; its evidence in closed form
    (the d = 1 case of the formula above with the node's own sub-matrix of T and R, and a_l - d + 1 degrees of freedom)."""
    d, n = 4, 12
    x = _data(n, d, seed=4)
    al, am = d + 2.0, 1.0
    t = am * (al - d - 1) / (am + 1)
    want = 0.0
    for j in range(d):
        xb = x[:, j].mean()
        r = t + ((x[:, j] - xb) ** 2).sum() + n * am / (n + am) * xb * xb
        a0 = 0.5 * (al - d + 1)
        want += (-0.5 * n * math.log(math.pi) + 0.5 * math.log(am / (am + n)) + special.gammaln(a0 + 0.5 * n) - special.gammaln(a0)
                 + a0 * math.log(t) - (a0 + 0.5 * n) * math.log(r))
    assert abs(_bge(np.zeros((d, d)), x) - want) < 1e-9 * abs(want)


def test_linear_gaussian_log_joint_against_scipy():
    """log p(theta | G) + log p(D | G, theta) = sum_{ij} G_ij logN(theta_ij; mu, sig) + sum_{n, j not intervened} logN(x_nj; (x (G o theta))_nj, sqrt(s2))"""
    rng = np.random.default_rng(0)
    d, n = 5, 9
    g = np.triu((rng.random((d, d)) < 0.5).astype(float), 1)
    theta = rng.normal(size=(d, d))
    x = rng.normal(size=(n, d))
    interv = (rng.random((n, d)) < 0.2).astype(float)
    hp = O.LinGaussParams(obs_noise=0.3, mean_edge=0.2, sig_edge=1.5)
    mean = x @ (g * theta)
    want = (g * stats.norm.logpdf(theta, 0.2, 1.5)).sum() + ((1 - interv) * stats.norm.logpdf(x, mean, math.sqrt(0.3))).sum()
    got = float(O.lingauss_log_joint(_xt(g), _xt(theta), _xt(x), _xt(interv), hp))
    assert abs(got - want) < 1e-10 * abs(want)


def test_acyclicity_constraint_known_values():
    """h(G) = tr((I + G / d)^d) - d  (graph_utils.py:8-28): 0 for every DAG; 2-cycle on two nodes: tr([[1, .5], [.5, 1]]^2) - 2 = 0.5;
    directed 3-cycle: (I + P / 3)^3 = I + P + P^2 / 3 + P^3 / 27 with P^3 = I  ->  tr = 3 + 3 / 27, h = 1 / 9."""
    for d in (2, 5, 9):
        assert abs(float(O.acyclic_constr(_xt(np.triu(np.ones((d, d)), 1)), d))) < 1e-12
    assert abs(float(O.acyclic_constr(_xt([[0, 1], [1, 0]]), 2)) - 0.5) < 1e-14
    p3 = np.roll(np.eye(3), 1, axis=1)
    assert abs(float(O.acyclic_constr(_xt(p3), 3)) - 1.0 / 9.0) < 1e-14


def test_graph_priors_on_soft_graphs():
    """Erdos-Renyi (graph.py:93-108): E log p + (N - E) log(1 - p) with E = sum of the soft edges, N = d (d - 1) / 2 (the number of
    unordered pairs: a DAG has at most one edge per pair) and p = edges_per_node * d / N;  scale-free (graph.py:182-196):
    sum_j -3 log(1 + soft in-degree_j);  uniform: 0.  On a hard DAG the ER value is the log-pmf of N independent Bernoulli(p) pairs."""
    rng = np.random.default_rng(3)
    d = 6
    g = rng.random((d, d)) * (1 - np.eye(d))
    n_pairs = d * (d - 1) / 2
    p = 2 * d / n_pairs
    want_er = g.sum() * math.log(p) + (n_pairs - g.sum()) * math.log(1 - p)
    dag = np.triu((rng.random((d, d)) < 0.4).astype(float), 1)
    pair = dag[np.triu_indices(d, 1)]
    assert abs(float(O.log_graph_prior_soft(_xt(dag), O.GraphPrior("er", 2), d)) - stats.bernoulli.logpmf(pair, p).sum()) < 1e-10
    want_sf = (-3.0 * np.log(1.0 + g.sum(0))).sum()
    assert abs(float(O.log_graph_prior_soft(_xt(g), O.GraphPrior("er", 2), d)) - want_er) < 1e-10 * abs(want_er)
    assert abs(float(O.log_graph_prior_soft(_xt(g), O.GraphPrior("sf", 2), d)) - want_sf) < 1e-10 * abs(want_sf)
    assert float(O.log_graph_prior_soft(_xt(g), O.GraphPrior("uniform", 2), d)) == 0.0


def test_edge_probabilities_and_latent_log_prob():
    """p(G_ij = 1 | Z) = sigmoid(alpha u_i . v_j), no self-loops (dibs.py:168-184); log p(G | Z) = sum_{i != j} Bernoulli log-pmf
    (dibs.py:187-229)."""
    rng = np.random.default_rng(5)
    d, k, alpha = 4, 3, 0.7
    z = rng.normal(size=(d, k, 2))
    p = special.expit(alpha * z[:, :, 0] @ z[:, :, 1].T) * (1 - np.eye(d))
    assert np.allclose(np.asarray(O.edge_probs(_xt(z), alpha)), p, rtol=1e-12, atol=0)
    g = (rng.random((d, d)) < 0.5).astype(float) * (1 - np.eye(d))
    off = ~np.eye(d, dtype=bool)
    want = stats.bernoulli.logpmf(g[off], p[off]).sum()
    assert abs(float(O.latent_log_prob(_xt(g), _xt(z), alpha)) - want) < 1e-10 * abs(want)


def test_svgd_kernels_and_rmsprop_known_values():
    """k(x, x') = scale exp(-|x - x'|^2 / h) per factor, summed over Z and Theta for the joint kernel (kernel.py:20-30, 52-71);
    rmsprop(step, gamma = 0.9, eps = 1e-8): v <- 0.9 v + 0.1 g^2, x <- x - step g / sqrt(v + eps)."""
    cfg = O.Config()
    cfg.joint, cfg.scale_latent, cfg.h_latent, cfg.scale_theta, cfg.h_theta = True, 2.0, 5.0, 3.0, 7.0
    za, zb = _xt([[[1.0, 0.0]], [[0.0, 2.0]]]), _xt([[[0.0, 0.0]], [[0.0, 0.0]]])          # |za - zb|^2 = 5
    ta, tb = [_xt([1.0, 2.0, 2.0])], [_xt([0.0, 0.0, 0.0])]                                  # |ta - tb|^2 = 9
    want = 2.0 * math.exp(-5.0 / 5.0) + 3.0 * math.exp(-9.0 / 7.0)
    assert abs(float(O.f_kernel(cfg, za, ta, zb, tb)) - want) < 1e-14
    cfg.joint = False
    assert abs(float(O.f_kernel(cfg, za, None, zb, None)) - 2.0 * math.exp(-1.0)) < 1e-14
    cfg.optimizer, cfg.stepsize = "rmsprop", 0.005
    x, v = O.opt_update(cfg, _xt([2.0, -4.0]), _xt([1.0, 1.0]), _xt([0.0, 1.0]))
    assert np.allclose(np.asarray(v), [0.4, 0.9 + 1.6], rtol=1e-15)
    assert np.allclose(np.asarray(x), [1.0 - 0.005 * 2.0 / math.sqrt(0.4 + 1e-8), 1.0 + 0.005 * 4.0 / math.sqrt(2.5 + 1e-8)], rtol=1e-15)


def test_svgd_transform_is_liu_wang_2016():
    """One SVGD step with plain gradient descent moves every particle by  + stepsize * phi*(x_a),
        phi*(x_a) = 1/n sum_b [ k(x_b, x_a) grad log p(x_b) + grad_{x_b} k(x_b, x_a) ]        (Liu & Wang 2016, eq. 8)
    with the RBF kernel's gradient in closed form, -2 / h (x_b - x_a) k.  The reference hands -phi* to an optimizer that subtracts
    (svgd.py:194-224, 265): sign, 1/n scale and the kernel's argument order are what this pins."""
    rng = np.random.default_rng(7)
    M, d, k = 3, 2, 2
    cfg = O.Config()
    cfg.joint, cfg.scale_latent, cfg.h_latent, cfg.optimizer, cfg.stepsize = False, 1.5, 4.0, "gd", 0.01
    z = rng.normal(size=(M, d, k, 2))
    score = rng.normal(size=(M, d, k, 2))          # any "grad log p" per particle
    zt = _xt(z)
    kxx = O.kernel_mat(cfg, zt, None)
    phi_z, _ = O.svgd_phi(cfg, zt, None, kxx, _xt(score), None)
    z_new, _ = O.opt_update(cfg, phi_z, zt, torch.zeros_like(zt))
    want = z.copy()
    for a in range(M):
        acc = np.zeros_like(z[0])
        for b in range(M):
            kba = 1.5 * math.exp(-((z[b] - z[a]) ** 2).sum() / 4.0)
            acc += kba * score[b] + (-2.0 / 4.0) * (z[b] - z[a]) * kba
        want[a] += 0.01 * acc / M
    assert np.allclose(np.asarray(z_new), want, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("act,bias", [("relu", True), ("tanh", False), ("leakyrelu", True), ("sigmoid", True)])
def test_dense_nonlinear_gaussian_log_joint_against_numpy(act, bias):
    """Per-node MLP on the parent-masked inputs (nonlinearGaussian.py:35-81, 248-326): node j sees x o G[:, j]; Gaussian prior on every
    leaf, the first-layer weights counted only for actual parents; Gaussian likelihood on non-intervened entries."""
    rng = np.random.default_rng(11)
    d, n, hid = 4, 6, (5, 3)
    g = np.triu((rng.random((d, d)) < 0.6).astype(float), 1)
    x = rng.normal(size=(n, d))
    interv = (rng.random((n, d)) < 0.25).astype(float)
    sizes = (d,) + hid + (1,)
    Ws = [rng.normal(size=(d, sizes[i], sizes[i + 1])) for i in range(len(sizes) - 1)]
    bs = [rng.normal(size=(d, sizes[i + 1])) for i in range(len(sizes) - 1)]
    f = {"relu": lambda v: np.maximum(v, 0), "tanh": np.tanh, "sigmoid": special.expit, "leakyrelu": lambda v: np.where(v > 0, v, 0.01 * v)}[act]
    want = 0.0
    for j in range(d):
        h = x * g[:, j][None, :]
        for li, w in enumerate(Ws):
            h = h @ w[j] + (bs[li][j] if bias else 0.0)
            if li < len(Ws) - 1:
                h = f(h)
        want += ((1 - interv[:, j]) * stats.norm.logpdf(x[:, j], h[:, 0], math.sqrt(0.2))).sum()
        want += (g[:, j][:, None] * stats.norm.logpdf(Ws[0][j], 0.0, 1.3)).sum()
        want += sum(stats.norm.logpdf(w[j], 0.0, 1.3).sum() for w in Ws[1:])
        if bias:
            want += sum(stats.norm.logpdf(b[j], 0.0, 1.3).sum() for b in bs)
    hp = O.DenseNNParams(hidden_layers=hid, obs_noise=0.2, sig_param=1.3, activation=act, bias=bias)
    theta = []
    for w, b in zip(Ws, bs):
        theta.append(_xt(w))
        if bias:
            theta.append(_xt(b))
    got = float(O.densenn_log_joint(_xt(g), theta, _xt(x), _xt(interv), hp))
    assert abs(got - want) < 1e-10 * abs(want)


def test_score_function_estimator_against_exact_enumeration():
    """dibs.py:325-391 estimates  grad_Z log E_{p(G | Z)}[p(D | G)]  by Monte Carlo.  For three variables the expectation is a sum over
    the 64 graphs without self-loops: the estimator (4 000 samples) has to agree with autograd of the enumerated sum within its
    Monte-Carlo error.  Pins the estimator (ratio form, signs, the alpha inside p(G | Z)) to the quantity it is defined to estimate."""
    from oracle import prng
    d, k, t = 3, 2, 8.0
    x = _data(15, d, seed=3)
    rng = np.random.default_rng(1)
    z = _xt(rng.normal(size=(d, k, 2)) * 0.7)
    cfg = O.Config()
    cfg.n_grad_mc_samples, cfg.alpha_linear = 4000, 0.05
    xt, interv = _xt(x), torch.zeros((15, d), dtype=torch.float64)
    pairs = [(i, j) for i in range(d) for j in range(d) if i != j]
    zz = z.clone().requires_grad_(True)
    terms = []
    for bits in itertools.product((0.0, 1.0), repeat=len(pairs)):
        g = torch.zeros((d, d), dtype=torch.float64)
        for (i, j), b in zip(pairs, bits):
            g[i, j] = b
        terms.append(O.latent_log_prob(g, zz, cfg.alpha_linear * t) + O.bge_log_marginal(g, xt, interv, cfg.bge).detach())
    exact = torch.autograd.grad(torch.logsumexp(torch.stack(terms), 0), zz)[0].numpy().reshape(-1)
    est, _, _ = O.grad_z_likelihood_score_function(cfg, z, None, torch.zeros((), dtype=torch.float64), t, prng.PRNGKey(3), xt, interv)
    est = est.numpy().reshape(-1)
    cos = float(est @ exact / (np.linalg.norm(est) * np.linalg.norm(exact)))
    assert cos > 0.97 and abs(np.linalg.norm(est) / np.linalg.norm(exact) - 1.0) < 0.2, (cos, est, exact)
