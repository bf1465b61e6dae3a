"""Host-side logic of the drop-in facade (no GPU): constructor contract, error behaviour, metrics, factories."""
import numpy as np
import pytest

from conftest import make_data, rel_err
from dibs_amd import random
from dibs_amd.graph_utils import acyclic_constr_nograd, mat_is_dag
from dibs_amd.inference import JointDiBS, MarginalDiBS
from dibs_amd.metrics import ParticleDistribution, expected_edges, expected_shd, pairwise_structural_hamming_distance, \
    threshold_metrics
from dibs_amd.models import BGe, DenseNonlinearGaussian, ErdosReniDAGDistribution, LinearGaussian, \
    ScaleFreeDAGDistribution


def test_constructor_defaults_match_reference():
    data, gm, lm = make_data(6, edges_per_node=1)
    m = MarginalDiBS(x=data.x, graph_model=gm, likelihood_model=lm)  # svgd.py:60-83
    assert (m.kernel.h, m.optimizer, m.optimizer_param, m.grad_estimator_z) == (5.0, "rmsprop", {"stepsize": 0.005}, "score")
    assert m.alpha(3) == 3.0 and m.n_grad_mc_samples == 128 and m.n_acyclicity_mc_samples == 32
    assert m.interv_mask.shape == data.x.shape and not m.interv_mask.any()
    dataj, gmj, lmj = make_data(6, edges_per_node=1, joint=True)
    j = JointDiBS(x=dataj.x, graph_model=gmj, likelihood_model=lmj)  # svgd.py:425-448
    assert (j.kernel.h_latent, j.kernel.h_theta, j.grad_estimator_z) == (5.0, 500.0, "reparam")
    assert abs(j.alpha(10) - 0.5) < 1e-12
    cfg = j._make_config(8, None)
    assert (cfg.joint, cfg.likelihood, cfg.grad_estimator_z, cfg.n_dim) == (1, 1, 1, 6)


def test_error_behaviour():
    data, gm, lm = make_data(6, edges_per_node=1)
    with pytest.raises(ValueError):  # svgd.py:122
        MarginalDiBS(x=data.x, graph_model=gm, likelihood_model=lm, optimizer="adam")
    with pytest.raises(NotImplementedError):
        MarginalDiBS(x=data.x, graph_model=gm, likelihood_model=LinearGaussian(n_vars=6))
    with pytest.raises(NotImplementedError):  # JointDiBS + BGe is not constructible in the reference either
        JointDiBS(x=data.x, graph_model=gm, likelihood_model=lm)
    with pytest.raises(NotImplementedError):
        MarginalDiBS(x=data.x, graph_model=object(), likelihood_model=lm)
    with pytest.raises(NotImplementedError):  # linearGaussian.py:53-54
        lm.sample_parameters(key=random.PRNGKey(0), n_vars=6)
    with pytest.raises(KeyError):  # nonlinearGaussian.py:61
        DenseNonlinearGaussian(n_vars=4, hidden_layers=(5,), activation="gelu")
    with pytest.raises(AssertionError):  # linearGaussian.py:47
        BGe(n_vars=5, alpha_lambd=5.5)
    m = MarginalDiBS(x=data.x, graph_model=gm, likelihood_model=lm, grad_estimator_z="nope")
    with pytest.raises(ValueError):  # dibs.py:318
        m._make_config(4, None)


def test_particle_to_g_lim_and_edge_probs():
    data, gm, lm = make_data(5)
    m = MarginalDiBS(x=data.x, graph_model=gm, likelihood_model=lm)
    z = random.normal(random.PRNGKey(0), (3, 5, 5, 2))
    g = m.particle_to_g_lim(z)
    s = np.einsum("mik,mjk->mij", z[..., 0], z[..., 1])
    assert g.dtype == np.int32 and (np.diagonal(g, axis1=1, axis2=2) == 0).all()
    off = ~np.eye(5, dtype=bool)
    assert np.array_equal(g[:, off], (s > 0)[:, off].astype(np.int32))
    p = m.edge_probs(z, 4)
    assert np.allclose(p[:, off], 1 / (1 + np.exp(-4 * s[:, off])), atol=1e-6) and (np.diagonal(p, axis1=1, axis2=2) == 0).all()


def test_shd_and_expected_shd():
    g = np.zeros((4, 4), np.int32)
    g[0, 1] = g[1, 2] = 1
    a = g.copy()
    a[0, 1], a[1, 0] = 0, 1  # a reversal counts once
    b = g.copy()
    b[2, 3] = 1  # an insertion
    assert pairwise_structural_hamming_distance(x=np.stack([a, b, g]), y=g[None]).squeeze(1).tolist() == [1, 1, 0]
    cyc = np.zeros((4, 4), np.int32)
    cyc[0, 1] = cyc[1, 0] = 1
    dist = ParticleDistribution(logp=np.log([0.25, 0.25, 0.25, 0.25]), g=np.stack([a, b, g, cyc]))
    assert abs(expected_shd(dist=dist, g=g) - (1 + 1 + 0) / 3) < 1e-12  # cyclic particle dropped, weights renormalised
    assert abs(expected_edges(dist=dist) - (2 + 3 + 2) / 3) < 1e-12
    allcyc = ParticleDistribution(logp=np.log([1.0]), g=cyc[None])
    assert expected_shd(dist=allcyc, g=g) == 4 * 3 / 2
    tm = threshold_metrics(dist=dist, g=g)
    assert 0.5 <= tm["roc_auc"] <= 1.0


def test_acyclicity_f32_filter_matches_reference_semantics():
    d = 20
    chain = np.zeros((d, d), np.float32)
    for i in range(d - 1):
        chain[i, i + 1] = 1
    assert acyclic_constr_nograd(chain, d) == 0 and mat_is_dag(chain)
    two = chain.copy()
    two[1, 0] = 1
    assert acyclic_constr_nograd(two, d) > 0 and not mat_is_dag(two)
    # a single long cycle falls below float32 resolution -> counted as a DAG, like the reference
    ring = chain.copy()
    ring[d - 1, 0] = 1
    assert not mat_is_dag(ring) and acyclic_constr_nograd(ring, d) == 0


def test_graph_models_and_factories():
    er = ErdosReniDAGDistribution(20, n_edges_per_node=2)
    assert abs(er.p - 40 / 190) < 1e-12
    g = er.sample_G(random.PRNGKey(0))
    assert mat_is_dag(g) and g.shape == (20, 20)
    soft = np.full((20, 20), 0.3)
    e = soft.sum()
    assert abs(er.unnormalized_log_prob_soft(soft_g=soft) - (e * np.log(er.p) + (190 - e) * np.log(1 - er.p))) < 1e-9
    sf = ScaleFreeDAGDistribution(20)
    gs = sf.sample_G(random.PRNGKey(1))
    assert mat_is_dag(gs) and gs.sum() >= 19
    assert abs(sf.unnormalized_log_prob_soft(soft_g=soft) - np.sum(-3 * np.log(1 + soft.sum(0)))) < 1e-9
    data, gm, lm = make_data(12, joint=True)
    assert data.x.shape == (100, 12) and data.x_ho.shape == (100, 12) and len(data.x_interv) == 10
    interv, xi = data.x_interv[0]
    assert len(interv) == 2 and all(np.all(xi[:, k] == 0) for k in interv)
    # a data set is a deterministic function of its key
    data2, _, _ = make_data(12, joint=True)
    assert np.array_equal(data.x, data2.x) and np.array_equal(data.g, data2.g)


def test_lingauss_sample_parameters_stream():
    lm = LinearGaussian(n_vars=4)
    th = lm.sample_parameters(key=random.PRNGKey(3), n_vars=4, n_particles=5)
    raw = random.normal(random.PRNGKey(3), (5, 4, 4))
    assert th.shape == (5, 4, 4) and np.allclose(th, raw + np.sign(raw) * 0.5) and np.abs(th).min() >= 0.5
    nn = DenseNonlinearGaussian(n_vars=4, hidden_layers=(3,))
    tree = nn.sample_parameters(key=random.PRNGKey(4), n_vars=4, n_particles=2)
    assert tree[0][0].shape == (2, 4, 4, 3) and tree[0][1].shape == (2, 4, 3) and tree[1] == () and tree[2][0].shape == (2, 4, 3, 1)
    flat = nn.tree_to_flat(tree)
    back = nn.flat_to_tree(flat, 4)
    assert flat.shape == (2, 4 * 4 * 3 + 4 * 3 + 4 * 3 + 4) and np.array_equal(back[2][1], tree[2][1])


def test_model_log_prob_helpers_match_oracle():
    """LinearGaussian / DenseNonlinearGaussian.log_prob_parameters + log_likelihood (host-side helpers with the reference's
    names) add up to the oracle's log joint (linearGaussian.py:278-338, nonlinearGaussian.py:248-326)."""
    import torch
    from oracle import dibs_oracle as O
    from dibs_amd import random
    from dibs_amd.models import LinearGaussian, DenseNonlinearGaussian
    rng = np.random.default_rng(0)
    d, N = 6, 20
    x = rng.normal(size=(N, d))
    it = (rng.random((N, d)) < 0.2).astype(np.int32)
    g = (rng.random((d, d)) < 0.4).astype(np.int32)
    np.fill_diagonal(g, 0)
    t64 = lambda a: torch.as_tensor(np.asarray(a, np.float64))
    lin = LinearGaussian(n_vars=d)
    th = rng.normal(size=(d, d))
    ref = float(O.lingauss_log_joint(t64(g), t64(th), t64(x), t64(it), O.LinGaussParams()))
    got = lin.log_prob_parameters(theta=th, g=g) + lin.log_likelihood(x=x, theta=th, g=g, interv_targets=it)
    assert abs(got - ref) < 1e-8 * abs(ref)
    nn = DenseNonlinearGaussian(n_vars=d, hidden_layers=(4,), activation="tanh")
    theta = nn.sample_parameters(key=random.PRNGKey(3), n_vars=d)
    tht = [t64(leaf) for lay in theta for leaf in lay]   # the oracle's container: flat list of leaves
    ref = float(O.densenn_log_joint(t64(g), tht, t64(x), t64(it), O.DenseNNParams(hidden_layers=(4,), activation="tanh")))
    got = nn.log_prob_parameters(theta=theta, g=g) + nn.log_likelihood(x=x, theta=theta, g=g, interv_targets=it)
    assert abs(got - ref) < 1e-6 * abs(ref)


def test_dibs_base_host_helpers_match_autograd_oracle():
    """public helpers of the reference's DiBS base that live on the host (dibs/inference/dibs.py:102-247, 557-623): graph samples from given
    keys / noise, edge log-probabilities, log p(G | Z) and its Z-gradient (closed form here, autograd in the oracle), soft graph prior,
    acyclicity value of a Gumbel-soft graph"""
    import torch
    from oracle import dibs_oracle as O, prng
    from dibs_amd import random
    from dibs_amd.inference import MarginalDiBS
    d, k, t = 6, 4, 3
    data, gm, lm = make_data(d, seed=2)
    dibs = MarginalDiBS(x=data.x, graph_model=gm, likelihood_model=lm, alpha_linear=0.7, tau=1.3)
    rng = np.random.default_rng(0)
    z = rng.normal(size=(d, k, 2)).astype(np.float32)
    zt = torch.as_tensor(z.astype(np.float64))
    alpha = 0.7 * t
    key = random.PRNGKey(5)
    # sample_g: same stream as the oracle's (jax.random.bernoulli layering), int32, zero diagonal
    p = dibs.edge_probs(z, t)
    g = dibs.sample_g(p, key, 7)
    assert g.dtype == np.int32 and g.shape == (7, d, d) and not g[:, np.arange(d), np.arange(d)].any()
    assert np.array_equal(g, O.sample_g(torch.as_tensor(p.astype(np.float64)), prng.PRNGKey(5), 7, "legacy").numpy().astype(np.int32))
    eps = random.logistic(key, (d, d))
    soft = dibs.particle_to_soft_graph(z, eps, t)
    ref = O.particle_to_soft_graph(zt, torch.as_tensor(eps.astype(np.float64)), alpha, 1.3).numpy()
    assert rel_err(soft, ref) < 1e-6 and not np.diag(soft).any()
    hard = dibs.particle_to_hard_graph(z, eps, t)
    assert hard.dtype == np.float32 and np.array_equal(hard, dibs._zero_diag(((eps + np.float32(alpha) * dibs._scores(z)) > 0).astype(np.float32)))
    lp, l1p = dibs.edge_log_probs(z, t)
    s = dibs._scores(z).astype(np.float64)
    off = ~np.eye(d, dtype=bool)
    assert rel_err(lp[off], -np.log1p(np.exp(-alpha * s))[off]) < 1e-6 and rel_err(l1p[off], -np.log1p(np.exp(alpha * s))[off]) < 1e-6
    assert not np.diag(lp).any() and not np.diag(l1p).any()
    # latent log-probability and its gradient (the reference differentiates latent_log_prob; here the closed form)
    llp = dibs.latent_log_prob(g[0], z, t)
    assert abs(float(llp) - float(O.latent_log_prob(torch.as_tensor(g[0].astype(np.float64)), zt, alpha))) < 1e-4
    grads = dibs.eltwise_grad_latent_log_prob(g, z, t)
    assert grads.shape == (7, d, k, 2)
    for q in range(7):
        zz = zt.clone().requires_grad_(True)
        (gr,) = torch.autograd.grad(O.latent_log_prob(torch.as_tensor(g[q].astype(np.float64)), zz, alpha), zz)
        assert rel_err(grads[q], gr.numpy()) < 1e-5
    # graph prior on the edge probabilities, acyclicity value of the soft graph
    assert abs(float(dibs.log_graph_prior_particle(z, t)) -
               float(O.log_graph_prior_soft(O.edge_probs(zt, alpha), O.GraphPrior("er", gm.n_edges_per_node), d))) < 1e-4
    h = dibs.constraint_gumbel(z, eps, t)
    assert abs(float(h) - float(O.acyclic_constr(torch.as_tensor(ref), d))) < 1e-4 * max(1.0, abs(float(h)))
