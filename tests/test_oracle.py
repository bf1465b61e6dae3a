"""Oracle self-consistency (CPU): the C port (closed-form gradients) against the autograd restatement, the
golden fixtures, finite differences and independent library routines."""
import os

import numpy as np
import pytest
import torch

from conftest import make_data, rel_err
from dibs_amd._abi import make_config
from oracle import dibs_oracle as O, prng

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _xt(x):
    return torch.as_tensor(np.asarray(x, np.float64))


def test_bge_masked_slogdet_equals_compact_cholesky(c_oracle64):
    data, _, _ = make_data(8, seed=3, edges_per_node=2)
    cfg = make_config(n_vars=8, n_particles=1, n_observations=100)
    rng = np.random.default_rng(0)
    g = (rng.random((20, 8, 8)) < 0.4).astype(np.int32)
    g[:, np.arange(8), np.arange(8)] = 0
    a = c_oracle64.score_graphs(cfg, data.x, None, g, bge_mode=0)
    b = c_oracle64.score_graphs(cfg, data.x, None, g, bge_mode=1)
    assert rel_err(a, b) < 1e-12
    # against the torch restatement of linearGaussian.py:63-170
    for q in range(5):
        ref = O.bge_log_marginal(_xt(g[q]), _xt(data.x), torch.zeros(100, 8, dtype=torch.float64), O.BGeParams())
        assert abs(float(ref) - a[q]) < 1e-8 * abs(a[q])


def test_bge_with_interventions_matches_autograd_oracle(c_oracle64):
    data, _, _ = make_data(6, seed=4, edges_per_node=1)
    rng = np.random.default_rng(1)
    mask = (rng.random((100, 6)) < 0.15).astype(np.int32)
    g = (rng.random((6, 6, 6)) < 0.3).astype(np.int32)
    g[:, np.arange(6), np.arange(6)] = 0
    cfg = make_config(n_vars=6, n_particles=1, n_observations=100, edges_per_node=1, has_interventions=True)
    a = c_oracle64.score_graphs(cfg, data.x, mask, g)
    for q in range(6):
        ref = O.bge_log_marginal(_xt(g[q]), _xt(data.x), _xt(mask), O.BGeParams())
        assert abs(float(ref) - a[q]) < 1e-8 * abs(a[q])


def test_acyclicity_gradient_closed_form():
    # dh/dG = ((I + G/d)^(d-1))^T   (graph_utils.py:8-28)
    d = 7
    g = torch.rand(d, d, dtype=torch.float64, requires_grad=True)
    h = O.acyclic_constr(g, d)
    (gr,) = torch.autograd.grad(h, g)
    m = torch.eye(d, dtype=torch.float64) + g.detach() / d
    assert torch.allclose(gr, torch.linalg.matrix_power(m, d - 1).T, atol=1e-12)


@pytest.mark.parametrize("prior", ["er", "sf", "uniform"])
def test_marginal_step_cport_vs_autograd(c_oracle64, prior):
    d, M, S, Sa = 5, 3, 16, 4
    data, _, _ = make_data(d, seed=1)
    cfg = make_config(n_vars=d, n_particles=M, n_observations=100, edges_per_node=1, graph_prior=prior,
                      n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa)
    ocfg = O.Config(prior=O.GraphPrior(prior, 1), n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa)
    st = O.init_state(ocfg, prng.PRNGKey(1), M, d)
    cs = c_oracle64.new_state(cfg, prng.PRNGKey(1))
    assert np.array_equal(cs["z"], st.z.numpy()) and (cs["key"] == st.key).all()
    for t in range(3):
        st, aux = O.svgd_step(ocfg, st, _xt(data.x), torch.zeros(100, d, dtype=torch.float64), t, return_aux=True)
        dbg = c_oracle64.step(cfg, data.x, None, cs, t, debug=True, bge_mode=t % 2)
        lp = torch.stack([a["logprobs"] for a in aux["lik_aux"]]).numpy()
        assert rel_err(dbg["logprobs_z"], lp) < 1e-10
        assert rel_err(dbg["kxx"], aux["kxx"].numpy()) < 1e-10
        assert rel_err(dbg["grad_z"], (aux["dz_lik"] + aux["dz_prior"]).numpy()) < 1e-7
        assert rel_err(dbg["phi_z"], aux["phi_z"].numpy()) < 1e-7
        assert rel_err(cs["z"], st.z.numpy()) < 1e-7
        assert (cs["key"] == st.key).all()


def test_score_function_baseline_path(c_oracle64):
    d, M, S, Sa = 4, 2, 8, 2
    data, _, _ = make_data(d, seed=2)
    cfg = make_config(n_vars=d, n_particles=M, n_observations=100, edges_per_node=1, n_grad_mc_samples=S,
                      n_acyclicity_mc_samples=Sa, score_function_baseline=0.1)
    ocfg = O.Config(prior=O.GraphPrior("er", 1), n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa,
                    score_function_baseline=0.1)
    st = O.init_state(ocfg, prng.PRNGKey(5), M, d)
    cs = c_oracle64.new_state(cfg, prng.PRNGKey(5))
    for t in range(1, 3):
        st = O.svgd_step(ocfg, st, _xt(data.x), torch.zeros(100, d, dtype=torch.float64), t)
        c_oracle64.step(cfg, data.x, None, cs, t)
        assert rel_err(cs["baseline"], st.sf_baseline.numpy()) < 1e-10
        assert rel_err(cs["z"], st.z.numpy()) < 1e-6


@pytest.mark.parametrize("est", ["reparam", "score"])
def test_joint_lingauss_step_cport_vs_autograd(c_oracle64, est):
    d, M, S, Sa = 5, 3, 12, 3
    data, _, _ = make_data(d, seed=2, joint=True)
    rng = np.random.default_rng(0)
    mask = (rng.random((100, d)) < 0.1).astype(np.int32)
    cfg = make_config(n_vars=d, n_particles=M, n_observations=100, edges_per_node=1, joint=True, likelihood="lingauss",
                      grad_estimator_z=est, graph_prior="sf", n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa,
                      has_interventions=True)
    ocfg = O.Config(joint=True, likelihood="lingauss", alpha_linear=0.05, grad_estimator_z=est,
                    prior=O.GraphPrior("sf", 1), n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa)
    st = O.init_state(ocfg, prng.PRNGKey(3), M, d)
    cs = c_oracle64.new_state(cfg, prng.PRNGKey(3))
    th0 = np.stack([st.theta[m][0].numpy().reshape(-1) for m in range(M)])
    assert np.array_equal(cs["theta"], th0)
    for t in range(1, 3):
        st, aux = O.svgd_step(ocfg, st, _xt(data.x), _xt(mask), t, return_aux=True)
        dbg = c_oracle64.step(cfg, data.x, mask, cs, t, debug=True)
        gth = np.stack([aux["dtheta"][m][0].numpy().reshape(-1) for m in range(M)])
        th = np.stack([st.theta[m][0].numpy().reshape(-1) for m in range(M)])
        assert rel_err(dbg["grad_theta"], gth) < 1e-9
        assert rel_err(dbg["grad_z"], (aux["dz_lik"] + aux["dz_prior"]).numpy()) < 1e-6
        assert rel_err(cs["z"], st.z.numpy()) < 1e-6
        assert rel_err(cs["theta"], th) < 1e-9


def test_golden_config1_trajectory(c_oracle64):
    """BASELINE.json configs[0] (plumbing): the C port reproduces the committed 50-step trajectory and the
    committed autograd steps."""
    gold = np.load(os.path.join(GOLD, "config1_marginal_bge_d5.npz"))
    cfg = make_config(n_vars=5, n_particles=4, n_observations=100, edges_per_node=1)
    cs = c_oracle64.new_state(cfg, gold["key"])
    assert np.array_equal(cs["z"], gold["z_init"]) and (cs["key"] == gold["key_after_init"]).all()
    for t in range(50):
        c_oracle64.step(cfg, gold["x"], None, cs, t)
        if t + 1 <= 5:
            assert rel_err(cs["z"], gold["z_autograd_steps1to5"][t]) < 1e-7
        if t + 1 in (1, 2, 5, 10, 20, 50):
            assert rel_err(cs["z"], gold[f"z_cport_step{t + 1}"]) < 1e-9
    assert (cs["key"] == gold["key_after_50"]).all()


def test_f32_port_tracks_f64_single_step(c_oracle32, c_oracle64):
    d, M = 10, 4
    data, _, _ = make_data(d, seed=5)
    cfg = make_config(n_vars=d, n_particles=M, n_observations=100)
    a = c_oracle64.new_state(cfg, prng.PRNGKey(9))
    b = c_oracle32.new_state(cfg, prng.PRNGKey(9))
    c_oracle64.step(cfg, data.x, None, a, 0)
    c_oracle32.step(cfg, data.x, None, b, 0)
    assert rel_err(b["z"], a["z"]) < 1e-4  # the tolerance north_star states for fp32


@pytest.mark.parametrize("est,bias,act", [("reparam", True, "relu"), ("score", False, "tanh"), ("reparam", True, "leakyrelu")])
def test_joint_densenn_step_cport_vs_autograd(c_oracle64, est, bias, act):
    """DenseNonlinearGaussian (nonlinearGaussian.py:155-186, 248-326): stax-style init stream, forward and the manual
    backprop of the C port against autograd; interventions on."""
    from dibs_amd.models import DenseNonlinearGaussian
    d, M, N, H = 5, 3, 40, 4
    rng = np.random.default_rng(0)
    x = rng.normal(size=(N, d)).astype(np.float32).astype(np.float64)
    mask = (rng.random((N, d)) < 0.1).astype(np.int32)
    cfg = make_config(n_vars=d, n_particles=M, n_observations=N, edges_per_node=1, joint=True, likelihood="densenn",
                      grad_estimator_z=est, n_grad_mc_samples=8, n_acyclicity_mc_samples=2, nn_hidden=(H,),
                      nn_activation=act, nn_bias=bias, has_interventions=True)
    ocfg = O.Config(joint=True, likelihood="densenn", alpha_linear=0.05, grad_estimator_z=est, prior=O.GraphPrior("er", 1),
                    n_grad_mc_samples=8, n_acyclicity_mc_samples=2,
                    nn=O.DenseNNParams(hidden_layers=(H,), activation=act, bias=bias))
    flat = lambda th: np.stack([np.concatenate([leaf.numpy().reshape(-1) for leaf in th[m]]) for m in range(M)])
    st = O.init_state(ocfg, prng.PRNGKey(3), M, d)
    cs = c_oracle64.new_state(cfg, prng.PRNGKey(3))
    assert np.array_equal(cs["theta"], flat(st.theta))
    # the product's host-side sample_parameters follows the same key discipline
    k, sub = prng.split(prng.PRNGKey(3))
    ik, _ = prng.split(sub)
    _, tsub = prng.split(ik)
    nn = DenseNonlinearGaussian(n_vars=d, hidden_layers=(H,), activation=act, bias=bias)
    assert np.array_equal(nn.tree_to_flat(nn.sample_parameters(key=tsub, n_vars=d, n_particles=M)), flat(st.theta).astype(np.float32))
    for t in range(1, 3):
        st, aux = O.svgd_step(ocfg, st, _xt(x), _xt(mask), t, return_aux=True)
        dbg = c_oracle64.step(cfg, x, mask, cs, t, debug=True)
        gth = np.stack([np.concatenate([leaf.numpy().reshape(-1) for leaf in aux["dtheta"][m]]) for m in range(M)])
        assert rel_err(dbg["grad_theta"], gth) < 1e-9
        assert rel_err(dbg["grad_z"], (aux["dz_lik"] + aux["dz_prior"]).numpy()) < 1e-6
        assert rel_err(cs["z"], st.z.numpy()) < 1e-6
        assert rel_err(cs["theta"], flat(st.theta)) < 1e-9
