"""GPU tests (`pytest -m gpu`) of the public estimator methods of the reference's DiBS base class that run on the device with the
caller's PRNG keys: eltwise_grad_z_likelihood, eltwise_grad_theta_likelihood, eltwise_grad_latent_prior, eltwise_log_joint_prob
(dibs/inference/dibs.py:255-269, 295-321, 467-485, 626-658).  Checker: the per-particle torch-autograd restatement in oracle/dibs_oracle.py
with the SAME explicit subkeys, and -- for the key plumbing -- the engine's own step: with the keys of one SVGD step passed explicitly the
two parts add up to the GRAD_Z buffer of that step."""
import numpy as np
import pytest
import torch

from conftest import make_data, rel_err
from dibs_amd import random
from dibs_amd.inference import JointDiBS, MarginalDiBS
from oracle import dibs_oracle as O, prng

pytestmark = pytest.mark.gpu


def _xt(a):
    return torch.as_tensor(np.asarray(a, np.float64))


def _subkeys(seed, M):
    return np.stack([prng.split(prng.PRNGKey(seed + 7 * m), 2, "legacy")[1] for m in range(M)]).astype(np.uint32)   # arbitrary, unrelated keys


@pytest.mark.parametrize("prior,est,t", [("er", "score", 2), ("sf", "score", 0), ("er", "reparam", 3)])
def test_marginal_estimators_with_explicit_keys(prior, est, t):
    d, M, S, Sa = 6, 3, 16, 4
    data, gm, lm = make_data(d, seed=2, prior=prior)
    dibs = MarginalDiBS(x=data.x, graph_model=gm, likelihood_model=lm, n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa,
                        grad_estimator_z=est, alpha_linear=0.4)
    ocfg = O.Config(prior=O.GraphPrior(prior, gm.n_edges_per_node if prior != "uniform" else 2), n_grad_mc_samples=S,
                    n_acyclicity_mc_samples=Sa, grad_estimator_z=est, alpha_linear=0.4)
    rng = np.random.default_rng(1)
    zs = (rng.normal(size=(M, d, d, 2)) / np.sqrt(d)).astype(np.float32)
    bl = np.zeros(M, np.float32)
    ka, kb = _subkeys(11, M), _subkeys(23, M)
    gz, bl_out = dibs.eltwise_grad_z_likelihood(zs, None, bl, t, ka)
    gp = dibs.eltwise_grad_latent_prior(zs, kb, t)
    assert gz.shape == gp.shape == (M, d, d, 2) and bl_out.shape == (M,)
    x, iv = _xt(data.x), torch.zeros(100, d, dtype=torch.float64)
    fn = O.grad_z_likelihood_score_function if est == "score" else O.grad_z_likelihood_gumbel
    for m in range(M):
        ref, _, _ = fn(ocfg, _xt(zs[m]), None, torch.zeros((), dtype=torch.float64), t, ka[m], x, iv)
        assert rel_err(gz[m], ref.numpy()) < (2e-3 if est == "score" else 5e-4), (m, rel_err(gz[m], ref.numpy()))
        refp = O.grad_latent_prior(ocfg, _xt(zs[m]), kb[m], t, 1.0 / np.sqrt(d))
        assert rel_err(gp[m], refp.numpy()) < 2e-5, (m, rel_err(gp[m], refp.numpy()))
    # eltwise_log_joint_prob: log p(D | G) of hard graphs on the training data
    g = (rng.random((5, d, d)) < 0.3).astype(np.int32)
    g[:, np.arange(d), np.arange(d)] = 0
    lp = dibs.eltwise_log_joint_prob(g, None, None)
    for q in range(5):
        assert abs(lp[q] - float(O.bge_log_marginal(_xt(g[q]), x, iv, O.BGeParams()))) < 2e-5 * abs(lp[q])


def test_joint_estimators_with_explicit_keys():
    d, M, S, Sa, t = 5, 3, 16, 4, 2
    data, gm, lm = make_data(d, seed=3, joint=True)
    dibs = JointDiBS(x=data.x, graph_model=gm, likelihood_model=lm, n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa)
    ocfg = O.Config(joint=True, likelihood="lingauss", prior=O.GraphPrior("er", gm.n_edges_per_node), n_grad_mc_samples=S,
                    n_acyclicity_mc_samples=Sa, grad_estimator_z="reparam", alpha_linear=0.05)
    rng = np.random.default_rng(4)
    zs = (rng.normal(size=(M, d, d, 2)) / np.sqrt(d)).astype(np.float32)
    th = rng.normal(size=(M, d, d)).astype(np.float32)
    ka, kb, kc = _subkeys(5, M), _subkeys(6, M), _subkeys(9, M)
    gz, _ = dibs.eltwise_grad_z_likelihood(zs, th, np.zeros(M, np.float32), t, ka)
    gth = dibs.eltwise_grad_theta_likelihood(zs, th, t, kb)
    gp = dibs.eltwise_grad_latent_prior(zs, kc, t)
    assert gth.shape == (M, d, d)
    x, iv = _xt(data.x), torch.zeros(100, d, dtype=torch.float64)
    for m in range(M):
        ref, _, _ = O.grad_z_likelihood_gumbel(ocfg, _xt(zs[m]), [_xt(th[m])], torch.zeros((), dtype=torch.float64), t, ka[m], x, iv)
        assert rel_err(gz[m], ref.numpy()) < 5e-4
        rth, _ = O.grad_theta_likelihood(ocfg, _xt(zs[m]), [_xt(th[m])], t, kb[m], x, iv)
        assert rel_err(gth[m], rth[0].numpy()) < 5e-4
        assert rel_err(gp[m], O.grad_latent_prior(ocfg, _xt(zs[m]), kc[m], t, 1.0 / np.sqrt(d)).numpy()) < 2e-5
    g = (rng.random((4, d, d)) < 0.3).astype(np.int32)
    g[:, np.arange(d), np.arange(d)] = 0
    lp = dibs.eltwise_log_joint_prob(g, th[0], None)
    for q in range(4):
        assert abs(lp[q] - float(O.log_joint_prob(ocfg, _xt(g[q]), [_xt(th[0])], x, iv))) < 2e-5 * abs(lp[q])


def test_explicit_keys_reproduce_the_engines_own_step():
    """keys of one SVGD step (rows 1..M of split(carry, M + 1), svgd.py:245-251) passed explicitly: likelihood part + prior part == the
    GRAD_Z buffer of dibs_engine_run for that step, at the headline kernels' sizes (d = 50: bf16 / f16 acyclicity kernel, queued BGe)"""
    from dibs_amd._abi import make_config
    from dibs_amd.engine import Engine
    d, M, t = 50, 8, 3
    data, gm, lm = make_data(d, seed=0)
    dibs = MarginalDiBS(x=data.x, graph_model=gm, likelihood_model=lm)
    cfg = dibs._make_config(M, d)
    eng = Engine(cfg)
    eng.set_data(data.x)
    eng.init_particles(random.PRNGKey(1))
    eng.run(0, t)
    st = eng.get_state()
    carry = st["key"]
    k1 = prng.split(carry, M + 1, "legacy")
    k2 = prng.split(k1[0], M + 1, "legacy")
    eng.run(t, 1)
    total = eng.read("GRAD_Z").reshape(M, d, d, 2)
    eng.close()
    gz, _ = dibs.eltwise_grad_z_likelihood(st["z"], None, st["baseline"], t, k1[1:])
    gp = dibs.eltwise_grad_latent_prior(st["z"], k2[1:], t)
    assert rel_err(gz + gp, total) < 2e-6


@pytest.mark.parametrize("joint,overlapped", [(False, False), (False, True), (True, False), (True, True)])
def test_native_sharded_loop_world1_is_bit_identical_to_engine_run(joint, overlapped):
    """dibs_engine_run_sharded: the step loop of a sharded run in C with the collective issued by the engine itself through RCCL
    (ncclCommInitRank from a unique id, ncclAllGather in place on the engine's streams).  One GPU = a communicator of one rank: the whole
    RCCL call path runs and the result must equal dibs_engine_run bit for bit, for both exchange protocols, including a chunk boundary
    and a state replaced between chunks (checkpoint restore: the gathered values are stale)."""
    from dibs_amd.engine import Engine
    d, M = (12, 8) if joint else (50, 16)
    data, gm, lm = make_data(d, seed=1, joint=joint)
    dibs = (JointDiBS if joint else MarginalDiBS)(x=data.x, graph_model=gm, likelihood_model=lm, n_grad_mc_samples=32, n_acyclicity_mc_samples=8)
    a, b = dibs._new_engine(M, d), dibs._new_engine(M, d)
    try:
        b.comm_init(b.comm_unique_ids(2 if overlapped else 1))
        for e in (a, b):
            e.init_particles(random.PRNGKey(3))
        a.run(0, 5)
        b.run_sharded(0, 3, overlapped)
        b.run_sharded(3, 2, overlapped)
        sa, sb = a.get_state(), b.get_state()
        assert np.array_equal(sa["z"], sb["z"]) and (sa["key"] == sb["key"]).all()
        if joint:
            assert np.array_equal(sa["theta"], sb["theta"])
        z_all, th_all = b.gather_particles()
        assert np.array_equal(z_all, sa["z"]) and (not joint or np.array_equal(th_all, sa["theta"]))
        # replaced state between chunks
        c = dibs._new_engine(M, d)
        c.init_particles(random.PRNGKey(8))
        c.run(0, 2)
        sc = {k: v for k, v in c.get_state().items() if v is not None}
        c.close()
        a.set_state(**sc)
        b.set_state(**sc)
        a.run(2, 3)
        b.run_sharded(2, 3, overlapped)
        assert np.array_equal(a.get_state()["z"], b.get_state()["z"])
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("joint", [False, True])
def test_native_sharded_loop_mixed_protocols(joint):
    """An overlapped chunk followed by a packed chunk (and by plain dibs_engine_run) on ONE engine with two communicators: the packed /
    single-rank steps move the particles without touching plane 0, so the values gathered at the end of the overlapped chunk are stale
    afterwards -- gather_particles must re-gather and the next overlapped chunk must exchange again (engine.hip: step_update clears
    vals_fresh).  Everything bit-identical to dibs_engine_run."""
    d, M = (12, 8) if joint else (50, 16)
    data, gm, lm = make_data(d, seed=1, joint=joint)
    dibs = (JointDiBS if joint else MarginalDiBS)(x=data.x, graph_model=gm, likelihood_model=lm, n_grad_mc_samples=32, n_acyclicity_mc_samples=8)
    a, b = dibs._new_engine(M, d), dibs._new_engine(M, d)
    try:
        b.comm_init(b.comm_unique_ids(2))
        for e in (a, b):
            e.init_particles(random.PRNGKey(5))
        b.run_sharded(0, 2, True)      # overlapped: plane 0 holds the particles after step 1
        b.run_sharded(2, 2, False)     # packed: particles move, plane 0 does not
        a.run(0, 4)
        z_all, th_all = b.gather_particles()
        assert np.array_equal(z_all, a.get_state()["z"]), "gather_particles returned the stale plane of the overlapped chunk"
        assert not joint or np.array_equal(th_all, a.get_state()["theta"])
        b.run_sharded(4, 2, True)      # overlapped again: must not run phase B on the stale values
        a.run(4, 2)
        assert np.array_equal(a.get_state()["z"], b.get_state()["z"])
        b.run_sharded(6, 1, True)
        b.run_sharded(7, 1, False)
        b.run_sharded(8, 1, True)
        a.run(6, 3)
        sa, sb = a.get_state(), b.get_state()
        assert np.array_equal(sa["z"], sb["z"]) and (sa["key"] == sb["key"]).all()
        z_all, _ = b.gather_particles()
        assert np.array_equal(z_all, sa["z"])
    finally:
        a.close()
        b.close()
