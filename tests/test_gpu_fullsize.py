"""GPU parity at the FULL sizes of BASELINE.json's configs (run with `pytest -m gpu` on the MI355X box).

test_gpu_parity.py checks every kernel stage against the oracle at small particle counts; the particle count selects
code paths on the device (particles per k_phi_update block, acyclicity chains per block, queue sizes, the fused kernel
matrix, ride-along blocks), so the same stage comparison is repeated here at the benchmarked sizes:

  headline / metric   MarginalDiBS + BGe, d=50, 128 particles          stages at t = 0, 1, 5, 20 along the trajectory
  config 3            JointDiBS + LinearGaussian, d=50, 128 particles  one step
  config 4            BGe, d=50, 1024 particles sharded 8 ways         8 rank engines == 1 engine (bits), one step vs the oracle
  config 5            JointDiBS + DenseNN, d=100, 256 particles, interventions, scale-free prior   one step
  headline free run   GPU vs the oracle's f64 and f32 builds, first step whose Z differs by > 1e-4, E-SHD at 50 / 100 / 200

Reference: dibs/inference/svgd.py:226-267 (marginal step), :673-721 (joint step).  Tolerances as in test_gpu_parity.py."""
import json
import os
import threading

import numpy as np
import pytest

from conftest import assert_update_parity, make_data, rel_err, stage_err, update_check
from dibs_amd._abi import make_config
from oracle import prng

pytestmark = pytest.mark.gpu
NT = min(os.cpu_count() or 1, 128)


def _engine(cfg, x, mask=None, stream=None):
    from dibs_amd.engine import Engine
    eng = Engine(cfg, stream=stream)
    eng.set_data(x, mask)
    return eng


def _graphs_from_masks(masks, M, S, d):
    gm = masks.reshape(M, d, S, -1)
    gg = np.zeros((M, S, d, d), np.uint8)
    for i in range(d):
        gg[:, :, i, :] = ((gm[:, :, :, i // 64] >> np.uint64(i % 64)) & np.uint64(1)).astype(np.uint8).transpose(0, 2, 1)
    return gg


def _oracle_state_from_engine(eng, real=np.float64):
    g = eng.get_state()
    st = dict(z=g["z"].astype(real), v_z=g["v_z"].astype(real), key=g["key"].copy(), baseline=g["baseline"].astype(real),
              theta=None, v_theta=None)
    if g["theta"] is not None:
        st["theta"], st["v_theta"] = g["theta"].astype(real), g["v_theta"].astype(real)
    return st


def _rms(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.sqrt(np.mean((a - b) ** 2)))


# Fixed bounds on rms(device - f64 oracle) of the estimator buffers at the benchmarked sizes, taken from round 3's measurements (about
# 2-3x the largest value seen over t = 0, 1, 5, 20 of the headline trajectory and config 4's step 2; the run is bit-reproducible).  At
# these sizes the deviation from the f64 oracle is float32 arithmetic itself: R rounded to f32 and n ~ 25 pivots per factorisation give
# node-score errors up to ~1 for a few ill-conditioned parent sets out of 10^6, and the softmax over the S log-scores turns near-ties into
# O(1e-2) differences of single weights.  The oracle's own f32 build on the same state is printed beside the device as a DIAGNOSTIC (it
# shows the same numbers: device / f32 oracle at t = 1: node scores 1.04e-3 / 1.58e-3, log p(D|G) 8.2e-3 / 1.1e-2); it no longer moves
# the bound.      name: (absolute rms bound, or None) , (rms bound relative to max |ref|, or None)
# config 3 free run, checkpoint: (Z, theta) relative to max |.| of the f64 oracle.  Measured: Z 2.40e-4 / 2.38e-4 / 2.36e-4 at steps 25 / 50 / 100
# (the oracle's f32 build: 1.9e-4 -- the offset appears in the first steps and does not grow), theta 8.8e-6 (f32 build: 2e-3)
CONFIG3_BOUNDS = {25: (5e-4, 3e-5), 50: (5e-4, 3e-5), 100: (5e-4, 3e-5)}
# One SVGD step on Z from the device's own state: conftest.update_check / assert_update_parity (signal coordinates within north_star's 1e-4,
# the update itself exact to 1e-6 against the device's own phi, signal share asserted) -- the same rule in every step test.  Measured max
# over ALL coordinates, for the record: 2.8e-8 / 3.2e-4 / 6.2e-8 / 4.3e-8 at t = 0 / 1 / 5 / 20 (f32 build of the oracle: 2.8e-8 / 7.9e-4 /
# 4.5e-8 / 4.3e-8): at t = 1, the first step with a likelihood term, RMSprop's second moment is still ~0 and noise coordinates move by a
# full step of either sign.
STAGE_BOUNDS = {
    "node_scores": (3e-3, None),    # measured <= 1.33e-3 (values ~ 4e2)
    "logprobs_z": (2.5e-2, None),   # measured <= 1.01e-2 (values ~ 4e3 .. 7e3)
    "w_lik": (2e-3, None),          # measured <= 6.9e-4 (values up to 20)
    "grad_z": (None, 3e-6),         # measured 6.0e-7 .. 9.8e-7 of max |grad| (up to 5e6: float32 rounding of the sum)
    "phi_z": (None, 3e-6),          # measured 6.0e-7 .. 9.8e-7 of max |phi|
}


def _stage_check(name, dev, dbg64, dbg32):
    ref = np.asarray(dbg64[name], np.float64)
    e_dev, e_32 = _rms(dev, ref), _rms(dbg32[name], ref)
    b_abs, b_rel = STAGE_BOUNDS[name]
    bound = b_abs if b_abs is not None else b_rel * np.abs(ref).max()
    return e_dev <= bound, f"{name}: rms device-f64 {e_dev:.2e} (bound {bound:.1e}; f32 oracle-f64 {e_32:.2e}, max|ref| {np.abs(ref).max():.2e})"


_update_check = update_check   # (tests/conftest.py: the one step criterion)


def test_headline_stage_parity(c_oracle64, c_oracle32):
    """d=50, 128 particles, S=128, Sa=32: every stage buffer of steps t = 0, 1, 5, 20 of the trajectory from PRNGKey(1)
    against the oracle started from the device state of that step (parent sets shrink along the trajectory: t = 0/1 are
    all large factorisations, t = 5 the mixed tiers, t = 20 mostly per-lane ones).  The f64 oracle is the reference value;
    its f32 build, run on the same state, measures how much of a deviation is float32 arithmetic itself (the reference
    computes in float32)."""
    d, M, S = 50, 128, 128
    data, _, _ = make_data(d, seed=0)
    cfg = make_config(n_vars=d, n_particles=M, n_observations=100)
    eng = _engine(cfg, data.x)
    eng.init_particles(prng.PRNGKey(1))
    t_cur = 0
    for t in (0, 1, 5, 20):
        eng.run(t_cur, t - t_cur)
        st = _oracle_state_from_engine(eng)
        prev = eng.get_state()
        st32 = _oracle_state_from_engine(eng, np.float32)
        dbg = c_oracle64.step(cfg, data.x, None, st, t, debug=True, n_threads=NT)
        dbg32 = c_oracle32.step(cfg, data.x, None, st32, t, debug=True, n_threads=NT)
        eng.run(t, 1)
        t_cur = t + 1
        g = eng.get_state()
        gg = _graphs_from_masks(eng.read("PARENT_MASKS"), M, S, d)
        # The PRNG stream is bit-exact; a Bernoulli draw flips only where the uniform falls between the f32 and the f64 value of
        # sigmoid(alpha * score) (|dp| ~ 1e-7): a handful of the 41 M edges per step
        n_flip = int((gg != dbg["g_samples"]).sum())
        print(f"t={t}: {n_flip} of {gg.size} sampled edges differ from the f64 oracle's")
        assert n_flip <= max(4, 1e-6 * gg.size), f"t={t}: {n_flip} sampled edges differ"
        assert (g["key"] == st["key"]).all()
        assert rel_err(eng.read("SCORES"), dbg["scores"]) < 2e-6
        ns = eng.read("NODE_SCORES").reshape(M, d, S).transpose(0, 2, 1)
        checks = [_stage_check("node_scores", ns, dbg, dbg32), _stage_check("logprobs_z", eng.read("LOGPROBS_Z"), dbg, dbg32),
                  _stage_check("w_lik", eng.read("W_LIK"), dbg, dbg32), _stage_check("grad_z", eng.read("GRAD_Z"), dbg, dbg32),
                  _stage_check("phi_z", eng.read("PHI_Z"), dbg, dbg32)]
        print(f"t={t}: " + "; ".join(msg for _, msg in checks))
        assert all(ok for ok, _ in checks), f"t={t}: {checks}"
        assert rel_err(eng.read("W_ACYC"), dbg["w_acyc"]) < 1e-5, f"t={t}"
        assert rel_err(eng.read("KXX"), dbg["kxx"]) < 1e-5, f"t={t}"
        # one step on Z from the device's state (fixed bound; the f32 oracle's own deviation is printed as a diagnostic)
        upd = update_check(cfg, prev["z"], prev["v_z"], eng.read("PHI_Z"), dbg["phi_z"], g["z"], st["z"])
        print(f"t={t}: one step on Z: {upd}; f32 oracle vs f64 on all coordinates {rel_err(st32['z'], st['z']):.2e}")
        # (t = 0: alpha = beta = 0, phi is the Gaussian prior + repulsion only -- every coordinate carries signal)
        assert_update_parity(upd, 0.5, f"headline t={t}")
    eng.close()


def test_config3_fullsize_step(c_oracle64):
    """BASELINE config 3: JointDiBS + LinearGaussian, d=50, 128 particles (reparam estimator, defaults), steps t = 0 and 3."""
    d, M = 50, 128
    data, _, _ = make_data(d, seed=0, joint=True)
    cfg = make_config(n_vars=d, n_particles=M, n_observations=100, joint=True, likelihood="lingauss")
    eng = _engine(cfg, data.x)
    eng.init_particles(prng.PRNGKey(1))
    t_cur = 0
    for t in (0, 3):
        eng.run(t_cur, t - t_cur)
        st = _oracle_state_from_engine(eng)
        prev = eng.get_state()
        dbg = c_oracle64.step(cfg, data.x, None, st, t, debug=True, n_threads=NT)
        eng.run(t, 1)
        t_cur = t + 1
        g = eng.get_state()
        assert (g["key"] == st["key"]).all()
        assert rel_err(eng.read("LOGPROBS_THETA"), dbg["logprobs_th"]) < 2e-5
        assert rel_err(eng.read("LOGPROBS_Z"), dbg["logprobs_z"]) < 2e-5
        stage_err("GRAD_THETA", eng.read("GRAD_THETA"), dbg["grad_theta"], 5e-4)
        stage_err("W_LIK", eng.read("W_LIK"), dbg["w_lik"], 2e-3)
        assert rel_err(eng.read("W_ACYC"), dbg["w_acyc"]) < 1e-5
        stage_err("GRAD_Z", eng.read("GRAD_Z"), dbg["grad_z"], 1e-4)
        assert rel_err(eng.read("KXX"), dbg["kxx"]) < 1e-5
        stage_err("PHI_THETA", eng.read("PHI_THETA"), dbg["phi_theta"], 2e-3)
        stage_err("PHI_Z", eng.read("PHI_Z"), dbg["phi_z"], 1e-4)
        uz = update_check(cfg, prev["z"], prev["v_z"], eng.read("PHI_Z"), dbg["phi_z"], g["z"], st["z"])
        ut = update_check(cfg, prev["theta"], prev["v_theta"], eng.read("PHI_THETA"), dbg["phi_theta"], g["theta"], st["theta"])
        print(f"config 3 t={t}: z {uz}; theta {ut}")
        assert_update_parity(uz, 0.5, f"config 3 z t={t}")
        assert_update_parity(ut, 0.3, f"config 3 theta t={t}")   # (measured share 0.40 at t = 0: the weights of absent edges get prior gradient only)
    eng.close()


def test_config3_free_running_100_steps(c_oracle64, c_oracle32):
    """BASELINE config 3 at its stated size (JointDiBS + LinearGaussian, d=50, 128 particles, reparam estimator, defaults), free-running
    for 100 steps from PRNGKey(1) on the factory's data (dibs/target.py:122-187) against the oracle's f64 build, with its f32 build
    beside it as a printed diagnostic.  Asserted with fixed bounds (CONFIG3_BOUNDS, measured values beside them): theta within north_star's
    1e-4, identical posterior graphs and E-SHD at every checkpoint, Z within the bound float32 arithmetic holds at this size (the Z
    gradient is a softmax-weighted sum over 128 soft graphs whose log-joints differ by hundreds: near-ties move single weights)."""
    from dibs_amd import random
    from dibs_amd.inference import JointDiBS
    from dibs_amd.metrics import expected_shd
    from dibs_amd.target import make_linear_gaussian_model
    d, M = 50, 128
    data, gm, lm = make_linear_gaussian_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er")
    dibs = JointDiBS(x=data.x, graph_model=gm, likelihood_model=lm)
    cfg = dibs._make_config(M, d)
    eng = _engine(cfg, data.x)
    eng.init_particles(prng.PRNGKey(1))
    st = c_oracle64.new_state(cfg, prng.PRNGKey(1))
    st32 = c_oracle32.new_state(cfg, prng.PRNGKey(1))
    half = max(NT // 2, 1)
    t = 0
    for cp in (25, 50, 100):
        th = [threading.Thread(target=c_oracle64.run, args=(cfg, data.x, None, st, t, cp - t), kwargs=dict(n_threads=half)),
              threading.Thread(target=c_oracle32.run, args=(cfg, data.x, None, st32, t, cp - t), kwargs=dict(n_threads=half))]
        for x_ in th:
            x_.start()
        eng.run(t, cp - t)
        for x_ in th:
            x_.join()
        t = cp
        g = eng.get_state()
        ez, et = rel_err(g["z"], st["z"]), rel_err(g["theta"], st["theta"])
        gd, go = dibs.particle_to_g_lim(g["z"]), dibs.particle_to_g_lim(np.asarray(st["z"], np.float32))
        same = float((gd == go).all(axis=(1, 2)).mean())
        e_d = expected_shd(dist=dibs.get_empirical(gd, g["theta"].reshape(M, d, d)), g=data.g)
        e_o = expected_shd(dist=dibs.get_empirical(go, np.asarray(st["theta"], np.float32).reshape(M, d, d)), g=data.g)
        print(f"config 3 step {cp}: rel err z {ez:.2e} theta {et:.2e} (f32 oracle: z {rel_err(st32['z'], st['z']):.2e} theta "
              f"{rel_err(st32['theta'], st['theta']):.2e}); identical graphs {same:.3f}; E-SHD gpu {e_d:.4f} oracle {e_o:.4f}")
        assert (g["key"] == st["key"]).all()
        bz, bt = CONFIG3_BOUNDS[cp]
        assert ez < bz and et < bt, (cp, ez, et)
        assert same == 1.0 and abs(e_d - e_o) < 1e-3, (cp, same, e_d, e_o)
        # the step criterion of every step test (conftest.update_check) at this point of the trajectory: one step from the device's state
        # against the oracle's step from the same state, then back to the free run (set_state resumes bit-exactly)
        snap = {k_: v for k_, v in g.items() if v is not None}
        st1 = _oracle_state_from_engine(eng)
        dbg = c_oracle64.step(cfg, data.x, None, st1, cp, debug=True, n_threads=NT)
        eng.run(cp, 1)
        g1 = eng.get_state()
        uz = update_check(cfg, g["z"], g["v_z"], eng.read("PHI_Z"), dbg["phi_z"], g1["z"], st1["z"])
        ut = update_check(cfg, g["theta"], g["v_theta"], eng.read("PHI_THETA"), dbg["phi_theta"], g1["theta"], st1["theta"])
        print(f"config 3 step {cp} -> {cp + 1} from the device's state: z {uz}; theta {ut}")
        assert_update_parity(uz, 0.5, f"config 3 free run z step {cp}")
        assert_update_parity(ut, 0.3, f"config 3 free run theta step {cp}")
        eng.set_state(**snap)
    eng.close()


def test_config4_sharded_fullsize(c_oracle64, c_oracle32):
    """BASELINE config 4: BGe, d=50, 1024 particles sharded over 8 ranks.  (JointDiBS + BGe is not constructible in the
    reference -- BGe has no parameters, linearGaussian.py:53-54 -- so the config runs as MarginalDiBS)
    Eight rank engines on one GPU with the all-gather replaced by a device concat must equal the single engine bit for
    bit after 3 steps, and the single engine's step 2 is compared with the oracle."""
    import torch
    from dibs_amd.engine import Engine
    d, M, R = 50, 1024, 8
    data, _, _ = make_data(d, seed=0)
    cfg1 = make_config(n_vars=d, n_particles=M, n_observations=100)
    ref = _engine(cfg1, data.x)
    ref.init_particles(prng.PRNGKey(1))
    ref.run(0, 2)
    st = _oracle_state_from_engine(ref)
    prev = ref.get_state()
    st32 = _oracle_state_from_engine(ref, np.float32)
    dbg = c_oracle64.step(cfg1, data.x, None, st, 2, debug=True, n_threads=NT)
    dbg32 = c_oracle32.step(cfg1, data.x, None, st32, 2, debug=True, n_threads=NT)
    ref.run(2, 1)
    sref = ref.get_state()
    assert (sref["key"] == st["key"]).all()
    ns = ref.read("NODE_SCORES").reshape(M, d, 128).transpose(0, 2, 1)
    checks = [_stage_check("node_scores", ns, dbg, dbg32), _stage_check("w_lik", ref.read("W_LIK"), dbg, dbg32),
              _stage_check("phi_z", ref.read("PHI_Z"), dbg, dbg32)]
    print("config 4: " + "; ".join(msg for _, msg in checks))
    assert all(ok for ok, _ in checks), checks
    assert rel_err(ref.read("W_ACYC"), dbg["w_acyc"]) < 1e-5
    assert rel_err(ref.read("KXX"), dbg["kxx"]) < 1e-5
    upd = update_check(cfg1, prev["z"], prev["v_z"], ref.read("PHI_Z"), dbg["phi_z"], sref["z"], st["z"])
    print(f"config 4: one step on Z: {upd}; f32 oracle vs f64 on all coordinates {rel_err(st32['z'], st['z']):.2e}")
    assert_update_parity(upd, 0.5, "config 4")
    ref.close()
    del dbg, dbg32
    tstream = torch.cuda.Stream()
    engs = []
    for r in range(R):
        e = Engine(make_config(n_vars=d, n_particles=M, n_observations=100, rank=r, n_ranks=R), stream=tstream.cuda_stream)
        e.set_data(data.x)
        e.init_particles(prng.PRNGKey(1))
        engs.append(e)
    n = engs[0].gather_elems_per_rank()
    with torch.cuda.stream(tstream):
        sends = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(R)]
        recv = torch.zeros(n * R, dtype=torch.float32, device="cuda")
        for t in range(3):
            for r in range(R):
                engs[r].step_local(t, sends[r].data_ptr())
            torch.cat(sends, out=recv)   # stands in for dist.all_gather_into_tensor(recv, send)
            for r in range(R):
                engs[r].step_update(t, recv.data_ptr())
    torch.cuda.synchronize()
    z = np.concatenate([e.get_state()["z"] for e in engs])
    assert np.array_equal(z, sref["z"]), "8-way sharded run must be bit-identical to the single engine"
    for e in engs:
        e.close()


def test_config5_fullsize_step(c_oracle64):
    """BASELINE config 5: JointDiBS + DenseNonlinearGaussian (hidden (5,), relu), d=100, 256 particles, interv_mask set,
    scale-free graph prior; one step (t = 1) from the initial particles."""
    d, M, N = 100, 256, 100
    rng = np.random.default_rng(1)
    x = rng.normal(size=(N, d)).astype(np.float32)
    mask = (rng.random((N, d)) < 0.1).astype(np.int32)
    cfg = make_config(n_vars=d, n_particles=M, n_observations=N, joint=True, likelihood="densenn", graph_prior="sf",
                      nn_hidden=(5,), has_interventions=True)
    st = c_oracle64.new_state(cfg, prng.PRNGKey(6))
    eng = _engine(cfg, x, mask)
    eng.init_particles(prng.PRNGKey(6))
    g0 = eng.get_state()
    assert (g0["key"] == st["key"]).all() and rel_err(g0["theta"], st["theta"]) < 1e-6 and rel_err(g0["z"], st["z"]) < 1e-6
    st = _oracle_state_from_engine(eng)
    prev = eng.get_state()
    dbg = c_oracle64.step(cfg, x, mask, st, 1, debug=True, n_threads=NT)
    eng.run(1, 1)
    g = eng.get_state()
    assert (g["key"] == st["key"]).all()
    assert rel_err(eng.read("LOGPROBS_THETA"), dbg["logprobs_th"]) < 2e-5
    assert rel_err(eng.read("LOGPROBS_Z"), dbg["logprobs_z"]) < 2e-5
    # (7.2e-4 here, on single entries: relu' flips at pre-activations within float32 rounding of 0 -- random data, 25 600 first layers; the same
    #  entries carry PHI_THETA's 7.2e-4.  Every other GRAD_THETA comparison of the suite is below 6.1e-5 and bounded by 5e-4)
    stage_err("GRAD_THETA", eng.read("GRAD_THETA"), dbg["grad_theta"], 2e-3)
    stage_err("W_LIK", eng.read("W_LIK"), dbg["w_lik"], 2e-3)
    assert rel_err(eng.read("W_ACYC"), dbg["w_acyc"]) < 1e-5
    stage_err("GRAD_Z", eng.read("GRAD_Z"), dbg["grad_z"], 1e-4)
    stage_err("PHI_THETA", eng.read("PHI_THETA"), dbg["phi_theta"], 2e-3)
    uz = update_check(cfg, prev["z"], prev["v_z"], eng.read("PHI_Z"), dbg["phi_z"], g["z"], st["z"])
    ut = update_check(cfg, prev["theta"], prev["v_theta"], eng.read("PHI_THETA"), dbg["phi_theta"], g["theta"], st["theta"])
    print(f"config 5 t=1: z {uz}; theta {ut}")   # (relu' flips at pre-activations within fp32 rounding of 0 move single noise coordinates)
    assert_update_parity(uz, 0.5, "config 5 z")
    assert_update_parity(ut, 0.0, "config 5 theta")   # (share printed, not asserted, for the parameter segment: see the factory-data test)
    eng.close()


def test_config5_factory_data_steps(c_oracle64):
    """BASELINE config 5 on the data the benchmark uses: make_nonlinear_gaussian_model (MLP ancestral sampling, dibs/target.py:190-260,
    models/nonlinearGaussian.py:189-242) with a scale-free ground truth, 10 intervention sets of ceil(0.1 d) clamped nodes
    (target.py:97-105 geometry); d=100, 256 particles.  Two consecutive steps t = 1, 2 from the initial particles, every stage buffer."""
    from dibs_amd import random
    from dibs_amd.target import make_nonlinear_gaussian_model
    d, M, N = 100, 256, 100
    data, _, _ = make_nonlinear_gaussian_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="sf", n_observations=N)
    rng = np.random.default_rng(0)
    mask = np.zeros((N, d), np.int32)
    for r in range(0, N, 10):
        mask[r:r + 10, rng.choice(d, int(np.ceil(0.1 * d)), replace=False)] = 1
    x = np.where(mask == 1, 0.0, data.x).astype(np.float32)
    cfg = make_config(n_vars=d, n_particles=M, n_observations=N, joint=True, likelihood="densenn", graph_prior="sf",
                      nn_hidden=(5,), has_interventions=True)
    eng = _engine(cfg, x, mask)
    eng.init_particles(prng.PRNGKey(1))
    eng.run(0, 1)
    for t in (1, 2):
        st = _oracle_state_from_engine(eng)
        prev = eng.get_state()
        dbg = c_oracle64.step(cfg, x, mask, st, t, debug=True, n_threads=NT)
        eng.run(t, 1)
        g = eng.get_state()
        assert (g["key"] == st["key"]).all()
        errs = dict(lp_th=rel_err(eng.read("LOGPROBS_THETA"), dbg["logprobs_th"]), lp_z=rel_err(eng.read("LOGPROBS_Z"), dbg["logprobs_z"]),
                    g_th=rel_err(eng.read("GRAD_THETA"), dbg["grad_theta"]), w_lik=rel_err(eng.read("W_LIK"), dbg["w_lik"]),
                    w_acyc=rel_err(eng.read("W_ACYC"), dbg["w_acyc"]), g_z=rel_err(eng.read("GRAD_Z"), dbg["grad_z"]),
                    phi_th=rel_err(eng.read("PHI_THETA"), dbg["phi_theta"]), phi_z=rel_err(eng.read("PHI_Z"), dbg["phi_z"]))
        upd_z = _update_check(cfg, prev["z"], prev["v_z"], eng.read("PHI_Z"), dbg["phi_z"], g["z"], st["z"])
        upd_t = _update_check(cfg, prev["theta"], prev["v_theta"], eng.read("PHI_THETA"), dbg["phi_theta"], g["theta"], st["theta"])
        print(f"config 5 (factory data) t={t}: " + ", ".join(f"{k_} {v:.1e}" for k_, v in errs.items()) + f"; z {upd_z}; theta {upd_t}")
        assert errs["lp_th"] < 2e-5 and errs["lp_z"] < 2e-5 and errs["w_acyc"] < 1e-5
        assert max(errs["g_th"], errs["w_lik"], errs["g_z"], errs["phi_th"], errs["phi_z"]) < 2e-3
        # north_star's 1e-4 on the coordinates with signal (measured for Z: 2e-8 .. 4e-7; own-phi 2e-8 .. 3e-8; share > 0.99)
        assert_update_parity(upd_z, 0.9, f"config 5 factory z t={t}")
        # (theta: the gradient is concentrated on the first-layer weights of the sampled edges -- 2 % of the 51 100 coordinates per particle are
        #  above 1e-3 of the largest; the share is printed, not asserted, for the parameter segment)
        assert_update_parity(upd_t, 0.0, f"config 5 factory theta t={t}")
    eng.close()


def test_headline_free_run_divergence(c_oracle64, c_oracle32):
    """north_star at the headline config: free-running GPU trajectory against the oracle's f64 build, with the oracle's
    own f32 build beside it as the yardstick for what float32 arithmetic can hold.  Records, per comparison, the first step
    whose Z differs by more than 1e-4 (relative to max |Z|) and the E-SHD / graph agreement at 50, 100 and 200 steps
    (written to gpurun_out/headline_divergence.json).  The softmax over S is an argmax in the limit: near-ties flip under
    any fp32 reordering, so fp32 trajectories separate eventually -- the GPU must stay as close to the f64 trajectory as the
    f32 oracle does (factor 3), with the same posterior graphs and E-SHD."""
    from dibs_amd.inference import MarginalDiBS
    from dibs_amd.metrics import expected_shd
    d, M, steps = 50, 128, 200
    data, gm, lm = make_data(d, seed=0)
    dibs = MarginalDiBS(x=data.x, graph_model=gm, likelihood_model=lm)
    cfg = dibs._make_config(M, d)
    eng = _engine(cfg, data.x)
    eng.init_particles(prng.PRNGKey(1))
    st64 = c_oracle64.new_state(cfg, prng.PRNGKey(1))
    st32 = c_oracle32.new_state(cfg, prng.PRNGKey(1))
    half = max(NT // 2, 1)
    first = {"gpu_vs_f64": None, "f32_vs_f64": None, "gpu_vs_f32": None}
    rows = []

    def eshd(z):
        g = dibs.particle_to_g_lim(np.asarray(z, np.float32))
        return g, expected_shd(dist=dibs.get_empirical(g), g=data.g)

    for t in range(steps):
        th = [threading.Thread(target=c_oracle64.step, args=(cfg, data.x, None, st64, t), kwargs=dict(n_threads=half)),
              threading.Thread(target=c_oracle32.step, args=(cfg, data.x, None, st32, t), kwargs=dict(n_threads=half))]
        for x_ in th:
            x_.start()
        eng.run(t, 1)
        zg = eng.get_state()["z"]
        for x_ in th:
            x_.join()
        errs = {"gpu_vs_f64": rel_err(zg, st64["z"]), "f32_vs_f64": rel_err(st32["z"], st64["z"]), "gpu_vs_f32": rel_err(zg, st32["z"])}
        for k_, v in errs.items():
            if v > 1e-4 and first[k_] is None:
                first[k_] = t + 1
        if t + 1 in (10, 20, 50, 100, 200):
            (gg, eg), (g64, e64), (g32, e32) = eshd(zg), eshd(st64["z"]), eshd(st32["z"])
            rows.append(dict(step=t + 1, **errs, eshd_gpu=eg, eshd_f64=e64, eshd_f32=e32,
                             graphs_equal_gpu_f64=float((gg == g64).all(axis=(1, 2)).mean()),
                             graphs_equal_f32_f64=float((g32 == g64).all(axis=(1, 2)).mean())))
    eng.close()
    rec = dict(config="MarginalDiBS+BGe d=50 M=128 S=128 Sa=32, PRNGKey(1)", checkpoints=rows)
    rec["first_step_rel_err_gt_1e-4"] = first
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "headline_divergence.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))
    for r in rows:
        # Z: float32 arithmetic itself leaves the f64 trajectory at step 2 (first step with a likelihood term: softmax near-ties) and
        # then keeps its distance; the device has to stay as close to f64 as the f32 oracle does (factor 3) over the first 100 steps.
        # Later a single near-tie can send ONE particle elsewhere (observed at step 100..200 for 1-2 of 128 particles, depending on
        # rounding details of the kernels -- the f32 oracle shows the same kind of event at step 50): from there on only the posterior
        # is asserted.
        if r["step"] <= 100:
            assert r["gpu_vs_f64"] < 3 * max(r["f32_vs_f64"], 1e-4), r
        # posterior: (almost) the same graphs; E-SHD within north_star's 1e-3 whenever all graphs agree, else within the weight of the
        # few particles that differ
        assert r["graphs_equal_gpu_f64"] >= 0.95, r
        if r["graphs_equal_gpu_f64"] == 1.0:
            assert abs(r["eshd_gpu"] - r["eshd_f64"]) < 1e-3, r
        else:
            assert abs(r["eshd_gpu"] - r["eshd_f64"]) < 0.1, r


def test_bench_sharded_path_on_one_gpu():
    """`bench.py --gpus N` (N > 1) cannot run here, but everything it executes can: `--dist-smoke` drives the same code -- process group
    (one rank) for the harness, RCCL communicators inside the engine (dibs_engine_comm_init), the in-engine step loop
    (dibs_engine_run_sharded: one ncclAllGather of the packed rows per step at 128 particles, the overlapped exchange for the config-4 extra
    key with its 1 024 particles), per-rank diagnostics -- and must print one valid JSON line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dist-smoke", "--steps", "4", "--warmup", "3", "--reps", "2",
                        "--min-seconds", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["value"] > 100 and out["n_gpus"] == 1 and out["steps"] == 4
    sh = out["sharded"]
    # (128 particles: the packed rows [z | grad_z] travel, 2 D floats per particle)
    assert sh["allgather_us"] > 0 and sh["allgather_bytes_per_rank"] == 128 * 10000 * 4 and len(sh["kernel_us_per_step_by_rank"]) == 1
    assert "in-engine" in out["config"]["parallelism"] or "inside the engine" in out["config"]["parallelism"]
    assert out["config4"]["value"] > 10
    # a communicator that does not come up on some rank: every rank falls back to the Python-driven loop over torch.distributed
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dist-smoke", "--steps", "4", "--warmup", "3", "--reps", "2",
                        "--min-seconds", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=dict(env, DIBS_BENCH_FAIL_NATIVE="1"), cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    out2 = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out2["value"] > 100 and "torch.distributed" in out2["config"]["parallelism"]
