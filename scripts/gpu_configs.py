"""BASELINE.json configs 2-5 on ONE GPU (configs 4/5 are quoted on 8 GPUs; here all particles sit on one device):
steps/s after warm-up, finiteness, and the final graphs' basic statistics."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dibs_amd import random
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd.target import make_linear_gaussian_equivalent_model, make_linear_gaussian_model, make_nonlinear_gaussian_model

def run(name, cfg, x, mask, warm, steps):
    eng = Engine(cfg); eng.set_data(x, mask); eng.init_particles(random.PRNGKey(1))
    eng.run(0, warm); eng.sync()
    t0 = time.perf_counter(); eng.run(warm, steps); eng.sync(); dt = time.perf_counter() - t0
    st = eng.get_state()
    z = st["z"]; ok = np.isfinite(z).all() and (st.get("theta") is None or np.isfinite(st["theta"]).all())
    g = (np.einsum("mik,mjk->mij", z[..., 0], z[..., 1]) > 0)
    print(f"{name}: {steps / dt:8.1f} steps/s ({dt / steps * 1e3:.3f} ms/step) finite={ok} mean edges/graph={g.sum((1, 2)).mean():.1f}", flush=True)
    eng.close()

which = sys.argv[1:] or ["2", "3", "4", "5"]
if "2" in which:
    data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=20, graph_prior_str="er")
    run("config2 marginal BGe d=20 M=32", make_config(n_vars=20, n_particles=32, n_observations=100), data.x, None, 50, 950)
if "3" in which:
    data, _, _ = make_linear_gaussian_model(key=random.PRNGKey(0), n_vars=50, graph_prior_str="er")
    run("config3 joint LinGauss d=50 M=128", make_config(n_vars=50, n_particles=128, n_observations=100, joint=True, likelihood="lingauss"),
        data.x, None, 20, 200)
if "4" in which:
    data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=50, graph_prior_str="er")
    run("config4 marginal BGe d=50 M=1024 (1 GPU)", make_config(n_vars=50, n_particles=1024, n_observations=100), data.x, None, 20, 100)
if "5" in which:
    data, _, _ = make_nonlinear_gaussian_model(key=random.PRNGKey(0), n_vars=100, graph_prior_str="sf")
    rng = np.random.default_rng(0)
    mask = np.zeros((100, 100), np.int32)
    for r in range(0, 100, 10):   # 10 intervention sets of ceil(0.1 d) nodes each (target.py:97-105 geometry)
        mask[r:r + 10, rng.choice(100, 10, replace=False)] = 1
    x = np.where(mask == 1, 0.0, data.x).astype(np.float32)
    run("config5 joint DenseNN d=100 M=256 interv (1 GPU)", make_config(n_vars=100, n_particles=256, n_observations=100, joint=True,
        likelihood="densenn", nn_hidden=(5,), graph_prior="sf", has_interventions=True), x, mask, 3, 10)
