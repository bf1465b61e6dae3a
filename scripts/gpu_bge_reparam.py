"""MarginalDiBS + BGe with grad_estimator_z="reparam" (soft-graph BGe, kernels_bge_soft.h): ms/step at two sizes and the facade."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dibs_amd import random
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd.inference import MarginalDiBS
from dibs_amd.target import make_linear_gaussian_equivalent_model
for d, M in ((20, 32), (50, 128)):
    data, gm, lm = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er")
    cfg = make_config(n_vars=d, n_particles=M, n_observations=100, grad_estimator_z="reparam")
    eng = Engine(cfg); eng.set_data(data.x); eng.init_particles(random.PRNGKey(1))
    eng.run(0, 2); eng.sync()
    t0 = time.perf_counter(); eng.run(2, 10); eng.sync(); dt = (time.perf_counter() - t0) / 10
    z = eng.get_state()["z"]
    print(f"BGe reparam d={d} M={M}: {dt*1e3:.2f} ms/step ({1/dt:.1f} steps/s) finite={np.isfinite(z).all()}", flush=True)
    eng.close()
data, gm, lm = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=10, graph_prior_str="er")
g = MarginalDiBS(x=data.x, graph_model=gm, likelihood_model=lm, grad_estimator_z="reparam").sample(key=random.PRNGKey(2), n_particles=8, steps=50)
print("sample() reparam ->", g.shape, "mean edges", g.sum((1, 2)).mean())
