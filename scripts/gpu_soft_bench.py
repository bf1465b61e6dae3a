import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from dibs_amd import random
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd.target import make_linear_gaussian_equivalent_model
d, M = 50, 128
data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er", n_observations=100)
cfg = make_config(n_vars=d, n_particles=M, n_observations=100, grad_estimator_z="reparam")
eng = Engine(cfg); eng.set_data(data.x); eng.init_particles(random.PRNGKey(1))
eng.run(0, 2)
t0 = time.perf_counter(); eng.run(2, 10); dt = (time.perf_counter() - t0) / 10
print(f"BGe reparam d={d} M={M}: {dt*1e3:.2f} ms/step")
eng.set_profiling(True); eng.reset_timers(); eng.run(12, 5)
print({k: round(v[0] / v[1] * 1e3, 1) for k, v in eng.timers().items()})
