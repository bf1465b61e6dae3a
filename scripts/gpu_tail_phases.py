import sys, numpy as np
sys.path.insert(0, "/root/repo")
from dibs_amd import random
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd.target import make_linear_gaussian_equivalent_model
data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=50, graph_prior_str="er", n_observations=100)
cfg = make_config(n_vars=50, n_particles=128, n_observations=100)
eng = Engine(cfg); eng.set_data(data.x); eng.init_particles(random.PRNGKey(1))
eng.run(0, 5)
eng.set_profiling(True); eng.reset_timers(); eng.run(5, 20)
c = eng.counters(); t = eng.timers()
print('after 20 steps'); 
c = eng.counters(); t = eng.timers()
print("k_particle_grad phase ns/launch (A, stage, B, C):", [float(x) / 20 * 10 for x in c[1:5]])
print("k_edge_scores   phase ns/launch (load+store, barrier, MFMA, epilogue):", [float(x) / 20 * 10 for x in c[9:13]])
print({k: v[0] / v[1] * 1e3 for k, v in t.items()})

