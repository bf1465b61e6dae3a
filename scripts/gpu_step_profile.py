"""Per-kernel time as a function of the step index t along the headline trajectory."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd import random
from dibs_amd.target import make_linear_gaussian_equivalent_model, make_linear_gaussian_model
d, M = 50, 128
JOINT = len(sys.argv) > 1 and sys.argv[1] == "joint"
if JOINT:
    data, _, _ = make_linear_gaussian_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er")
    cfg = make_config(n_vars=d, n_particles=M, n_observations=100, joint=True, likelihood="lingauss")
else:
    data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er")
    cfg = make_config(n_vars=d, n_particles=M, n_observations=100)
eng = Engine(cfg); eng.set_data(data.x); eng.init_particles(random.PRNGKey(1))
eng.set_profiling(True)
t = 0
names = None
for tt in [0, 1, 3, 5, 8, 12, 20, 30, 50, 100, 200, 300, 500, 1000]:
    if tt > t:
        eng.run(t, tt - t)
    eng.reset_timers()
    eng.run(tt, 4); t = tt + 4
    tm = eng.timers()
    if names is None:
        names = list(tm.keys()); print("t      " + " ".join(f"{n:>11s}" for n in names) + "      total_us")
    print(f"{tt:5d}  " + " ".join(f"{tm[n][0] / tm[n][1] * 1e3:11.1f}" for n in names) + f"  {sum(v[0] / v[1] for v in tm.values()) * 1e3:10.1f}")
eng.close()
