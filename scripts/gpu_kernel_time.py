"""steady-state per-kernel times (t = 300..339) of the headline config; env overrides allowed"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd import random
from dibs_amd.target import make_linear_gaussian_equivalent_model
d, M = 50, 128
data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er")
cfg = make_config(n_vars=d, n_particles=M, n_observations=100)
eng = Engine(cfg); eng.set_data(data.x); eng.init_particles(random.PRNGKey(1))
t0 = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng.run(0, t0)
eng.set_profiling(True); eng.reset_timers(); eng.run(t0, 40)
tm = eng.timers()
print(os.environ.get("TAG", ""), " ".join(f"{k}={v[0]/v[1]*1e3:.1f}" for k, v in tm.items()), f"total={sum(v[0]/v[1] for v in tm.values())*1e3:.1f}us")
eng.set_profiling(False)
import time
eng.run(t0 + 40, 10)
t_ = time.perf_counter(); eng.run(t0 + 50, 400); dt = time.perf_counter() - t_
print(f"   unprofiled: {dt / 400 * 1e6:.1f} us/step = {400 / dt:.0f} steps/s")
eng.close()
