"""Step time and per-kernel timers of the global-memory paths (113 .. 256 variables): MarginalDiBS + BGe at d = 128 / 160 / 200 and
JointDiBS + LinearGaussian at d = 128, default sample counts, a few steps from the initial particles."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dibs_amd import random
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd.target import make_linear_gaussian_equivalent_model, make_linear_gaussian_model

for name, d, M, joint in (("bge", 128, 32, False), ("bge", 160, 32, False), ("bge", 200, 16, False), ("lingauss", 128, 16, True)):
    N = 2 * d
    f = make_linear_gaussian_model if joint else make_linear_gaussian_equivalent_model
    data, _, _ = f(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er", n_observations=N)
    kw = dict(joint=True, likelihood="lingauss") if joint else {}
    eng = Engine(make_config(n_vars=d, n_particles=M, n_observations=N, **kw))
    eng.set_data(data.x)
    eng.init_particles(random.PRNGKey(1))
    eng.run(0, 2)
    t0 = time.perf_counter(); eng.run(2, 5); dt = (time.perf_counter() - t0) / 5
    eng.set_profiling(True); eng.reset_timers(); eng.run(7, 3)
    tm = {k: v[0] / 3 for k, v in eng.timers().items()}
    eng.close()
    print(f"{name} d={d} M={M} S=128 Sa=32 N={N}: {dt * 1e3:8.2f} ms/step   " + " ".join(f"{k}={v:.2f}ms" for k, v in tm.items()), flush=True)
