"""Steady-state step time of the headline config (t >= T0): steps/s and per-kernel microseconds, to see what the saturated-chain skip of the
acyclicity kernel buys in the long runs BASELINE's configs are defined on (1 000 - 2 000 steps).   python scripts/gpu_steady_bench.py [T0]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dibs_amd import random
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd.target import make_linear_gaussian_equivalent_model
T0s = [int(a) for a in sys.argv[1:]] or [100, 300, 600, 1000]
data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=50, graph_prior_str="er", n_observations=100)
eng = Engine(make_config(n_vars=50, n_particles=128, n_observations=100)); eng.set_data(data.x); eng.init_particles(random.PRNGKey(1))
t = 0
for T0 in T0s:
    eng.run(t, T0 - t); t = T0
    eng.sync(); t0 = time.perf_counter(); eng.run(t, 50); dt = (time.perf_counter() - t0) / 50; t += 50
    eng.set_profiling(True); eng.reset_timers(); eng.run(t, 20); t += 20
    tm = {k: round(v[0] / 20 * 1e3, 1) for k, v in eng.timers().items()}; eng.set_profiling(False)
    print(f"t={T0}: {1 / dt:.0f} steps/s ({dt * 1e6:.1f} us/step)  serial kernels {tm}", flush=True)
