"""Headline window (t = 5..24) in a few seconds: steps/s (median of 5) and per-kernel microseconds, serial and concurrent.
    python scripts/gpu_quick_bench.py [n_particles] [d]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dibs_amd import random
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd.target import make_linear_gaussian_equivalent_model
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
d = int(sys.argv[2]) if len(sys.argv) > 2 else 50
W, K = 5, 20
data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er", n_observations=100)
cfg = make_config(n_vars=d, n_particles=M, n_observations=100)
eng = Engine(cfg); eng.set_data(data.x); eng.init_particles(random.PRNGKey(1))
eng.run(0, W)
snap = {k: v for k, v in eng.get_state().items() if v is not None}
ts = []
for rep in range(6):
    eng.set_state(**snap)
    t0 = time.perf_counter(); eng.run(W, K); ts.append(time.perf_counter() - t0)
el = float(np.median(ts[1:]))
print(f"M={M} d={d}: {K / el:.0f} steps/s  ({1e6 * el / K:.1f} us/step; reps {[round(1e6 * t / K, 1) for t in ts]})")
for mode in (1, 2):
    eng.set_state(**snap); eng.set_profiling(mode); eng.reset_timers(); eng.run(W, K)
    print("serial    " if mode == 1 else "concurrent", {k: round(v[0] / K * 1e3, 1) for k, v in eng.timers().items()})
    eng.set_profiling(False)
