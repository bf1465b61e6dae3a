"""Per-step kernel times (engine HIP-event timers) and parent-set tier histogram for t = 0 .. T-1 of the headline trajectory
(the driver's bench window is t = 5..24)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd import random
from dibs_amd.target import make_linear_gaussian_equivalent_model
d, M = 50, 128
T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er")
cfg = make_config(n_vars=d, n_particles=M, n_observations=100)
eng = Engine(cfg); eng.set_data(data.x); eng.init_particles(random.PRNGKey(1))
pop = np.array([bin(i).count("1") for i in range(65536)], np.uint8)
def popcnt(a):
    a = a.astype(np.uint64)
    return sum(pop[((a >> np.uint64(s)) & np.uint64(0xFFFF)).astype(np.int64)] for s in (0, 16, 32, 48))
eng.set_profiling(True)
names = None
edges = [0, 1, 8, 12, 16, 24, 32, 200]   # l bins: 0 | 1-7 | 8-11 | 12-15 | 16-23 | 24-31 | >=32
for t in range(T):
    eng.reset_timers()
    eng.run(t, 1)
    tm = eng.timers()
    l = popcnt(eng.read("PARENT_MASKS")).astype(np.int64).reshape(-1)
    h = np.histogram(l, bins=edges)[0]
    if names is None:
        names = list(tm.keys())
        print("t    " + " ".join(f"{n:>10s}" for n in names) + "   total |  mean_l  l=0    1-7   8-11  12-15  16-23  24-31   >=32   sum(n^3)/6 MFMA")
    fm = np.sum((l[l > 0] + 1.0) ** 3) / 6 / 1e6
    print(f"{t:3d}  " + " ".join(f"{tm[n][0] / tm[n][1] * 1e3:10.1f}" for n in names) + f" {sum(v[0] / v[1] for v in tm.values()) * 1e3:7.1f} | "
          f"{l.mean():6.2f} " + " ".join(f"{x / l.size:6.3f}" for x in h) + f"  {fm:8.1f}")
eng.close()
