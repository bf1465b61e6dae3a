"""How large are the sampled parent sets along the headline trajectory? (guides the BGe kernel tiers)"""
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
pop = np.array([bin(i).count("1") for i in range(65536)], np.uint8)
def popcnt(a):
    a = a.astype(np.uint64)
    return sum(pop[((a >> np.uint64(s)) & np.uint64(0xFFFF)).astype(np.int64)] for s in (0, 16, 32, 48))
t = 0
for tt in [0, 1, 2, 3, 5, 8, 12, 16, 20, 30, 40, 60, 80, 100, 150, 200, 300, 500, 1000, 2000]:
    eng.run(t, tt - t + 1); t = tt + 1
    mk = eng.read("PARENT_MASKS").reshape(M, d, 128).transpose(0, 2, 1)
    l = popcnt(mk).astype(np.int64)  # [M,S,d]
    n = l + 1
    # per (m,j) spread across samples
    spread = (l.max(1) - l.min(1))
    hist = np.bincount(np.minimum(n.reshape(-1), 40), minlength=41)
    frac = lambda a, b: hist[a:b].sum() / hist.sum()
    print(f"t={tt:5d} mean l={l.mean():6.2f} max={l.max():3d} p99={np.percentile(l,99):5.1f} n<=4:{frac(0,5):.3f} n<=8:{frac(0,9):.3f} n<=16:{frac(0,17):.3f} n<=32:{frac(0,33):.3f} "
          f"sum n^3/3={np.sum(n.astype(np.float64)**3)/3/1e6:9.1f} MFLOP  spread(mean/max)={spread.mean():.1f}/{spread.max()}")
eng.close()
