"""Fraction of Bernoulli thresholds that are deterministic (thr == 0: p == 0.0f, thr == 2^23: p == 1.0f) or minimal (thr == 1)
along the headline trajectory, and how many scores fall in the band where float32 sigmoid overflows to 0 but the double one does not."""
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
t = 0
off = ~np.eye(d, dtype=bool)
for tt in [1, 5, 20, 50, 100, 150, 200, 300, 500, 1000]:
    eng.run(t, tt - t + 1); t = tt + 1
    sc = (eng.read("SCORES").reshape(M, d, d)[:, off] * np.float32(tt)).astype(np.float64)  # alpha = t (alpha_linear = 1)
    with np.errstate(over="ignore"):
        pf = (1.0 / (1.0 + np.exp(-sc))).astype(np.float32)
    thr = np.ceil(pf.astype(np.float64) * 2.0 ** 23).astype(np.int64)
    print(f"t={tt:5d} thr==0 {np.mean(thr == 0):.3f}  thr==1 {np.mean(thr == 1):.3f}  thr==2^23 {np.mean(thr == 1 << 23):.3f}  "
          f"x<-88.72 {np.mean(sc < -88.7228):.3f}  x<-103.97 {np.mean(sc < -103.97):.3f}  x>16.64 {np.mean(sc > 16.64):.3f}")
eng.close()
