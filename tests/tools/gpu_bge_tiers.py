"""Node scores of the device against the oracle, broken down by factorisation size n (which BGe tier handled the problem)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conftest import make_data
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from oracle.c_oracle import COracle
from oracle import prng
d = int(sys.argv[1]) if len(sys.argv) > 1 else 50
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 2, 6, 12]
co = COracle("f64")
data, _, _ = make_data(d, seed=0)
cfg = make_config(n_vars=d, n_particles=M, n_observations=100)
eng = Engine(cfg); eng.set_data(data.x); eng.init_particles(prng.PRNGKey(1))
tc = 0
for t in steps:
    eng.run(tc, t - tc)
    g = eng.get_state()
    st = dict(z=g["z"].astype(np.float64), v_z=g["v_z"].astype(np.float64), key=g["key"].copy(), baseline=g["baseline"].astype(np.float64), theta=None, v_theta=None)
    dbg = co.step(cfg, data.x, None, st, t, debug=True)
    eng.run(t, 1); tc = t + 1
    ns = eng.read("NODE_SCORES").reshape(M, d, 128).transpose(0, 2, 1)
    l = dbg["g_samples"].sum(axis=2)  # [M,S,d] parents of j = column sums over i
    n = np.minimum(l + 1, d - l)
    err = np.abs(ns - dbg["node_scores"])
    scale = np.abs(dbg["node_scores"]).max()
    print(f"t={t}  max|score|={scale:.1f}")
    for lo in range(0, 60, 4):
        sel = (n > lo) & (n <= lo + 4) & (l > 0)
        if sel.any():
            e = err[sel]
            print(f"   n in ({lo:2d},{lo+4:2d}]  count {sel.sum():7d}  nan {np.isnan(e).sum():6d}  max abs err {np.nanmax(e):.3e}  rel {np.nanmax(e)/scale:.2e}  comp frac {(l[sel] + 1 > d - l[sel]).mean():.2f}")
    sel = l == 0
    if sel.any(): print(f"   l == 0        count {sel.sum():7d}  max abs err {err[sel].max():.3e}")
eng.close()
