"""How accurate are the device's BGe node scores compared with what float32 arithmetic gives the oracle itself?
Prints max / rms / 99.9 % quantile of |node score - f64 oracle| for the device and for the oracle's f32 build, same state."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conftest import make_data
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from oracle.c_oracle import COracle
from oracle import prng
d = 50
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 5, 20]
co, co32 = COracle("f64"), COracle("f32")
data, _, _ = make_data(d, seed=0)
cfg = make_config(n_vars=d, n_particles=M, n_observations=100)
eng = Engine(cfg); eng.set_data(data.x); eng.init_particles(prng.PRNGKey(1))
tc = 0
def stats(e): return f"max {e.max():.2e} rms {np.sqrt((e ** 2).mean()):.2e} q99.9 {np.quantile(e, 0.999):.2e}"
for t in steps:
    eng.run(tc, t - tc)
    g = eng.get_state()
    mk = lambda r: dict(z=g["z"].astype(r), v_z=g["v_z"].astype(r), key=g["key"].copy(), baseline=g["baseline"].astype(r), theta=None, v_theta=None)
    dbg = co.step(cfg, data.x, None, mk(np.float64), t, debug=True)
    dbg32 = co32.step(cfg, data.x, None, mk(np.float32), t, debug=True)
    eng.run(t, 1); tc = t + 1
    ns = eng.read("NODE_SCORES").reshape(M, d, 128).transpose(0, 2, 1)
    e_dev, e_32 = np.abs(ns - dbg["node_scores"]).ravel(), np.abs(dbg32["node_scores"].astype(np.float64) - dbg["node_scores"]).ravel()
    print(f"t={t} M={M}  |score| max {np.abs(dbg['node_scores']).max():.1f}")
    print("   node scores   device:", stats(e_dev), " | oracle f32:", stats(e_32))
    lp = eng.read("LOGPROBS_Z").reshape(M, 128)
    print("   log p(D|G_s)  device:", stats(np.abs(lp - dbg["logprobs_z"]).ravel()), " | oracle f32:", stats(np.abs(dbg32["logprobs_z"].astype(np.float64) - dbg["logprobs_z"]).ravel()))
    w = eng.read("W_LIK").reshape(M, d, d)
    print("   W_lik         device:", stats(np.abs(w - dbg["w_lik"]).ravel()), " | oracle f32:", stats(np.abs(dbg32["w_lik"].astype(np.float64) - dbg["w_lik"]).ravel()))
eng.close()
