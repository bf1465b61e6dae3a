import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from oracle.c_oracle import COracle
from oracle import prng
from dibs_amd import random
from dibs_amd.target import make_linear_gaussian_equivalent_model
co = COracle("f64")
import sys
DS = [int(a) for a in sys.argv[1:]] or [48, 60, 64, 65, 70, 80, 100]
for d in DS:
    data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er")
    cfg = make_config(n_vars=d, n_particles=2, n_observations=100, n_grad_mc_samples=8, n_acyclicity_mc_samples=8)
    st = co.new_state(cfg, prng.PRNGKey(1))
    eng = Engine(cfg); eng.set_data(data.x)
    st["z"] = st["z"].astype(np.float32).astype(np.float64)
    eng.set_state(z=st["z"], v_z=st["v_z"], key=st["key"], baseline=st["baseline"])
    dbg = co.step(cfg, data.x.astype(np.float64), None, st, 1, debug=True)
    eng.run(1, 1)
    wa = eng.read("W_ACYC").reshape(2, d, d)
    ref = dbg["w_acyc"]
    bad = ~np.isfinite(wa)
    err = np.abs(np.where(bad, 0, wa) - ref).max() / np.abs(ref).max()
    print(d, "nan count", int(bad.sum()), "rel err (finite part)", err, "ref max", np.abs(ref).max())
    if bad.any():
        ii = np.argwhere(bad)
        print("   first bad:", ii[:5].tolist(), "rows with bad:", sorted(set(ii[:,1].tolist()))[:20], "cols:", sorted(set(ii[:,2].tolist()))[:20])
    eng.close()
