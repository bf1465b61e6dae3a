"""Free-running drift of the joint models against the f64 C oracle (d=20, 16 particles) after 50 / 100 / 200 steps."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dibs_amd import random
from dibs_amd.inference import JointDiBS
from dibs_amd.target import make_linear_gaussian_model, make_nonlinear_gaussian_model
from oracle.c_oracle import COracle
co = COracle("f64")
for name, f in (("lingauss", make_linear_gaussian_model), ("densenn", make_nonlinear_gaussian_model)):
    for steps in (50, 100, 200):
        d, M = 20, 16
        data, gm, lm = f(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er")
        dibs = JointDiBS(x=data.x, graph_model=gm, likelihood_model=lm)
        g, th = dibs.sample(key=random.PRNGKey(1), n_particles=M, steps=steps)
        zg = dibs.last_state["z"]
        cfg = dibs._make_config(M, d)
        st = co.new_state(cfg, random.PRNGKey(1))
        co.run(cfg, data.x, None, st, 0, steps, n_threads=16)
        go = dibs.particle_to_g_lim(st["z"])
        print(name, steps, "z relerr", np.abs(zg - st["z"]).max() / np.abs(st["z"]).max(), "same graphs", (g == go).all(axis=(1, 2)).mean(), flush=True)
