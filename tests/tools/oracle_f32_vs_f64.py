"""How far do the oracle's own float32 and float64 builds drift apart on a free-running trajectory?  (CPU only.)
Same config / seeds as tests/tools/gpu_eshd_compare.py: the GPU-vs-oracle(f64) gap after 1000 steps is of the same size as the
f32-vs-f64 gap of the oracle itself, i.e. it is the chaotic amplification of fp32 rounding (softmax over samples is close to
an argmax; near-ties flip), not a property of the HIP kernels."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dibs_amd import random
from dibs_amd.inference import MarginalDiBS
from dibs_amd.metrics import expected_shd, expected_edges
from dibs_amd.target import make_linear_gaussian_equivalent_model
from oracle.c_oracle import COracle

d, M = 20, 32
data, gm, lm = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er")
dibs = MarginalDiBS(x=data.x, graph_model=gm, likelihood_model=lm)
cfg = dibs._make_config(M, d)
for steps in (200, 1000):
    res = {}
    for prec in ("f64", "f32"):
        co = COracle(prec)
        st = co.new_state(cfg, random.PRNGKey(1))
        co.run(cfg, data.x, None, st, 0, steps, n_threads=8)
        g = dibs.particle_to_g_lim(st["z"])
        dist = dibs.get_empirical(g)
        res[prec] = (st["z"].astype(np.float64), g, expected_shd(dist=dist, g=data.g), expected_edges(dist=dist))
    z64, g64, s64, e64 = res["f64"]; z32, g32, s32, e32 = res["f32"]
    print(f"steps={steps}: oracle f64 E-SHD {s64:.3f} E-edges {e64:.3f} | oracle f32 E-SHD {s32:.3f} E-edges {e32:.3f} | "
          f"identical final graphs {100 * (g64 == g32).all(axis=(1, 2)).mean():.1f} %  max|dz|/max|z| {np.abs(z64 - z32).max() / np.abs(z64).max():.3e}")
