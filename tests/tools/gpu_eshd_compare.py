"""Posterior quality parity: E-SHD / expected #edges of the GPU engine vs the oracle (C port, f64) on the same seeded inputs.
BASELINE.json configs[1]: MarginalDiBS + BGe, d=20, 32 particles, 1000 steps (plus a JointDiBS + LinearGaussian run)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dibs_amd import random
from dibs_amd._abi import make_config
from dibs_amd.inference import MarginalDiBS, JointDiBS
from dibs_amd.metrics import expected_shd, expected_edges, threshold_metrics
from dibs_amd.target import make_linear_gaussian_equivalent_model, make_linear_gaussian_model
from oracle.c_oracle import COracle

def run(joint, d, M, steps, seed=0):
    f = make_linear_gaussian_model if joint else make_linear_gaussian_equivalent_model
    data, gm, lm = f(key=random.PRNGKey(seed), n_vars=d, graph_prior_str="er")
    cls = JointDiBS if joint else MarginalDiBS
    dibs = cls(x=data.x, graph_model=gm, likelihood_model=lm)
    t0 = time.time()
    out = dibs.sample(key=random.PRNGKey(1), n_particles=M, steps=steps)
    tg = time.time() - t0
    g = out[0] if joint else out
    dist = dibs.get_empirical(g, out[1]) if joint else dibs.get_empirical(g)
    zg = dibs.last_state["z"]
    # oracle
    cfg = dibs._make_config(M, d)
    co = COracle("f64")
    st = co.new_state(cfg, random.PRNGKey(1))
    t0 = time.time()
    co.run(cfg, data.x, None, st, 0, steps, n_threads=min(os.cpu_count(), 16))  # more threads than particles only adds OpenMP overhead
    to = time.time() - t0
    go = dibs.particle_to_g_lim(st["z"])
    disto = dibs.get_empirical(go, st["theta"].reshape(M, d, d)) if joint else dibs.get_empirical(go)
    name = ("JointDiBS+LinearGaussian" if joint else "MarginalDiBS+BGe") + f" d={d} M={M} steps={steps}"
    print(f"{name}: true edges {int(data.g.sum())}")
    print(f"  GPU    : E-SHD {expected_shd(dist=dist, g=data.g):8.3f}  E-edges {expected_edges(dist=dist):7.3f}  AUROC {threshold_metrics(dist=dist, g=data.g)['roc_auc']:.4f}  ({tg:.2f} s, {steps/tg:.0f} steps/s incl. setup)")
    print(f"  oracle : E-SHD {expected_shd(dist=disto, g=data.g):8.3f}  E-edges {expected_edges(dist=disto):7.3f}  AUROC {threshold_metrics(dist=disto, g=data.g)['roc_auc']:.4f}  ({to:.2f} s, {steps/to:.1f} steps/s, {min(os.cpu_count(), 16)} threads)")
    same = (g == go).all(axis=(1, 2)).mean()
    print(f"  particles with identical final graph: {same*100:.1f} %   max |z_gpu - z_oracle| / max|z| = {np.abs(zg - st['z']).max() / np.abs(st['z']).max():.3e}")

if __name__ == "__main__":
    run(False, 20, 32, 1000)
    run(False, 20, 32, 200)
    run(True, 20, 32, 500)
    run(False, 50, 128, 300)
