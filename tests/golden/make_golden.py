"""Generates the committed golden fixtures under tests/golden/.

The reference (larslorch/dibs) cannot be imported in the build container (it needs jax / igraph, absent and
un-installable), so these vectors are produced by THIS REPO'S ORACLE, not by the reference:
  * `*_autograd`  : oracle/dibs_oracle.py (torch float64, autograd wherever the reference calls jax.grad)
  * `*_cport`     : oracle/dibs_oracle.c  (float64 closed forms), which tests pin against the autograd oracle
Run:  python tests/golden/make_golden.py      (about 3 minutes)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dibs_oracle as O, prng  # noqa: E402
from oracle.c_oracle import COracle  # noqa: E402
from dibs_amd._abi import make_config  # noqa: E402
from dibs_amd import random  # noqa: E402
from dibs_amd.target import make_linear_gaussian_equivalent_model, make_linear_gaussian_model  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def config1():
    """BASELINE.json configs[0]: MarginalDiBS + BGe, d=5, ER-1, 4 particles, 50 steps."""
    d, M, steps = 5, 4, 50
    data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=d, graph_prior_str="er",
                                                       edges_per_node=1)
    x = data.x.astype(np.float32)
    key = prng.PRNGKey(1)
    ocfg = O.Config(prior=O.GraphPrior("er", 1))
    st = O.init_state(ocfg, key, M, d)
    out = dict(x=x, g_true=data.g, key=key, z_init=st.z.numpy(), key_after_init=st.key)
    xt, mt = torch.as_tensor(x, dtype=torch.float64), torch.zeros(x.shape, dtype=torch.float64)
    zs = []
    for t in range(5):
        st = O.svgd_step(ocfg, st, xt, mt, t)
        zs.append(st.z.numpy().copy())
    out["z_autograd_steps1to5"] = np.stack(zs)
    out["key_after_5"] = st.key
    co = COracle("f64")
    cfg = make_config(n_vars=d, n_particles=M, n_observations=x.shape[0], edges_per_node=1)
    cs = co.new_state(cfg, key)
    snaps = {}
    for t in range(steps):
        co.step(cfg, x, None, cs, t)
        if t + 1 in (1, 2, 5, 10, 20, 50):
            snaps[t + 1] = cs["z"].copy()
    for k, v in snaps.items():
        out[f"z_cport_step{k}"] = v
    out["v_cport_step50"] = cs["v_z"].copy()
    out["key_after_50"] = cs["key"].copy()
    np.savez_compressed(os.path.join(HERE, "config1_marginal_bge_d5.npz"), **out)
    print("config1 done; autograd-vs-cport z diff after 5 steps:", np.abs(out["z_autograd_steps1to5"][4] - snaps[5]).max())


def joint_lingauss():
    """JointDiBS + LinearGaussian, d=5, 3 particles, S=16, Sa=4, interventions on, 3 autograd steps."""
    d, M = 5, 3
    data, _, _ = make_linear_gaussian_model(key=random.PRNGKey(2), n_vars=d, graph_prior_str="er", edges_per_node=1)
    x = data.x.astype(np.float32)
    mask = (random.uniform(random.PRNGKey(5), x.shape) < 0.1).astype(np.int32)
    key = prng.PRNGKey(3)
    ocfg = O.Config(joint=True, likelihood="lingauss", alpha_linear=0.05, grad_estimator_z="reparam",
                    prior=O.GraphPrior("er", 1), n_grad_mc_samples=16, n_acyclicity_mc_samples=4)
    st = O.init_state(ocfg, key, M, d)
    out = dict(x=x, mask=mask, key=key, z_init=st.z.numpy(),
               theta_init=np.stack([st.theta[m][0].numpy() for m in range(M)]))
    xt, mt = torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(mask, dtype=torch.float64)
    zs, ths = [], []
    for t in range(1, 4):
        st = O.svgd_step(ocfg, st, xt, mt, t)
        zs.append(st.z.numpy().copy())
        ths.append(np.stack([st.theta[m][0].numpy() for m in range(M)]))
    out["z_autograd_t1to3"] = np.stack(zs)
    out["theta_autograd_t1to3"] = np.stack(ths)
    out["key_after"] = st.key
    np.savez_compressed(os.path.join(HERE, "joint_lingauss_d5.npz"), **out)
    print("joint_lingauss done")


if __name__ == "__main__":
    config1()
    joint_lingauss()
