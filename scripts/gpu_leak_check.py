"""Create / run / destroy engines of every model family repeatedly and watch free device memory (leak check)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd import random
rng = np.random.default_rng(0)
x = rng.normal(size=(50, 12)).astype(np.float32)
mask = (rng.random((50, 12)) < 0.1).astype(np.int32)
cfgs = [dict(), dict(grad_estimator_z="reparam"), dict(joint=True, likelihood="lingauss"),
        dict(joint=True, likelihood="densenn", nn_hidden=(4,)), dict(has_interventions=True)]
free0 = None
for it in range(40):
    for kw in cfgs:
        e = Engine(make_config(n_vars=12, n_particles=8, n_observations=50, n_grad_mc_samples=16, n_acyclicity_mc_samples=4, **kw))
        e.set_data(x, mask if kw.get("has_interventions") else None)
        e.init_particles(random.PRNGKey(it))
        e.run(0, 3)
        e.close()
    torch.cuda.synchronize()
    free, _ = torch.cuda.mem_get_info()
    if it == 4:
        free0 = free
    if it in (4, 20, 39):
        print(f"iteration {it}: free device memory {free / 2**20:.1f} MiB", flush=True)
assert free0 - free < 8 * 2**20, f"device memory shrank by {(free0 - free) / 2**20:.1f} MiB over 35 x 5 engine lifetimes"
print("no leak")
