"""Which stage buffer differs between a single-rank engine and R rank engines (in-process harness) at a given size?"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_data
from dibs_amd import random as prng
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine

d, M, S, Sa, R, steps = [int(a) for a in sys.argv[1:7]]
data, _, _ = make_data(d, seed=0)
kw = dict(n_vars=d, n_particles=M, n_observations=100, n_grad_mc_samples=S, n_acyclicity_mc_samples=Sa)
ref = Engine(make_config(**kw)); ref.set_data(data.x); ref.init_particles(prng.PRNGKey(1))
ts = torch.cuda.Stream()
engs = []
for r in range(R):
    e = Engine(make_config(rank=r, n_ranks=R, **kw), stream=ts.cuda_stream); e.set_data(data.x); e.init_particles(prng.PRNGKey(1)); engs.append(e)
n = engs[0].gather_elems_per_rank()
names = ["SCORES", "W_ACYC", "NODE_SCORES", "LOGPROBS_Z", "W_LIK", "GRAD_Z", "KXX", "PHI_Z", "Z"]
with torch.cuda.stream(ts):
    sends = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(R)]
    recv = torch.zeros(n * R, dtype=torch.float32, device="cuda")
for t in range(steps):
    ref.run(t, 1)
    with torch.cuda.stream(ts):
        for r in range(R): engs[r].step_local(t, sends[r].data_ptr())
        torch.cat(sends, out=recv)
        for r in range(R): engs[r].step_update(t, recv.data_ptr())
    torch.cuda.synchronize()
    for nm in names:
        a = ref.read(nm)
        try:
            b = np.concatenate([e.read(nm) for e in engs])
        except Exception as ex:
            print(t, nm, "unreadable", ex); continue
        if a.shape != b.shape:
            print(t, nm, "shape", a.shape, b.shape); continue
        neq = int((a != b).sum())
        print(f"t={t} {nm:12s} differing={neq} of {a.size} maxabs={float(np.abs(a.astype(np.float64)-b).max()):.3e}")
