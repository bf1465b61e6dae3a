"""What one rank of an R-way sharded headline run costs per step on ONE GPU (no collective): R in-process rank engines
follow the real trajectory to t=T0, then rank 0 alone is timed (wall clock and per kernel) against frozen rows of the others.
Gives the compute+launch side of strong scaling; the all-gather (E*M*4 bytes per step) comes on top."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dibs_amd import random
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd.target import make_linear_gaussian_equivalent_model

T0 = int(sys.argv[1]) if len(sys.argv) > 1 else 300
data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=50, graph_prior_str="er", n_observations=100)
for R in (1, 2, 4, 8):
    ts = torch.cuda.Stream()
    engs = []
    for r in range(R):
        e = Engine(make_config(n_vars=50, n_particles=128, n_observations=100, rank=r, n_ranks=R), stream=ts.cuda_stream)
        e.set_data(data.x); e.init_particles(random.PRNGKey(1)); engs.append(e)
    n = engs[0].gather_elems_per_rank()
    with torch.cuda.stream(ts):
        sends = [torch.zeros(n, device="cuda") for _ in range(R)]
        recv = torch.zeros(n * R, device="cuda")
        for t in range(T0):
            for r in range(R):
                engs[r].step_local(t, sends[r].data_ptr())
            torch.cat(sends, out=recv)
            for r in range(R):
                engs[r].step_update(t, recv.data_ptr())
        torch.cuda.synchronize()
        e = engs[0]
        def go(t0, k):
            for t in range(t0, t0 + k):
                e.step_local(t, sends[0].data_ptr())
                recv[:n].copy_(sends[0])
                e.step_update(t, recv.data_ptr())
        go(T0, 20); torch.cuda.synchronize()
        t0 = time.perf_counter(); go(T0 + 20, 200); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        e.set_profiling(True); e.reset_timers(); go(T0 + 220, 50); torch.cuda.synchronize()
        tm = e.timers(); e.set_profiling(False)
    ks = {k: v[0] / 50 * 1e3 for k, v in tm.items()}
    print(f"R={R} Mloc={128 // R}: wall {dt * 1e6:7.1f} us/step   kernels sum {sum(ks.values()):7.1f} us   " +
          " ".join(f"{k}={v:.1f}" for k, v in ks.items()), flush=True)
    for e in engs:
        e.close()
