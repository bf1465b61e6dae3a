"""What one rank of an R-way sharded headline run costs per step on ONE GPU (no collective): R in-process rank engines
follow the real trajectory to t=T0, then rank 0 alone is timed (wall clock and per kernel) against frozen rows of the others.
Gives the compute+launch side of strong scaling; the all-gather (E*M*4 bytes per step) comes on top."""
import os, sys, time
os.environ.setdefault("DIBS_FLAGS_MULTI", "1")  # R rank engines in ONE process: use the in-kernel flags as a rank alone in its process would
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
                torch.mul(sends[0], 1, out=recv[:n])
                e.step_update(t, recv.data_ptr())
        go(T0, 20); torch.cuda.synchronize()
        t0 = time.perf_counter(); go(T0 + 20, 200); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        e.set_profiling(True); e.reset_timers(); go(T0 + 220, 50); torch.cuda.synchronize()
        tm = e.timers(); e.set_profiling(False)
        # the same rank under the overlapped protocol: values of all particles in plane 0 ahead of phase A (kernel matrix on the second
        # stream), only gradient rows between the phases
        nv = e.plane_elems_per_rank()
        planes = torch.zeros(2 * nv * R, device="cuda"); gs = torch.zeros(nv, device="cuda"); vs = torch.zeros(nv, device="cuda")
        ready = torch.cuda.Event()
        for r in range(R):
            engs[r].export_values(vs.data_ptr()); torch.mul(vs, 1, out=planes[r * nv:(r + 1) * nv])
        for o in engs[1:]:   # (their frozen rows stay in `planes`; eight engines' streams oversubscribe the hardware queues of one process)
            o.close()
        side, exported = torch.cuda.Stream(), torch.cuda.Event()
        e.kmat_values(planes.data_ptr(), ts.cuda_stream); ready.record(ts)
        def go2(t0, k):
            for t in range(t0, t0 + k):
                e.step_local_grads(t, gs.data_ptr())
                torch.mul(gs, 1, out=planes[nv * R:nv * R + nv])   # (stands in for the gradient all-gather; a copy KERNEL: Tensor.copy_ is hipMemcpyAsync)
                ts.wait_event(ready)
                e.step_update_planes(t, planes.data_ptr(), vs.data_ptr())
                exported.record(ts)
                with torch.cuda.stream(side):                  # (stands in for the all-gather of the values on the side stream)
                    side.wait_event(exported); torch.mul(vs, 1, out=planes[:nv]); e.kmat_values(planes.data_ptr(), side.cuda_stream); ready.record(side)
        go2(T0 + 270, 20); torch.cuda.synchronize()
        t0 = time.perf_counter(); go2(T0 + 290, 200); torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / 200
    # the same rank in the IN-ENGINE loop (dibs_engine_run_sharded, no Python between the steps) with a loopback communicator: the
    # collectives are skipped, the rows of the other ranks stay frozen -- the compute + launch side of a rank's step in the production loop
    st0 = {k: v for k, v in e.get_state().items() if v is not None}
    nat = {}
    for ov in (False, True):
        n_ = Engine(make_config(n_vars=50, n_particles=128, n_observations=100, rank=0, n_ranks=R))
        n_.set_data(data.x); n_.set_state(**st0); n_.comm_init(None, 2)
        n_.run_sharded(T0 + 500, 20, ov); t0 = time.perf_counter(); n_.run_sharded(T0 + 520, 200, ov); nat[ov] = (time.perf_counter() - t0) / 200
        n_.close()
    ks = {k: v[0] / 50 * 1e3 for k, v in tm.items()}
    print(f"R={R} Mloc={128 // R}: in-engine loop {nat[False] * 1e6:6.1f} us/step packed, {nat[True] * 1e6:6.1f} overlapped | Python-driven: wall {dt * 1e6:7.1f} us/step (overlapped protocol {dt2 * 1e6:7.1f})   kernels sum {sum(ks.values()):7.1f} us   " +
          " ".join(f"{k}={v:.1f}" for k, v in ks.items()), flush=True)
    engs[0].close()
