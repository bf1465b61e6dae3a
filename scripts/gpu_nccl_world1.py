"""Exercise the N>1 code path of bench.py / run_sharded on a 1-GPU box: a 1-rank RCCL group, engine + collective on a
dedicated stream, the real all_gather_into_tensor call each step; result must equal Engine.run bit for bit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from dibs_amd import random
from dibs_amd._abi import make_config
from dibs_amd.engine import Engine
from dibs_amd.target import make_linear_gaussian_equivalent_model

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
D_, M_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50, 128)   # (small sizes: the host-side cost per step of either protocol)
data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=D_, graph_prior_str="er", n_observations=100)
cfg = make_config(n_vars=D_, n_particles=M_, n_observations=100)
ts = torch.cuda.Stream()
a = Engine(cfg, stream=ts.cuda_stream); b = Engine(cfg)
for e in (a, b):
    e.set_data(data.x); e.init_particles(random.PRNGKey(1))
n = a.gather_elems_per_rank()
K = 200
with torch.cuda.stream(ts):
    send = torch.zeros(n, device="cuda"); recv = torch.zeros(n, device="cuda")
    def steps(t0, k):
        for t in range(t0, t0 + k):
            a.step_local(t, send.data_ptr())
            dist.all_gather_into_tensor(recv, send)
            a.step_update(t, recv.data_ptr())
    steps(0, 20); torch.cuda.synchronize()
    t0 = time.perf_counter(); steps(20, K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
b.run(0, 20); b.sync(); t0 = time.perf_counter(); b.run(20, K); b.sync(); dt1 = time.perf_counter() - t0
# the overlapped protocol through the same real RCCL calls (values on a side stream, gradients between the phases)
from dibs_amd.distributed import OverlapBuffers, run_sharded_overlapped
c = Engine(cfg, stream=ts.cuda_stream); c.set_data(data.x); c.init_particles(random.PRNGKey(1))
with torch.cuda.stream(ts):
    buf = OverlapBuffers(c, 1, torch.device("cuda", 0), torch.float32)
    run_sharded_overlapped(c, 0, 20, buf, always_collective=True); torch.cuda.synchronize()
    t0 = time.perf_counter(); run_sharded_overlapped(c, 20, K, buf, always_collective=True); torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
zc = c.get_state()["z"]
print(f"overlapped protocol, RCCL(world 1): {K/dt2:.0f} steps/s; bit-identical: {np.array_equal(zc, b.get_state()['z'])}")
assert np.array_equal(zc, b.get_state()["z"])
# the step loop inside the engine (dibs_engine_run_sharded): RCCL bound by libdibs_hip.so itself, no Python between the steps
for ov in (False, True):
    n_ = Engine(cfg); n_.set_data(data.x); n_.comm_init(n_.comm_unique_ids(2 if ov else 1)); n_.init_particles(random.PRNGKey(1))
    n_.run_sharded(0, 20, ov); t0 = time.perf_counter(); n_.run_sharded(20, K, ov); dtn = time.perf_counter() - t0
    same = np.array_equal(n_.get_state()["z"], b.get_state()["z"])
    print(f"in-engine loop ({'overlapped' if ov else 'one all-gather per step'}), RCCL(world 1): {K/dtn:.0f} steps/s; bit-identical: {same}")
    assert same
    n_.close()
za, zb = a.get_state()["z"], b.get_state()["z"]
print(f"per-step python+RCCL(world 1) path: {K/dt:.0f} steps/s; Engine.run: {K/dt1:.0f} steps/s; bit-identical: {np.array_equal(za, zb)}")
assert np.array_equal(za, zb)
dist.destroy_process_group()
