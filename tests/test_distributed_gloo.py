"""world_size-2 gloo test (CPU) of the particle-sharded step protocol in dibs_amd/distributed.py: phase A on the local
shard, ONE all-gather of the packed rows, phase B on the local shard.  The compute behind the protocol is the oracle's
C port here (there is no GPU in this container); the HIP engine exposes the same three calls.  The sharded run must be
bit-identical to the single-rank run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import make_data
from dibs_amd._abi import make_config
from dibs_amd.distributed import make_buffers, run_sharded
from oracle import prng


class OracleShardEngine:
    """adapter: the oracle's phase A / phase B behind the engine protocol used by run_sharded()"""

    def __init__(self, co, cfg, x, mask, key):
        self.co, self.cfg, self.x, self.mask = co, cfg, x, mask
        self.st = co.new_local_state(cfg, key)
        self.E = co.pack_stride(cfg)
        self.Ml = cfg.n_particles // cfg.n_ranks

    def gather_elems_per_rank(self):
        return self.Ml * self.E

    def _view(self, ptr, n):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_double * n).from_address(ptr))

    def step_local(self, t, send_ptr):
        self.co.step_local(self.cfg, self.x, self.mask, self.st, t, self._view(send_ptr, self.Ml * self.E), n_threads=2)

    def step_update(self, t, recv_ptr):
        self.co.step_update(self.cfg, self._view(recv_ptr, self.cfg.n_particles * self.E), self.st, n_threads=2)


def _cfg(joint, rank, n_ranks, d, M):
    kw = dict(joint=True, likelihood="lingauss") if joint else {}
    return make_config(n_vars=d, n_particles=M, n_observations=100, edges_per_node=1, n_grad_mc_samples=16,
                       n_acyclicity_mc_samples=4, rank=rank, n_ranks=n_ranks, **kw)


def _worker(rank, world, port, joint, d, M, steps, x, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.c_oracle import COracle
        co = COracle("f64")
        eng = OracleShardEngine(co, _cfg(joint, rank, world, d, M), x, None, prng.PRNGKey(4))
        send, recv = make_buffers(eng, world, torch.device("cpu"), torch.float64)
        run_sharded(eng, 0, steps, send, recv)
        z = torch.from_numpy(eng.st["z"])
        zs = [torch.empty_like(z) for _ in range(world)]
        dist.all_gather(zs, z)
        if rank == 0:
            np.save(os.path.join(out_dir, "z.npy"), torch.cat(zs).numpy())
            np.save(os.path.join(out_dir, "key.npy"), eng.st["key"])
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("joint", [False, True])
def test_two_rank_gloo_matches_single_rank(tmp_path, c_oracle64, joint):
    d, M, steps = 6, 8, 3
    data, _, _ = make_data(d, seed=3, joint=joint)
    x = np.ascontiguousarray(data.x, np.float64)
    mp.spawn(_worker, args=(2, _free_port(), joint, d, M, steps, x, str(tmp_path)), nprocs=2, join=True)
    z2 = np.load(tmp_path / "z.npy")
    key2 = np.load(tmp_path / "key.npy")
    cfg = _cfg(joint, 0, 1, d, M)
    st = c_oracle64.new_state(cfg, prng.PRNGKey(4))
    c_oracle64.run(cfg, x, None, st, 0, steps)
    assert np.array_equal(z2, st["z"]), "sharded run must be bit-identical to the single-rank run"
    assert (key2 == st["key"]).all()


def test_single_process_protocol_equals_fused_step(c_oracle64):
    """run_sharded with one rank (memcpy instead of a collective) == orc_run"""
    d, M = 5, 4
    data, _, _ = make_data(d, seed=1)
    x = np.ascontiguousarray(data.x, np.float64)
    cfg = _cfg(False, 0, 1, d, M)
    eng = OracleShardEngine(c_oracle64, cfg, x, None, prng.PRNGKey(2))
    send, recv = make_buffers(eng, 1, torch.device("cpu"), torch.float64)
    run_sharded(eng, 0, 4, send, recv)
    st = c_oracle64.new_state(cfg, prng.PRNGKey(2))
    c_oracle64.run(cfg, x, None, st, 0, 4)
    assert np.array_equal(eng.st["z"], st["z"])
