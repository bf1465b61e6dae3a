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
from dibs_amd.distributed import OverlapBuffers, make_buffers, run_sharded, run_sharded_overlapped
from oracle import prng


class OracleShardEngine:
    """adapter: the oracle's phase A / phase B behind the engine protocol used by run_sharded()"""

    def __init__(self, co, cfg, x, mask, key):
        self.co, self.cfg, self.x, self.mask = co, cfg, x, mask
        self.st = co.new_local_state(cfg, key)
        self.E = co.pack_stride(cfg)
        self.Ml = cfg.n_particles // cfg.n_ranks
        self.state_gen = 0        # as Engine.state_gen: bumped when the particles are replaced from outside
        self._slab = None         # the externally computed kernel slab (kmat_values): checksum of the values it was computed from
        self.slabs_consumed = 0

    def set_state(self, **kw):
        for k, v in kw.items():
            self.st[k][...] = v
        self.state_gen += 1
        self._slab = None         # (dibs_engine_set_state clears kmat_ext)

    def gather_elems_per_rank(self):
        return self.Ml * self.E

    def _view(self, ptr, n):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_double * n).from_address(ptr))

    def step_local(self, t, send_ptr):
        self.co.step_local(self.cfg, self.x, self.mask, self.st, t, self._view(send_ptr, self.Ml * self.E), n_threads=2)

    def step_update(self, t, recv_ptr):
        if self._slab is not None:   # (the engine consumes a pending external slab here too: it must belong to the current particles)
            own = self._slab.reshape(self.cfg.n_particles, self.Ev)[self.cfg.rank * self.Ml:(self.cfg.rank + 1) * self.Ml]
            assert np.array_equal(own[:, :self.D], self.st["z"].reshape(self.Ml, self.D)), "stale kernel slab consumed by a packed step"
            self._slab = None
        self.co.step_update(self.cfg, self._view(recv_ptr, self.cfg.n_particles * self.E), self.st, n_threads=2)
        self.state_gen += 1       # as Engine.step_update: the particles moved without the overlapped exchange

    # ---- overlapped protocol (values and gradients travel separately; include/dibs_hip.h) on top of the oracle's packed-row phases ----
    @property
    def D(self):
        return self.cfg.n_vars * self.cfg.n_dim * 2

    @property
    def P(self):
        return self.co.theta_size(self.cfg)

    @property
    def Ev(self):
        return (self.D + self.P + 3) & ~3

    def plane_elems_per_rank(self):
        return self.Ml * self.Ev

    def export_values(self, vals_send_ptr):
        v = self._view(vals_send_ptr, self.Ml * self.Ev).reshape(self.Ml, self.Ev)
        v[:, :self.D] = self.st["z"].reshape(self.Ml, self.D)
        if self.P:
            v[:, self.D:self.D + self.P] = self.st["theta"].reshape(self.Ml, self.P)

    def kmat_values(self, vals_all_ptr, stream):
        # The oracle's phase B computes the kernel matrix itself; what the adapter checks is the PROTOCOL of the externally computed slab
        # (the engine's kmat_ext flag): computed once per step from the gathered values, consumed exactly once, by the phase B whose
        # plane 0 holds those same values -- never a slab of replaced particles.
        assert self._slab is None, "kernel slab computed twice without a phase B in between"
        self._slab = self._view(vals_all_ptr, self.cfg.n_particles * self.Ev).copy()

    def step_local_grads(self, t, grads_send_ptr):
        D, P = self.D, self.P
        pack = np.zeros((self.Ml, self.E))
        self.co.step_local(self.cfg, self.x, self.mask, self.st, t, pack.reshape(-1), n_threads=2)
        g = self._view(grads_send_ptr, self.Ml * self.Ev).reshape(self.Ml, self.Ev)
        g[:, :D] = pack[:, D:2 * D]
        if P:
            g[:, D:D + P] = pack[:, 2 * D + P:2 * D + 2 * P]

    def step_update_planes(self, t, planes_ptr, vals_send_ptr=None):
        D, P, M = self.D, self.P, self.cfg.n_particles
        pl = self._view(planes_ptr, 2 * M * self.Ev).reshape(2, M, self.Ev)
        assert self._slab is not None, "phase B of the overlapped protocol without a kernel slab for this step"
        assert np.array_equal(self._slab, pl[0].reshape(-1)), "kernel slab was computed from other values than phase B reads"
        own = pl[0, self.cfg.rank * self.Ml:(self.cfg.rank + 1) * self.Ml]
        assert np.array_equal(own[:, :D], self.st["z"].reshape(self.Ml, D)), "plane 0 does not hold the engine's current particles"
        self._slab = None
        self.slabs_consumed += 1
        pack = np.zeros((M, self.E))
        pack[:, :D], pack[:, D:2 * D] = pl[0, :, :D], pl[1, :, :D]
        if P:
            pack[:, 2 * D:2 * D + P], pack[:, 2 * D + P:2 * D + 2 * P] = pl[0, :, D:D + P], pl[1, :, D:D + P]
        self.co.step_update(self.cfg, pack.reshape(-1), self.st, n_threads=2)
        if vals_send_ptr:
            self.export_values(vals_send_ptr)


def _cfg(joint, rank, n_ranks, d, M):
    kw = dict(joint=True, likelihood="lingauss") if joint else {}
    return make_config(n_vars=d, n_particles=M, n_observations=100, edges_per_node=1, n_grad_mc_samples=16,
                       n_acyclicity_mc_samples=4, rank=rank, n_ranks=n_ranks, **kw)


def _worker(rank, world, port, joint, d, M, steps, x, out_dir, overlapped=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.c_oracle import COracle
        co = COracle("f64")
        eng = OracleShardEngine(co, _cfg(joint, rank, world, d, M), x, None, prng.PRNGKey(4))
        if overlapped:   # two chunks: the values gathered after the last step of a chunk serve the first step of the next one
            buf = OverlapBuffers(eng, world, torch.device("cpu"), torch.float64)
            run_sharded_overlapped(eng, 0, steps - 1, buf)
            run_sharded_overlapped(eng, steps - 1, 1, buf)
        else:
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


@pytest.mark.parametrize("joint,overlapped", [(False, False), (True, False), (False, True), (True, True)])
def test_two_rank_gloo_matches_single_rank(tmp_path, c_oracle64, joint, overlapped):
    """both exchange protocols of dibs_amd/distributed.py: one all-gather of packed rows per step, and values / gradients gathered
    separately (the values after the optimizer step, the gradients between the phases)"""
    d, M, steps = 6, 8, 3
    data, _, _ = make_data(d, seed=3, joint=joint)
    x = np.ascontiguousarray(data.x, np.float64)
    mp.spawn(_worker, args=(2, _free_port(), joint, d, M, steps, x, str(tmp_path), overlapped), nprocs=2, join=True)
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


def test_single_process_overlapped_protocol_equals_fused_step(c_oracle64):
    """run_sharded_overlapped with one rank (copies instead of collectives) == orc_run, for the joint model (theta planes)"""
    d, M = 5, 4
    data, _, _ = make_data(d, seed=1, joint=True)
    x = np.ascontiguousarray(data.x, np.float64)
    cfg = _cfg(True, 0, 1, d, M)
    eng = OracleShardEngine(c_oracle64, cfg, x, None, prng.PRNGKey(2))
    buf = OverlapBuffers(eng, 1, torch.device("cpu"), torch.float64)
    run_sharded_overlapped(eng, 0, 4, buf)
    st = c_oracle64.new_state(cfg, prng.PRNGKey(2))
    c_oracle64.run(cfg, x, None, st, 0, 4)
    assert np.array_equal(eng.st["z"], st["z"]) and np.array_equal(eng.st["theta"], st["theta"])
    # plane 0 holds the values of the final state (what sample_sharded returns / hands to the callback)
    v = buf.vals.view(M, -1).numpy()
    assert np.array_equal(v[:, :eng.D], st["z"].reshape(M, -1)) and np.array_equal(v[:, eng.D:eng.D + eng.P], st["theta"].reshape(M, -1))


def test_overlapped_protocol_notices_a_replaced_state(c_oracle64):
    """set_state between two overlapped chunks (checkpoint restore): the values gathered after the last step of the first chunk -- and the
    kernel slab computed from them -- belong to the old particles and must not serve the first step of the second chunk (the engine
    clears kmat_ext, run_sharded_overlapped compares Engine.state_gen and gathers again).  Result == the same steps run from that state."""
    d, M = 5, 4
    data, _, _ = make_data(d, seed=1, joint=True)
    x = np.ascontiguousarray(data.x, np.float64)
    cfg = _cfg(True, 0, 1, d, M)
    eng = OracleShardEngine(c_oracle64, cfg, x, None, prng.PRNGKey(2))
    buf = OverlapBuffers(eng, 1, torch.device("cpu"), torch.float64)
    run_sharded_overlapped(eng, 0, 2, buf)
    assert eng.slabs_consumed == 2
    other = c_oracle64.new_state(cfg, prng.PRNGKey(9))          # a different particle set, as a checkpoint would bring
    c_oracle64.run(cfg, x, None, other, 0, 2)
    eng.set_state(**{k: other[k] for k in ("z", "v_z", "theta", "v_theta", "key", "baseline")})
    run_sharded_overlapped(eng, 2, 2, buf)                        # (the adapter's asserts fail if the stale slab / plane were used)
    assert eng.slabs_consumed == 4
    c_oracle64.run(cfg, x, None, other, 2, 2)
    assert np.array_equal(eng.st["z"], other["z"]) and np.array_equal(eng.st["theta"], other["theta"])


def test_overlapped_then_packed_then_overlapped(c_oracle64):
    """mixed protocols on one engine: the packed steps move the particles without refreshing plane 0; the next overlapped chunk must
    notice (Engine.step_update bumps state_gen) and gather again.  Result == the fused run."""
    d, M = 5, 4
    data, _, _ = make_data(d, seed=1, joint=True)
    x = np.ascontiguousarray(data.x, np.float64)
    cfg = _cfg(True, 0, 1, d, M)
    eng = OracleShardEngine(c_oracle64, cfg, x, None, prng.PRNGKey(2))
    buf = OverlapBuffers(eng, 1, torch.device("cpu"), torch.float64)
    send, recv = make_buffers(eng, 1, torch.device("cpu"), torch.float64)
    run_sharded_overlapped(eng, 0, 2, buf)
    run_sharded(eng, 2, 2, send, recv)
    run_sharded_overlapped(eng, 4, 2, buf)     # (the adapter asserts that plane 0 holds the current particles)
    st = c_oracle64.new_state(cfg, prng.PRNGKey(2))
    c_oracle64.run(cfg, x, None, st, 0, 6)
    assert np.array_equal(eng.st["z"], st["z"]) and np.array_equal(eng.st["theta"], st["theta"])
