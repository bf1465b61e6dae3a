"""Particle-sharded SVGD over several ranks (one process per GPU, torch.distributed).

The reference has no multi-device code.  Every estimator of a step is per particle; only the
kernel matrix / phi couple particles.  Two protocols over the same arithmetic (bit-identical results):

``run_sharded`` -- one collective per step
    phase A  engine.step_local(t, send)      estimators for this rank's M/G particles, packed rows
                                             [z | grad_z | theta | grad_theta]
    exchange ONE all_gather_into_tensor(recv, send)     (RCCL over xGMI with backend "nccl")
    phase B  engine.step_update(t, recv)     kernel slab, phi and optimizer step for the local particles

``run_sharded_overlapped`` -- the default of ``sample_sharded`` and ``bench.py``
    The kernel matrix and the repulsive term need the VALUES [z | theta] of all particles, and those are final as soon as the previous
    optimizer step is done; only the GRADIENTS depend on phase A.  So after phase B every rank exports its new values and they are
    all-gathered on a SIDE stream while phase A of the next step runs; the kernel-matrix slab is computed from them on that same stream,
    behind the gather.  Between the phases only the gradient rows travel (half the bytes), and phase B is phi + optimizer.
        side stream:   all_gather(values(t)) -> kernel-matrix slab(t) ----.
        main stream:   phase A(t) ------------> all_gather(gradients(t)) -> phase B(t), which also writes values(t+1) into the send rows
    Both collectives are issued on the same process group in the same order on every rank (values(t), then gradients(t)).

PRNG rows are indexed by the global particle id and phi sums over b in global order, so the result does not depend
on the number of ranks (tests/test_distributed_gloo.py checks bit-equality against the single-rank run for both protocols).

``engine`` is anything exposing the calls used below with tensors' data_ptr()s: the HIP Engine (dibs_amd.engine) in production, the
oracle adapter in the CPU gloo test."""
import numpy as np


def make_stream():
    """A dedicated (non-default) torch stream: engine kernels and the collective must share ONE stream so that phase A ->
    all-gather -> phase B are ordered.  torch's default stream has handle 0, which the engine cannot adopt."""
    import torch
    return torch.cuda.Stream()


def make_buffers(engine, world_size, device, dtype):
    import torch
    n = engine.gather_elems_per_rank()
    send = torch.zeros(n, dtype=dtype, device=device)
    recv = torch.zeros(n * world_size, dtype=dtype, device=device)
    return send, recv


def run_sharded(engine, t_start, n_steps, send, recv, group=None):
    """steps t_start .. t_start + n_steps - 1; one collective per step (none when the group has one rank)."""
    import torch.distributed as dist
    single = (not dist.is_initialized()) or dist.get_world_size(group) == 1
    for t in range(t_start, t_start + n_steps):
        engine.step_local(t, send.data_ptr())
        if single:
            recv.copy_(send)
        else:
            dist.all_gather_into_tensor(recv, send, group=group)
        engine.step_update(t, recv.data_ptr())


class OverlapBuffers:
    """Buffers of the overlapped protocol: per-rank send rows for values and gradients, and ONE allocation `planes` = [2][M][Ev]
    (plane 0: values [z | theta] of all particles, plane 1: their gradients) that phase B reads.  On a GPU the values travel on `side`
    (a second torch stream); `vals_ready` is recorded there after each gather of the values."""

    def __init__(self, engine, world_size, device, dtype):
        import torch
        n = engine.plane_elems_per_rank()
        self.n, self.world = n, world_size
        self.vsend = torch.zeros(n, dtype=dtype, device=device)
        self.gsend = torch.zeros(n, dtype=dtype, device=device)
        self.planes = torch.zeros(2 * n * world_size, dtype=dtype, device=device)
        self.vals, self.grads = self.planes[:n * world_size], self.planes[n * world_size:]
        self.cuda = torch.device(device).type == "cuda"
        self.side = torch.cuda.Stream(device=device) if self.cuda else None
        self.vals_ready = torch.cuda.Event() if self.cuda else None
        self.exported = torch.cuda.Event() if self.cuda else None
        self.fresh = False   # plane 0 holds the values of the engine's current state ...
        self.gen = None      # ... i.e. of this generation of it (Engine.state_gen: bumped by init_particles / set_state)


def torch_mul(src, dst):
    import torch
    torch.mul(src, 1, out=dst[:src.numel()] if dst.numel() != src.numel() else dst)


def _gather(dst, src, group, single):
    import torch.distributed as dist
    if single:
        if dst.is_cuda:
            # a copy KERNEL, not Tensor.copy_ (= hipMemcpyAsync device-to-device): on a side stream the runtime's copy path resolves the
            # cross-stream dependency on the host -- measured 360 us per step instead of 90 in the single-GPU emulation of an 8-way rank
            # (profiles/round5_shard_scaling.txt); the collectives of a real group are kernels
            torch_mul(src, dst)
        else:
            dst.copy_(src)
    else:
        dist.all_gather_into_tensor(dst, src, group=group)


def _exchange_values(engine, buf, group, single, exported=False):
    """this rank's [z | theta] (exported here on the current = engine stream unless phase B already wrote them into the send rows) are
    gathered into plane 0 on the side stream"""
    import torch
    if not exported:
        engine.export_values(buf.vsend.data_ptr())
    if buf.cuda:
        buf.exported.record(torch.cuda.current_stream())
        with torch.cuda.stream(buf.side):
            buf.side.wait_event(buf.exported)
            _gather(buf.vals, buf.vsend, group, single)
            engine.kmat_values(buf.vals.data_ptr(), buf.side.cuda_stream)   # kernel-matrix slab of the next phase B, behind the gather
            buf.vals_ready.record(buf.side)
    else:   # (CPU harness of the protocol: tests/test_distributed_gloo.py)
        _gather(buf.vals, buf.vsend, group, single)
        engine.kmat_values(buf.vals.data_ptr(), 0)
    buf.fresh = True
    buf.gen = getattr(engine, "state_gen", None)


def _stale(engine, buf):
    """plane 0 (and the kernel slab computed from it) does not belong to the engine's current particles"""
    return (not buf.fresh) or buf.gen != getattr(engine, "state_gen", None)


def _is_single(group, always_collective):
    """one rank: device copies stand in for the collectives, unless the caller wants the real calls (a process group of one rank)"""
    import torch.distributed as dist
    if not dist.is_initialized():
        return True
    return dist.get_world_size(group) == 1 and not always_collective


def refresh_values(engine, buf, group=None, always_collective=False):
    """(re-)gather the values after the engine's state was set from outside (init_particles / set_state); run_sharded_overlapped does it
    on demand, callers that time a run do it beforehand"""
    _exchange_values(engine, buf, group, _is_single(group, always_collective))


def run_sharded_overlapped(engine, t_start, n_steps, buf, group=None, always_collective=False):
    """steps t_start .. t_start + n_steps - 1 with the values exchanged off the critical path (see the module docstring).  Call it inside
    ``torch.cuda.stream(<the engine's stream>)`` on a GPU.  A state set from outside since the last call (init_particles, set_state) is
    noticed through the engine's state generation and the values are gathered again.  always_collective: issue the all-gathers also in a group of one rank (smoke tests
    of the RCCL call path on one GPU)."""
    import torch
    single = _is_single(group, always_collective)
    if _stale(engine, buf):
        _exchange_values(engine, buf, group, single)
    for t in range(t_start, t_start + n_steps):
        engine.step_local_grads(t, buf.gsend.data_ptr())                                # phase A
        _gather(buf.grads, buf.gsend, group, single)                                    # the exchange on the critical path: gradients only
        if buf.cuda:
            torch.cuda.current_stream().wait_event(buf.vals_ready)                      # values + kernel matrix (done long ago)
        engine.step_update_planes(t, buf.planes.data_ptr(), buf.vsend.data_ptr())       # phase B: phi + optimizer (+ new values -> send rows)
        _exchange_values(engine, buf, group, single, exported=True)                     # values of step t + 1, beside its phase A


def _all_particles(eng, buf, n_particles, group):
    """z (and theta) of ALL ranks' particles as numpy arrays, on every rank: plane 0 of the overlapped protocol already holds them on
    the device after every step (no extra collective, one device-to-host copy)."""
    import torch
    import torch.distributed as dist
    if _stale(eng, buf):
        _exchange_values(eng, buf, group, (not dist.is_initialized()) or dist.get_world_size(group) == 1)
    if buf.cuda:
        buf.vals_ready.synchronize()
    D, P = eng.d * eng.k * 2, eng.P
    v = buf.vals.view(n_particles, -1).cpu().numpy()
    return np.ascontiguousarray(v[:, :D]).reshape(n_particles, eng.d, eng.k, 2), (np.ascontiguousarray(v[:, D:D + P]) if P else None)


def init_ipc_comm(engine, group=None):
    """The engine's exchange through mapped peer memory (``dibs_engine_comm_init_ipc``) for the ranks of ``group``: several ranks on ONE
    device (RCCL refuses that) or devices with peer access.  torch.distributed (any backend, gloo included) only carries the 128-byte
    blobs; the data path never touches it."""
    import torch.distributed as dist
    blobs = [None] * dist.get_world_size(group)
    dist.all_gather_object(blobs, engine.ipc_export(), group=group)
    engine.comm_init_ipc(blobs)


def init_native_comm(engine, group=None, n_comms=2):
    """RCCL communicator(s) INSIDE the engine (dibs_engine_comm_init): rank 0 draws the unique ids, torch.distributed carries the bytes
    to the other ranks (control plane only -- the data path of ``engine.run_sharded`` never touches torch).
    ncclCommInitRank blocks until every rank of the group has entered it: callers that want a fallback must agree on a LOCAL probe
    (``engine.comm_unique_ids(1)`` raises when librccl or one of its symbols is missing) across the ranks BEFORE calling this
    (bench.py does); a failure inside the bring-up on a subset of the ranks is fatal (RCCL's timeout ends the job)."""
    import torch.distributed as dist
    box = [engine.comm_unique_ids(n_comms) if dist.get_rank(group) == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    engine.comm_init(box[0])


def sample_sharded_native(dibs, *, key, n_particles, steps, n_dim_particles=None, callback_every=None, callback=None, group=None,
                          overlapped=True):
    """``sample`` with the particles sharded over the ranks of ``group`` and the whole step loop, collectives included, inside the engine
    (``dibs_engine_run_sharded``: ncclAllGather on the engine's own streams).  Every rank calls it with the same arguments and gets
    the full result (all particles)."""
    import torch
    import torch.distributed as dist
    from . import random
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if n_particles % world:
        raise ValueError("n_particles must be divisible by the number of ranks")
    n_dim = n_dim_particles or dibs.n_vars
    eng = dibs._new_engine(n_particles, n_dim, rank=rank, n_ranks=world, device_id=torch.cuda.current_device())
    try:
        init_native_comm(eng, group, 2 if overlapped else 1)
        eng.init_particles(random.as_key(key))
        if dibs.latent_prior_std is None:
            dibs.latent_prior_std = float(np.float32(1.0) / np.sqrt(np.float32(n_dim)))
        callback_every = callback_every or steps
        for t in (range(0, steps, callback_every) if steps else range(0)):
            eng.run_sharded(t, callback_every, overlapped)
            if callback:
                zs_all, th_all = eng.gather_particles()
                kw = dict(dibs=dibs, t=t + callback_every, zs=zs_all)
                if dibs._joint:
                    kw["thetas"] = dibs._theta_out(th_all)
                callback(**kw)
        out_z, th_all = eng.gather_particles()
        if dibs._joint:
            return dibs.particle_to_g_lim(out_z), dibs._theta_out(th_all)
        return dibs.particle_to_g_lim(out_z)
    finally:
        eng.close()


def sample_sharded(dibs, *, key, n_particles, steps, n_dim_particles=None, callback_every=None, callback=None, group=None):
    """``MarginalDiBS.sample`` / ``JointDiBS.sample`` with the particles sharded over the ranks of ``group``, the step loop driven from
    Python with torch.distributed collectives (the harness the in-engine loop of ``sample_sharded_native`` is checked against).
    Every rank calls it with the same arguments and gets the full result (all particles)."""
    import torch
    import torch.distributed as dist
    from . import random
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if n_particles % world:
        raise ValueError("n_particles must be divisible by the number of ranks")
    n_dim = n_dim_particles or dibs.n_vars
    dev = torch.cuda.current_device()
    stream = make_stream()
    eng = dibs._new_engine(n_particles, n_dim, rank=rank, n_ranks=world, device_id=dev, stream=stream.cuda_stream)
    try:
        eng.init_particles(random.as_key(key))
        if dibs.latent_prior_std is None:
            dibs.latent_prior_std = float(np.float32(1.0) / np.sqrt(np.float32(n_dim)))
        callback_every = callback_every or steps
        with torch.cuda.stream(stream):
            buf = OverlapBuffers(eng, world, torch.device("cuda", dev), torch.float32)
            for t in (range(0, steps, callback_every) if steps else range(0)):
                run_sharded_overlapped(eng, t, callback_every, buf, group)
                if callback:
                    # same keyword arguments as MarginalDiBS.sample / JointDiBS.sample (svgd.py:318-324, :783-789): ALL particles,
                    # on every rank (the callback runs on every rank)
                    zs_all, th_all = _all_particles(eng, buf, n_particles, group)
                    kw = dict(dibs=dibs, t=t + callback_every, zs=zs_all)
                    if dibs._joint:
                        kw["thetas"] = dibs._theta_out(th_all)
                    callback(**kw)
            out_z, th_all = _all_particles(eng, buf, n_particles, group)
        stream.synchronize()
        if dibs._joint:
            return dibs.particle_to_g_lim(out_z), dibs._theta_out(th_all)
        return dibs.particle_to_g_lim(out_z)
    finally:
        eng.close()
