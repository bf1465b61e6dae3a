"""Particle-sharded SVGD over several ranks (one process per GPU, torch.distributed).

The reference has no multi-device code.  Every estimator of a step is per particle; only the
kernel matrix / phi couple particles, so a step is
    phase A  engine.step_local(t, send)      estimators for this rank's M/G particles, packed rows
                                             [z | grad_z | theta | grad_theta]
    exchange ONE all_gather_into_tensor(recv, send)     (RCCL over xGMI with backend "nccl")
    phase B  engine.step_update(t, recv)     kernel slab, phi and optimizer step for the local particles
PRNG rows are indexed by the global particle id and phi sums over b in global order, so the result does not depend
on the number of ranks (tests/test_distributed_gloo.py checks bit-equality against the single-rank run).

``engine`` is anything exposing step_local / step_update / gather_elems_per_rank with tensors' data_ptr()s: the HIP
Engine (dibs_amd.engine) in production, the oracle adapter in the CPU gloo test."""
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


def _gather_particles(eng, n_particles, group):
    """z (and theta) of all ranks' particles as numpy arrays, on every rank"""
    import torch
    import torch.distributed as dist
    eng.sync()
    st = eng.get_state()
    z = torch.from_numpy(st["z"]).cuda()
    zs = torch.empty((n_particles,) + tuple(z.shape[1:]), dtype=z.dtype, device=z.device)
    dist.all_gather_into_tensor(zs, z, group=group)
    ths = None
    if st["theta"] is not None:
        th = torch.from_numpy(st["theta"]).cuda()
        ths = torch.empty((n_particles, th.shape[1]), dtype=th.dtype, device=th.device)
        dist.all_gather_into_tensor(ths, th, group=group)
        ths = ths.cpu().numpy()
    return zs.cpu().numpy(), ths


def sample_sharded(dibs, *, key, n_particles, steps, n_dim_particles=None, callback_every=None, callback=None, group=None):
    """``MarginalDiBS.sample`` / ``JointDiBS.sample`` with the particles sharded over the ranks of ``group``.
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
            send, recv = make_buffers(eng, world, torch.device("cuda", dev), torch.float32)
            for t in (range(0, steps, callback_every) if steps else range(0)):
                run_sharded(eng, t, callback_every, send, recv, group)
                if callback:
                    # same keyword arguments as MarginalDiBS.sample / JointDiBS.sample (svgd.py:318-324, :783-789): ALL particles
                    stream.synchronize()
                    zs_all, th_all = _gather_particles(eng, n_particles, group)
                    kw = dict(dibs=dibs, t=t + callback_every, zs=zs_all)
                    if dibs._joint:
                        kw["thetas"] = dibs._theta_out(th_all)
                    callback(**kw)
        stream.synchronize()
        out_z, th_all = _gather_particles(eng, n_particles, group)
        if dibs._joint:
            return dibs.particle_to_g_lim(out_z), dibs._theta_out(th_all)
        return dibs.particle_to_g_lim(out_z)
    finally:
        eng.close()
