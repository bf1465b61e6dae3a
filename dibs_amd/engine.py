"""Thin object wrapper over the C ABI (include/dibs_hip.h): one Engine = one device + one stream."""
import ctypes as C
import numpy as np

from . import _lib
from ._abi import BUF, K_COUNT, KERNELS, DibsConfig

_BUF_DTYPE = {"NODE_SCORES": np.float64, "PARENT_MASKS": np.uint64}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    def __init__(self, cfg: DibsConfig, stream=None):
        self.lib = _lib.load()
        self.cfg = cfg
        self._h = C.c_void_p()
        if stream is not None and int(stream) == 0:
            # handle 0 is HIP's legacy default stream; the C ABI reads NULL as "create your own stream", which would leave
            # the engine unordered with the caller's work (torch.cuda.current_stream() is 0 unless a Stream is active)
            raise ValueError("pass the handle of a non-default stream (e.g. torch.cuda.Stream().cuda_stream) or None")
        _lib.check(self.lib.dibs_engine_create(C.byref(cfg), C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.M, self.d, self.k = cfg.n_particles, cfg.n_vars, cfg.n_dim
        self.Mloc = cfg.n_particles // cfg.n_ranks
        self.P = int(self.lib.dibs_engine_theta_size(self._h))
        self.state_gen = 0   # bumped whenever the particles are replaced from outside (init_particles / set_state): the overlapped
                             # exchange of dibs_amd.distributed compares it with the generation its gathered values belong to

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.dibs_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_data(self, x, interv_mask=None, bge_mean_obs=None):
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape == (self.cfg.n_observations, self.d), x.shape
        m = None if interv_mask is None else np.ascontiguousarray(interv_mask, np.int32)
        mo = None if bge_mean_obs is None else np.ascontiguousarray(bge_mean_obs, np.float32)
        _lib.check(self.lib.dibs_engine_set_data(self._h, _ptr(x), _ptr(m), _ptr(mo)))

    def init_particles(self, key):
        key = np.ascontiguousarray(key, np.uint32).reshape(2)
        _lib.check(self.lib.dibs_engine_init_particles(self._h, _ptr(key)))
        self.state_gen += 1

    def set_state(self, z=None, v_z=None, theta=None, v_theta=None, key=None, baseline=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
        z, v_z, theta, v_theta, baseline = f(z), f(v_z), f(theta), f(v_theta), f(baseline)
        key = None if key is None else np.ascontiguousarray(key, np.uint32)
        _lib.check(self.lib.dibs_engine_set_state(self._h, _ptr(z), _ptr(v_z), _ptr(theta), _ptr(v_theta), _ptr(key),
                                                  _ptr(baseline)))
        self.state_gen += 1

    def get_state(self):
        z = np.empty((self.Mloc, self.d, self.k, 2), np.float32)
        v_z = np.empty_like(z)
        theta = np.empty((self.Mloc, self.P), np.float32) if self.P else None
        v_theta = np.empty_like(theta) if self.P else None
        key = np.empty(2, np.uint32)
        baseline = np.empty(self.Mloc, np.float32)
        _lib.check(self.lib.dibs_engine_get_state(self._h, _ptr(z), _ptr(v_z), _ptr(theta), _ptr(v_theta), _ptr(key),
                                                  _ptr(baseline)))
        return dict(z=z, v_z=v_z, theta=theta, v_theta=v_theta, key=key, baseline=baseline)

    def run(self, t_start, n_steps):
        _lib.check(self.lib.dibs_engine_run(self._h, int(t_start), int(n_steps)))
        self.state_gen += 1   # (the particles moved without the overlapped exchange: its gathered values are stale)

    def step_local(self, t, send_ptr):
        _lib.check(self.lib.dibs_engine_step_local(self._h, int(t), C.c_void_p(send_ptr)))

    def step_update(self, t, recv_ptr):
        _lib.check(self.lib.dibs_engine_step_update(self._h, int(t), C.c_void_p(recv_ptr)))
        self.state_gen += 1   # (packed protocol: plane 0 of dibs_amd.distributed's overlapped exchange does not follow)

    def gather_elems_per_rank(self):
        return int(self.lib.dibs_engine_gather_elems_per_rank(self._h))

    # overlapped exchange (include/dibs_hip.h): values right after the optimizer step, gradients between the phases
    def plane_elems_per_rank(self):
        return int(self.lib.dibs_engine_plane_elems_per_rank(self._h))

    def export_values(self, vals_send_ptr):
        _lib.check(self.lib.dibs_engine_export_values(self._h, C.c_void_p(vals_send_ptr)))

    def step_local_grads(self, t, grads_send_ptr):
        _lib.check(self.lib.dibs_engine_step_local_grads(self._h, int(t), C.c_void_p(grads_send_ptr)))

    def kmat_values(self, vals_all_ptr, stream):
        if not stream:
            raise ValueError("kmat_values needs the handle of the (non-default) stream the values were gathered on")
        _lib.check(self.lib.dibs_engine_kmat_values(self._h, C.c_void_p(vals_all_ptr), C.c_void_p(stream)))

    def step_update_planes(self, t, planes_ptr, vals_send_ptr=None):
        _lib.check(self.lib.dibs_engine_step_update_planes(self._h, int(t), C.c_void_p(planes_ptr),
                                                           C.c_void_p(vals_send_ptr) if vals_send_ptr else None))

    # in-engine exchange (include/dibs_hip.h): RCCL communicator(s) of this rank and the sharded step loop in C
    COMM_ID_BYTES = 128

    def comm_unique_ids(self, n=1):
        """n fresh RCCL unique ids (bytes of n * 128): call on ONE rank and hand the bytes to every rank"""
        buf = (C.c_char * (self.COMM_ID_BYTES * n))()
        for i in range(n):
            _lib.check(self.lib.dibs_comm_unique_id(C.byref(buf, i * self.COMM_ID_BYTES)))
        return bytes(buf)

    def comm_init(self, ids, n_loopback=2):
        """ids: the bytes of 1 or 2 unique ids (comm_unique_ids on one rank, handed to all).  ids=None: loopback (collectives skipped;
        what ONE rank of an N-way run costs per step, measured on one GPU -- scripts/gpu_shard_scaling.py)."""
        if ids is None:
            _lib.check(self.lib.dibs_engine_comm_init(self._h, None, int(n_loopback)))
            return
        ids = bytes(ids)
        assert len(ids) in (self.COMM_ID_BYTES, 2 * self.COMM_ID_BYTES)
        _lib.check(self.lib.dibs_engine_comm_init(self._h, ids, len(ids) // self.COMM_ID_BYTES))

    # the same loop over mapped peer memory (include/dibs_hip.h: ranks that share a device, one process per rank)
    IPC_HANDLE_BYTES = 128

    def ipc_export(self):
        """allocate this rank's exchange arena; returns the 128 bytes its peers need to map it"""
        buf = (C.c_char * self.IPC_HANDLE_BYTES)()
        _lib.check(self.lib.dibs_engine_ipc_export(self._h, buf))
        return bytes(buf)

    def comm_init_ipc(self, blobs):
        """blobs: the ipc_export() bytes of ALL ranks in rank order (a list of bytes or their concatenation)"""
        blobs = b"".join(blobs) if isinstance(blobs, (list, tuple)) else bytes(blobs)
        assert len(blobs) == self.IPC_HANDLE_BYTES * self.cfg.n_ranks, len(blobs)
        _lib.check(self.lib.dibs_engine_comm_init_ipc(self._h, blobs))

    def comm_destroy(self):
        _lib.check(self.lib.dibs_engine_comm_destroy(self._h))

    def run_sharded(self, t_start, n_steps, overlapped=False):
        _lib.check(self.lib.dibs_engine_run_sharded(self._h, int(t_start), int(n_steps), int(bool(overlapped))))
        self.state_gen += 1

    def gather_particles(self):
        """z [M, d, k, 2] (and theta [M, P]) of all ranks' particles, on every rank"""
        z = np.empty((self.M, self.d, self.k, 2), np.float32)
        th = np.empty((self.M, self.P), np.float32) if self.P else None
        _lib.check(self.lib.dibs_engine_gather_particles(self._h, _ptr(z), _ptr(th)))
        return z, th

    def eval_gradients(self, t, keys_theta=None, keys_lik=None, keys_prior=None):
        """Gradient estimators of one step for the current particles with explicit per-particle keys (include/dibs_hip.h,
        dibs_engine_eval_gradients).  keys_*: uint32 [Mloc, 2] or None.  Returns a dict with the outputs that were computed."""
        k = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, np.uint32).reshape(self.Mloc, 2))
        kt, kl, kp = k(keys_theta), k(keys_lik), k(keys_prior)
        lik, prior = kl is not None or kt is not None, kp is not None
        gz = np.empty((self.Mloc, self.d, self.k, 2), np.float32) if lik else None
        bl = np.empty(self.Mloc, np.float32) if lik else None
        gt = np.empty((self.Mloc, self.P), np.float32) if lik and self.P else None
        gp = np.empty((self.Mloc, self.d, self.k, 2), np.float32) if prior else None
        _lib.check(self.lib.dibs_engine_eval_gradients(self._h, int(t), _ptr(kt), _ptr(kl), _ptr(kp), _ptr(gz), _ptr(bl), _ptr(gt), _ptr(gp)))
        return dict(grad_z_lik=gz, baseline=bl, grad_theta=gt, grad_z_prior=gp)

    def flag_fallbacks(self):
        """chunks this engine had to repeat on events after an in-kernel flag wait timed out (include/dibs_hip.h)"""
        return int(self.lib.dibs_engine_flag_fallbacks(self._h))

    def debug_drop_next_flag(self):
        _lib.check(self.lib.dibs_engine_debug_drop_next_flag(self._h))

    def sync(self):
        _lib.check(self.lib.dibs_engine_sync(self._h))

    def read(self, name):
        nbytes = int(self.lib.dibs_engine_buffer_bytes(self._h, BUF[name]))
        if nbytes < 0:
            raise KeyError(name)
        dt = _BUF_DTYPE.get(name, np.float32)
        out = np.empty(nbytes // np.dtype(dt).itemsize, dt)
        _lib.check(self.lib.dibs_engine_read_buffer(self._h, BUF[name], _ptr(out), nbytes))
        return out

    def set_profiling(self, on):
        _lib.check(self.lib.dibs_engine_set_profiling(self._h, 2 if on == 2 else int(bool(on))))

    def reset_timers(self):
        _lib.check(self.lib.dibs_engine_reset_timers(self._h))

    def timers(self):
        ms = np.zeros(K_COUNT, np.float64)
        n = np.zeros(K_COUNT, np.int64)
        _lib.check(self.lib.dibs_engine_get_timers(self._h, _ptr(ms), _ptr(n), K_COUNT))
        return {KERNELS[i]: (float(ms[i]), int(n[i])) for i in range(K_COUNT) if n[i]}

    def counters(self, n=32):
        out = np.zeros(n, np.float64)
        _lib.check(self.lib.dibs_engine_get_counters(self._h, _ptr(out), n))
        return out
