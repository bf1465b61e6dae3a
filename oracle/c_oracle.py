"""ORACLE (test infrastructure only): ctypes wrapper of oracle/_build/liboracle_f{64,32}.so
(the C port in oracle/dibs_oracle.c).  Never imported by the product path."""
import ctypes as C
import os
import subprocess
import numpy as np

from dibs_amd._abi import DibsConfig  # struct layout only (include/dibs_hip.h)

_HERE = os.path.dirname(os.path.abspath(__file__))


class _Debug(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("scores", "logprobs_z", "logprobs_th", "w_lik", "w_acyc", "grad_z",
                                          "grad_theta", "kxx", "phi_z", "phi_theta", "node_scores", "g_samples")]


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


class COracle:
    def __init__(self, precision="f64"):
        asan = bool(os.environ.get("DIBS_ORACLE_ASAN"))   # sanitizer build (oracle/Makefile `asan`; needs LD_PRELOAD of libasan)
        path = os.path.join(_HERE, "_build", f"liboracle_{precision}{'_asan' if asan else ''}.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-s", "-C", _HERE] + (["asan"] if asan else []), check=True)
        self.lib = C.CDLL(path)
        self.real = np.float64 if precision == "f64" else np.float32
        self.lib.orc_theta_size.restype = C.c_int64
        assert self.lib.orc_real_bytes() == np.dtype(self.real).itemsize

    def _p(self, a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def theta_size(self, cfg):
        return int(self.lib.orc_theta_size(C.byref(cfg)))

    def split(self, key, num, layout=0):
        out = np.zeros((num, 2), np.uint32)
        self.lib.orc_split(self._p(np.asarray(key, np.uint32)), num, layout, self._p(out))
        return out

    def random_bits(self, key, n, layout=0):
        out = np.zeros(n, np.uint32)
        self.lib.orc_random_bits(self._p(np.asarray(key, np.uint32)), C.c_int64(n), layout, self._p(out))
        return out

    def normal(self, key, n, layout=0):
        out = np.zeros(n, np.float32)
        self.lib.orc_normal(self._p(np.asarray(key, np.uint32)), C.c_int64(n), layout, self._p(out))
        return out

    def logistic(self, key, n, layout=0, tiny=0):
        out = np.zeros(n, np.float32)
        self.lib.orc_logistic(self._p(np.asarray(key, np.uint32)), C.c_int64(n), layout, tiny, self._p(out))
        return out

    def init_particles(self, cfg, key):
        M, d, k = cfg.n_particles, cfg.n_vars, cfg.n_dim
        z = np.zeros((M, d, k, 2), self.real)
        P = self.theta_size(cfg)
        theta = np.zeros((M, P), self.real) if P else None
        key_out = np.zeros(2, np.uint32)
        rc = self.lib.orc_init_particles(C.byref(cfg), self._p(np.asarray(key, np.uint32)), self._p(z), self._p(theta),
                                         self._p(key_out))
        assert rc == 0, rc
        return z, theta, key_out

    def new_state(self, cfg, key):
        z, theta, key_out = self.init_particles(cfg, key)
        return dict(z=z, v_z=np.zeros_like(z), theta=theta, v_theta=None if theta is None else np.zeros_like(theta),
                    key=key_out, baseline=np.zeros(cfg.n_particles, self.real))

    def step(self, cfg, x, mask, st, t, debug=False, bge_mode=1, n_threads=0, mean_obs=None):
        x = np.ascontiguousarray(x, self.real)
        mask = None if mask is None else np.ascontiguousarray(mask, np.int32)
        mean_obs = None if mean_obs is None else np.ascontiguousarray(mean_obs, self.real)
        M, d, k, S = cfg.n_particles, cfg.n_vars, cfg.n_dim, cfg.n_grad_mc_samples
        P = self.theta_size(cfg)
        dbg, out = None, None
        if debug:
            r = self.real
            out = dict(scores=np.zeros((M, d, d), r), logprobs_z=np.zeros((M, S), r), logprobs_th=np.zeros((M, S), r),
                       w_lik=np.zeros((M, d, d), r), w_acyc=np.zeros((M, d, d), r), grad_z=np.zeros((M, d, k, 2), r),
                       grad_theta=np.zeros((M, max(P, 1)), r), kxx=np.zeros((M, M), r), phi_z=np.zeros((M, d, k, 2), r),
                       phi_theta=np.zeros((M, max(P, 1)), r), node_scores=np.zeros((M, S, d), r),
                       g_samples=np.zeros((M, S, d, d), np.uint8))
            dbg = _Debug(**{n: out[n].ctypes.data for n, _ in _Debug._fields_})
        rc = self.lib.orc_step(C.byref(cfg), self._p(x), self._p(mask), self._p(mean_obs), self._p(st["z"]), self._p(st["v_z"]),
                               self._p(st["theta"]), self._p(st["v_theta"]), self._p(st["key"]), self._p(st["baseline"]),
                               int(t), C.byref(dbg) if dbg is not None else None, int(bge_mode), int(n_threads))
        if rc != 0:
            raise RuntimeError(f"orc_step failed rc={rc}")
        return out

    def pack_stride(self, cfg):
        self.lib.orc_pack_stride.restype = C.c_int64
        return int(self.lib.orc_pack_stride(C.byref(cfg)))

    def new_local_state(self, cfg, key):
        """state of the shard cfg.rank of cfg.n_ranks (same init streams as the global state)"""
        full = DibsConfig.from_buffer_copy(cfg)
        full.rank, full.n_ranks = 0, 1
        st = self.new_state(full, key)
        Ml = cfg.n_particles // cfg.n_ranks
        sl = slice(cfg.rank * Ml, (cfg.rank + 1) * Ml)
        return {k: (v[sl].copy() if (v is not None and k != "key") else v) for k, v in st.items()}

    def step_local(self, cfg, x, mask, st, t, pack_local, bge_mode=1, n_threads=0, mean_obs=None):
        x = np.ascontiguousarray(x, self.real)
        mask = None if mask is None else np.ascontiguousarray(mask, np.int32)
        mean_obs = None if mean_obs is None else np.ascontiguousarray(mean_obs, self.real)
        rc = self.lib.orc_step_local(C.byref(cfg), self._p(x), self._p(mask), self._p(mean_obs), self._p(st["z"]),
                                     self._p(st["theta"]), self._p(st["key"]), self._p(st["baseline"]), int(t),
                                     self._p(pack_local), None, int(bge_mode), int(n_threads))
        if rc != 0:
            raise RuntimeError(f"orc_step_local failed rc={rc}")

    def step_update(self, cfg, pack_all, st, n_threads=0):
        rc = self.lib.orc_step_update(C.byref(cfg), self._p(pack_all), self._p(st["z"]), self._p(st["v_z"]),
                                      self._p(st["theta"]), self._p(st["v_theta"]), None, int(n_threads))
        if rc != 0:
            raise RuntimeError(f"orc_step_update failed rc={rc}")

    def run(self, cfg, x, mask, st, t_start, n_steps, bge_mode=1, n_threads=0, mean_obs=None):
        x = np.ascontiguousarray(x, self.real)
        mask = None if mask is None else np.ascontiguousarray(mask, np.int32)
        mean_obs = None if mean_obs is None else np.ascontiguousarray(mean_obs, self.real)
        rc = self.lib.orc_run(C.byref(cfg), self._p(x), self._p(mask), self._p(mean_obs), self._p(st["z"]), self._p(st["v_z"]),
                              self._p(st["theta"]), self._p(st["v_theta"]), self._p(st["key"]), self._p(st["baseline"]),
                              int(t_start), int(n_steps), int(bge_mode), int(n_threads))
        if rc != 0:
            raise RuntimeError(f"orc_run failed rc={rc}")

    def score_graphs(self, cfg, x, mask, g, theta=None, bge_mode=1, mean_obs=None):
        x = np.ascontiguousarray(x, self.real)
        mask = None if mask is None else np.ascontiguousarray(mask, np.int32)
        g = np.ascontiguousarray(g, np.int32)
        theta = None if theta is None else np.ascontiguousarray(theta, self.real)
        cfg2 = DibsConfig.from_buffer_copy(cfg)
        cfg2.n_observations = x.shape[0]
        out = np.zeros(g.shape[0], self.real)
        mean_obs = None if mean_obs is None else np.ascontiguousarray(mean_obs, self.real)
        rc = self.lib.orc_score_graphs(C.byref(cfg2), self._p(x), self._p(mask), self._p(mean_obs), self._p(g), self._p(theta),
                                       g.shape[0], self._p(out), int(bge_mode))
        assert rc == 0
        return out
