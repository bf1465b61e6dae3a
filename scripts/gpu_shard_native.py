"""Per-rank step cost of the in-engine sharded loop (dibs_engine_run_sharded) at R = 1, 2, 4, 8 on ONE GPU: one engine per PROCESS (the
production layout: three streams), loopback communicator (collectives skipped; the other ranks' rows stay zero), steady state.
    python scripts/gpu_shard_native.py            (spawns one child per (R, protocol))"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    R, ov = int(sys.argv[1]), bool(int(sys.argv[2]))
    from dibs_amd import random
    from dibs_amd._abi import make_config
    from dibs_amd.engine import Engine
    from dibs_amd.target import make_linear_gaussian_equivalent_model
    data, _, _ = make_linear_gaussian_equivalent_model(key=random.PRNGKey(0), n_vars=50, graph_prior_str="er", n_observations=100)
    e = Engine(make_config(n_vars=50, n_particles=128, n_observations=100, rank=0, n_ranks=R))
    e.set_data(data.x); e.init_particles(random.PRNGKey(1)); e.comm_init(None, 2)
    e.run_sharded(0, 320, ov)
    t0 = time.perf_counter(); e.run_sharded(320, 300, ov); dt = (time.perf_counter() - t0) / 300
    e.set_profiling(True); e.reset_timers(); e.run_sharded(620, 40, ov)
    tm = {k: round(v[0] / 40 * 1e3, 1) for k, v in e.timers().items()}
    print(f"R={R} Mloc={128 // R} {'overlapped' if ov else 'packed    '}: {dt * 1e6:6.1f} us/step   kernels {tm}", flush=True)
else:
    for R in (1, 2, 4, 8):
        for ov in (0, 1):
            subprocess.run([sys.executable, os.path.abspath(__file__), str(R), str(ov)])
