"""Per-kernel time of a config late in its trajectory: python scripts/gpu_late_breakdown.py <config> <t> [steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dibs_amd import random
from dibs_amd.engine import Engine
name, t0 = sys.argv[1], int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg, x, mask = bench.make_workload(name, bench.CONFIGS[name]["M"])
eng = Engine(cfg); eng.set_data(x, mask); eng.init_particles(random.PRNGKey(1))
t = 0
for cp in sorted(set([5, t0])):
    eng.run(t, cp - t); t = cp
    snap = {k: v for k, v in eng.get_state().items() if v is not None}
    eng.set_profiling(True); eng.reset_timers(); eng.run(t, K)
    tm = eng.timers(); eng.set_profiling(False); eng.set_state(**snap)
    print(f"config {name} t={t}: " + "  ".join(f"{k} {ms / K * 1e3:.1f}" for k, (ms, n) in tm.items()) + f"   sum {sum(ms for ms, n in tm.values()) / K * 1e3:.1f} us/step", flush=True)
    if "LOGPROBS_Z" in dir(eng) or True:
        try:
            lp = eng.read("LOGPROBS_Z").reshape(eng.Mloc, -1).astype(np.float64)
            w = np.exp(lp - lp.max(1, keepdims=True)); w /= w.sum(1, keepdims=True)
            print("   Z-estimator softmax weights: samples with w > 0 in float32 per particle: mean", float((w.astype(np.float32) > 0).sum(1).mean()), "max", int((w.astype(np.float32) > 0).sum(1).max()),
                  "| with w >= 2^-30: mean", float((w >= 2.0 ** -30).sum(1).mean()), "max", int((w >= 2.0 ** -30).sum(1).max()))
            lp = eng.read("LOGPROBS_THETA").reshape(eng.Mloc, -1).astype(np.float64)
            w = np.exp(lp - lp.max(1, keepdims=True)); w /= w.sum(1, keepdims=True)
            print("   theta-estimator: mean", float((w.astype(np.float32) > 0).sum(1).mean()), "max", int((w.astype(np.float32) > 0).sum(1).max()),
                  "| with w >= 2^-30: mean", float((w >= 2.0 ** -30).sum(1).mean()), "max", int((w >= 2.0 ** -30).sum(1).max()))
        except Exception as ex:
            print("   (no logprobs)", ex)
