"""run a config to step t0 untraced-ish, then 3 steps (for rocprofv3 --kernel-trace --stats): python scripts/debug/late_trace.py <config> <t0>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from dibs_amd import random
from dibs_amd.engine import Engine
name, t0 = sys.argv[1], int(sys.argv[2])
cfg, x, mask = bench.make_workload(name, bench.CONFIGS[name]["M"])
eng = Engine(cfg); eng.set_data(x, mask); eng.init_particles(random.PRNGKey(1))
eng.run(0, t0)
print("MARK late window begins", flush=True)
eng.run(t0, 3)
