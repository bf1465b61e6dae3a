"""When do the limit graphs of the DenseNN joint model stop being empty?  (device run; picks the checkpoints of the config-5 variant whose E-SHD check is not vacuous)"""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dibs_amd import random
from dibs_amd.engine import Engine
spec = importlib.util.spec_from_file_location("g", os.path.join(ROOT, "tests", "golden", "make_joint_golden.py"))
gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
d, M, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
gen.CONFIGS["probe"] = dict(d=d, M=M, checkpoints=())
dibs, x, mask, g_true = gen.workload("probe", seed)
eng = Engine(dibs._make_config(M, d)); eng.set_data(x, mask); eng.init_particles(random.PRNGKey(seed + 1))
t = 0
for cp in [int(a) for a in sys.argv[4:]]:
    eng.run(t, cp - t); t = cp
    st = eng.get_state()
    sm = gen.summarise(dibs, g_true, st["z"], st["theta"], d, M)
    g = dibs.particle_to_g_lim(st["z"])
    print(f"d={d} M={M} seed={seed} step {cp}: E-SHD {sm['eshd']:.3f} DAGs {sm['ndag']} E-edges {sm['edges']:.2f} edges/particle min {g.sum((1,2)).min()} max {g.sum((1,2)).max()} true edges {int(g_true.sum())}", flush=True)
