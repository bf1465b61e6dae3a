"""One single-rank run of a small workload in a process of its own (tests/test_gpu_flags.py: under HSA_CU_MASK, or beside a process that fills
the GPU).  python tests/tools/single_run_worker.py <out.npz> <case-json>;  case as in ipc_rank_worker.py, "chunks": [[t0, n], ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))


def main():
    out, case = sys.argv[1], json.loads(sys.argv[2])
    from dibs_amd import random as prng
    from dibs_amd.engine import Engine
    from ipc_rank_worker import case_config, case_data
    eng = Engine(case_config(case))
    eng.set_data(case_data(case))
    eng.init_particles(prng.PRNGKey(case.get("seed", 8)))
    for t0, n in case["chunks"]:
        eng.run(t0, n)
    st = eng.get_state()
    extra = {} if st["theta"] is None else dict(theta=st["theta"])
    np.savez(out, z=st["z"], key=st["key"], flag_fallbacks=np.int64(eng.flag_fallbacks()), **extra)
    eng.close()


def burn(seconds):
    """fill the GPU from another process: large matrix products back to back"""
    import time
    import torch
    a = torch.randn(8192, 8192, device="cuda")
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20):
            a = torch.tanh(a @ a * 1e-4)
        torch.cuda.synchronize()


if __name__ == "__main__":
    if sys.argv[1] == "--burn":
        burn(float(sys.argv[2]))
    else:
        main()
