"""per-phase time of k_nn_grad (library built with -DDIBS_NN_STAMPS into dibs_amd/csrc/_dbg):
   DIBS_HIP_LIB=dibs_amd/csrc/_dbg/libdibs_hip.so python scripts/debug/nn_grad_phases.py config5 300"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from dibs_amd import random, _lib
from dibs_amd.engine import Engine
name, t0 = sys.argv[1], int(sys.argv[2])
cfg, x, mask = bench.make_workload(name, bench.CONFIGS[name]["M"])
eng = Engine(cfg); eng.set_data(x, mask); eng.init_particles(random.PRNGKey(1))
eng.run(0, t0)
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 64)()
lib.dibs_debug_nn_stamps(buf, 1)
n = 5
eng.run(t0, n)
lib.dibs_debug_nn_stamps(buf, 0)
names = ["loop ctl / tail", "graph build", "fwd build T_h", "fwd gemm", "fwd epilogue", "dmean + b2 | bwd wait", "bwd build T_h", "bwd gemm",
         "dpre store + col sums", "b1 / W2 leaves", "xtr gemm", "xtr epilogue (RMW)", "(samples)", "combine"]
for mode, mn in ((0, "theta"), (1, "z (reparam)"), (2, "z (score)")):
    row = [buf[mode * 16 + k] for k in range(16)]
    ns = row[12]
    if not ns:
        continue
    print(f"{name} t={t0} mode {mn}: {ns / n:.0f} sample gradients per step; block-time per sample gradient (us, 100 MHz clock):")
    tot = 0.0
    for k, nm in enumerate(names):
        if k == 12:
            continue
        us = row[k] / 100.0 / ns
        tot += us
        print(f"   {nm:28s} {us:8.2f}")
    print(f"   {'sum':28s} {tot:8.2f}")
