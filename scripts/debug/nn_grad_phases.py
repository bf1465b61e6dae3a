"""per-phase time of k_nn_grad (library built with -DDIBS_NN_STAMPS into dibs_amd/csrc/_dbg: make -C dibs_amd/csrc stamps):
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
buf = (C.c_ulonglong * 128)()
lib.dibs_debug_nn_stamps(buf, 1)
n = 5
eng.run(t0, n)
lib.dibs_debug_nn_stamps(buf, 0)
names = ["barrier in front of a sample", "graph build", "slices build (+ barrier)", "fwd gemm + activations", "dmean", "validity bits | wait: slices free", "zero rows", "dpre store + col sums",
         "small leaves", "xtr gemm", "xtr epilogue (adds)", "Z epilogue", "(samples)", "combine", "softmax stats", "x -> LDS", "wait: own adds performed", "barrier after the samples", "weights of a chunk", "scan to the next own sample", "scan after the last sample"]
for mode, mn in ((0, "theta"), (1, "z (reparam)"), (2, "z (score)")):
    row = [buf[mode * 32 + k] for k in range(32)]
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
