"""Distribution of the factorisation sizes n = min(l + 1, d - l) that k_bge_chol sees at the headline config, per step, and the padding waste of its
size tiers (cost of a tier's problem ~ NMAX^3; n <= 4, 8, 12, 16 per lane, <= 20, 24 per lane pair, <= 28, 32 per quad)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dibs_amd import random
from dibs_amd.engine import Engine
cfg, x, mask = bench.make_workload("headline", 128)
eng = Engine(cfg); eng.set_data(x, mask); eng.init_particles(random.PRNGKey(1))
d, S, M = 50, 128, 128
t = 0
for cp in [int(a) for a in (sys.argv[1:] or ["5", "15", "24", "300"])]:
    eng.run(t, cp + 1 - t); t = cp + 1
    pm = eng.read("PARENT_MASKS").reshape(M, d, S)          # one u64 word per (m, j, s) at d <= 64
    l = np.zeros(pm.shape, np.int64)
    v = pm.copy()
    for _ in range(64):
        l += (v & np.uint64(1)).astype(np.int64); v >>= np.uint64(1)
    n = np.minimum(l + 1, d - l)
    n = n[l > 0]                                            # (l == 0: no factorisation, scored in the sampling kernel)
    hist = np.bincount(n, minlength=34)
    tiers = [4, 8, 12, 16, 20, 24, 28, 32]
    exact = float((hist * np.arange(len(hist)) ** 3).sum())
    padded = 0.0
    for lo, hi in zip([0] + tiers[:-1], tiers):
        padded += hist[lo + 1:hi + 1].sum() * hi ** 3
    print(f"step {cp}: {len(n)} problems of {M * d * S}; n histogram (1..{len(hist) - 1}): {hist[1:].tolist()}")
    print(f"   sum n^3 = {exact:.3e}, padded to the tier sizes {padded:.3e}: x{padded / exact:.2f}; with tiers of 2: x" +
          f"{sum(hist[k] * (k + (k & 1)) ** 3 for k in range(len(hist))) / exact:.2f}")
