#!/bin/bash
# usage (GPU box, repo root): scripts/collect_profiles.sh <tag>   -> gpurun_out/<tag>_*  (copy what should be judged into profiles/)
# Everything is taken at the DRIVER's bench arguments (--steps 20 --warmup 5: steps t = 5..24 of the trajectory), one repetition of the
# timed window per rocprof run so that per-launch averages are averages over exactly that window (+ its profiled replay):
# 1. plain bench line (7 repetitions, median)   2. rocprofv3 kernel stats   3. steady-state per-kernel averages (t >= 300)
# 4. HBM traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate passes)   5. issue / occupancy / LDS counters of every kernel
set -e
TAG=${1:-r2}
ARGS="--no-cpu-baseline --reps 1 --steps 20 --warmup 5"
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_err.log
# kernel stats twice: every kernel on one stream (DIBS_NO_ACYC_STREAM2: each duration is the kernel alone on the GPU -- what bench.py's
# `roofline` quotes) and the production schedule (the acyclicity kernel on the second stream: overlapping kernels share the GPU)
DIBS_NO_ACYC_STREAM2=1 bash scripts/rocprof_bench.sh ${TAG} --reps 1 --steps 20 --warmup 5 > /dev/null
bash scripts/rocprof_bench.sh ${TAG}_concurrent --reps 1 --steps 20 --warmup 5 > /dev/null
DIBS_NO_ACYC_STREAM2=1 bash scripts/rocprof_steady.sh ${TAG} 300 100 > /dev/null
LASTN=0 bash scripts/rocprof_pmc.sh ${TAG}_fetch FETCH_SIZE -- python bench.py $ARGS > gpurun_out/${TAG}_pmc_fetch.txt 2>&1
LASTN=0 bash scripts/rocprof_pmc.sh ${TAG}_write WRITE_SIZE -- python bench.py $ARGS > gpurun_out/${TAG}_pmc_write.txt 2>&1
LASTN=0 bash scripts/rocprof_pmc.sh ${TAG}_issue "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" -- python bench.py $ARGS > gpurun_out/${TAG}_pmc_issue_window.txt 2>&1
LASTN=0 bash scripts/rocprof_pmc.sh ${TAG}_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM" -- python bench.py $ARGS > gpurun_out/${TAG}_pmc_lds_window.txt 2>&1
LASTN=40 bash scripts/rocprof_pmc.sh ${TAG}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" -- python bench.py --no-cpu-baseline --reps 1 --steps 30 --warmup 300 > gpurun_out/${TAG}_pmc_mfma_steady.txt 2>&1
python - ${TAG} <<'PY'
import ast, json, sys
tag = sys.argv[1]
def parse(path, key):
    out = {}
    for line in open(path):
        if "{" not in line: continue
        name, rest = line.split("{", 1)
        d = ast.literal_eval("{" + rest.strip())
        if key in d: out[name.strip()] = float(d[key])
    return out
f, w = parse(f"gpurun_out/{tag}_pmc_fetch.txt", "FETCH_SIZE"), parse(f"gpurun_out/{tag}_pmc_write.txt", "WRITE_SIZE")
names = {"void k_acyc_bf<true>": "acyc", "void k_bge_sample<4, true>": "bge_nodes", "void k_bge_chol<true, false>": "bge_big", "k_lik_weights_score": "lik_weights",
         "k_kmat": "kmat", "void k_phi_update<8>": "phi_update", "k_edge_scores": "edge", "k_zgrad": "zgrad", "k_wtotal": "wtotal"}
res = {}
for k, short in names.items():
    if k in f and k in w:
        res[short] = {"rocprof_kernel": k, "FETCH_SIZE_KiB_per_launch": f[k], "WRITE_SIZE_KiB_per_launch": w[k],
                      "hbm_bytes_per_launch": (2.0 * f[k] + w[k]) * 1024.0,
                      "note": "avg over the launches of bench.py --reps 1 --steps 20 --warmup 5 (warm-up, timed window and its replay); FETCH_SIZE doubled (gfx950 "
                              "correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported"}
json.dump(res, open(f"gpurun_out/{tag}_pmc_hbm.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
