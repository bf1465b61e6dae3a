#!/bin/bash
# usage (GPU box, repo root): [CONFIG=headline|2|3|4|5] [QUICK=1] scripts/collect_profiles.sh <tag>   -> gpurun_out/<tag>_*
# (copy what should be judged into profiles/).  CONFIG selects bench.py's --config (default: the headline metric config).
# Everything is taken at the DRIVER's bench arguments (--steps 20 --warmup 5: steps t = 5..24 of the trajectory), one repetition of the
# timed window per rocprof run so that per-launch averages are averages over exactly that window (+ its profiled replays):
# 1. plain bench line (median of >= 7 repetitions)   2. rocprofv3 kernel stats, serial and production schedule
# 3. steady-state per-kernel averages (t >= 300; skipped with QUICK=1)   4. HBM traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate
# passes)   5. issue / occupancy / LDS counters of every kernel (LDS + steady MFMA passes skipped with QUICK=1)
set -e
TAG=${1:-r3}
CONFIG=${CONFIG:-headline}
ARGS="--config $CONFIG --no-cpu-baseline --reps 1 --min-seconds 0 --steps 20 --warmup 5"
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --config $CONFIG > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_err.log
# kernel stats twice: every kernel on one stream (DIBS_NO_ACYC_STREAM2: each duration is the kernel alone on the GPU -- what bench.py's
# `roofline` quotes) and the production schedule (the acyclicity kernel on the second stream: overlapping kernels share the GPU)
DIBS_NO_ACYC_STREAM2=1 bash scripts/rocprof_bench.sh ${TAG} $ARGS > /dev/null
bash scripts/rocprof_bench.sh ${TAG}_concurrent $ARGS > /dev/null
if [ -z "$QUICK" ]; then
  DIBS_NO_ACYC_STREAM2=1 CONFIG=$CONFIG bash scripts/rocprof_steady.sh ${TAG} 300 100 > /dev/null
fi
LASTN=0 bash scripts/rocprof_pmc.sh ${TAG}_fetch FETCH_SIZE -- python bench.py $ARGS > gpurun_out/${TAG}_pmc_fetch.txt 2>&1
LASTN=0 bash scripts/rocprof_pmc.sh ${TAG}_write WRITE_SIZE -- python bench.py $ARGS > gpurun_out/${TAG}_pmc_write.txt 2>&1
LASTN=0 bash scripts/rocprof_pmc.sh ${TAG}_issue "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" -- python bench.py $ARGS > gpurun_out/${TAG}_pmc_issue_window.txt 2>&1
if [ -z "$QUICK" ]; then
  LASTN=0 bash scripts/rocprof_pmc.sh ${TAG}_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM" -- python bench.py $ARGS > gpurun_out/${TAG}_pmc_lds_window.txt 2>&1
  LASTN=40 bash scripts/rocprof_pmc.sh ${TAG}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" -- python bench.py --config $CONFIG --no-cpu-baseline --reps 1 --min-seconds 0 --steps 30 --warmup 300 > gpurun_out/${TAG}_pmc_mfma_steady.txt 2>&1
fi
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
# rocprof kernel name (prefix) -> the engine's timer name (bench.py's roofline looks its dominant kernel up by that name)
prefixes = [("void k_acyc_hf", "acyc"), ("void k_acyc_bf", "acyc"), ("void k_acyc<", "acyc"), ("k_acyc_reduce", "acyc_reduce"), ("void k_bge_sample", "bge_nodes"),
            ("void k_bge_chol", "bge_big"), ("k_particle_grad", "particle_grad"), ("k_kmat", "kmat"), ("void k_phi_update", "phi_update"),
            ("void k_edge_scores", "edge"), ("k_edge_scores_p", "edge"), ("void k_lin_logprobs", "lin_logprobs"), ("void k_lin_grad", "lin_grad"), ("void k_nn_logprobs", "nn_logprobs"), ("k_nn_tables_hf", "nn_tables"),
            ("void k_nn_grad", "nn_grad"), ("k_nng_logprobs", "nng_logprobs"), ("k_nng_grad", "nng_grad"), ("void k_bge_soft", "bge_soft")]
res = {}
for k in f:
    if k not in w: continue
    short = next((s for p, s in prefixes if k.startswith(p)), None)
    if short is None or short in res: continue
    res[short] = {"rocprof_kernel": k, "FETCH_SIZE_KiB_per_launch": f[k], "WRITE_SIZE_KiB_per_launch": w[k],
                  "hbm_bytes_per_launch": (2.0 * f[k] + w[k]) * 1024.0,
                  "note": "avg over the launches of bench.py --reps 1 --steps 20 --warmup 5 (warm-up, timed window and its replays); FETCH_SIZE doubled (gfx950 "
                          "correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported"}
if "nn_logprobs" in res and "nn_grad" in res:   # the engine's timers nn_theta / nn_z each bracket one k_nn_logprobs + one k_nn_grad launch
    both = {"rocprof_kernel": "k_nn_logprobs + k_nn_grad", "hbm_bytes_per_launch": res["nn_logprobs"]["hbm_bytes_per_launch"] + res["nn_grad"]["hbm_bytes_per_launch"],
            "note": "sum of the two kernels one estimator launches (averages over both estimators)"}
    res["nn_theta"], res["nn_z"] = both, both
json.dump(res, open(f"gpurun_out/{tag}_pmc_hbm.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
