#!/bin/bash
# usage (GPU box, repo root): scripts/collect_profiles.sh <tag>   -> gpurun_out/<tag>_*  (copy what should be judged into profiles/)
# 1. plain bench line  2. rocprofv3 kernel stats of the default bench  3. steady-state per-kernel averages
# 4. HBM traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate passes)  5. MFMA / VALU counters of the dominant kernels
set -e
TAG=${1:-r1}
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_err.log
bash scripts/rocprof_bench.sh ${TAG} > /dev/null
bash scripts/rocprof_steady.sh ${TAG} 300 100 > /dev/null
LASTN=0 bash scripts/rocprof_pmc.sh ${TAG}_fetch FETCH_SIZE -- python bench.py --no-cpu-baseline --steps 50 --warmup 20 > gpurun_out/${TAG}_pmc_fetch.txt 2>&1
LASTN=0 bash scripts/rocprof_pmc.sh ${TAG}_write WRITE_SIZE -- python bench.py --no-cpu-baseline --steps 50 --warmup 20 > gpurun_out/${TAG}_pmc_write.txt 2>&1
LASTN=40 bash scripts/rocprof_pmc.sh ${TAG}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" -- python bench.py --no-cpu-baseline --steps 30 --warmup 300 > gpurun_out/${TAG}_pmc_mfma_steady.txt 2>&1
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
names = {"void k_acyc<4, true>": "acyc", "void k_bge_nodes<4, true>": "bge_nodes", "void k_bge_big<true>": "bge_big", "k_lik_weights_score": "lik_weights",
         "k_kmat": "kmat", "void k_phi_update<16>": "phi_update", "k_edge_scores": "edge", "k_zgrad": "zgrad", "k_wtotal": "wtotal"}
res = {}
for k, short in names.items():
    if k in f and k in w:
        res[short] = {"rocprof_kernel": k, "FETCH_SIZE_KiB_per_launch": f[k], "WRITE_SIZE_KiB_per_launch": w[k],
                      "hbm_bytes_per_launch": (2.0 * f[k] + w[k]) * 1024.0,
                      "note": "avg over the launches of bench.py --steps 50 --warmup 20; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported"}
json.dump(res, open(f"gpurun_out/{tag}_pmc_hbm.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
