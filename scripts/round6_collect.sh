#!/bin/bash
# round 6 artefacts in one go (GPU box, repo root): headline full set, the precision trade at one commit, configs 2 / 3 / 5 with their late windows,
# posterior parity log, late-trajectory kernel breakdown, the mapped-memory benches.  -> gpurun_out/round6_*
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/collect_profiles.sh round6 > gpurun_out/round6_collect.log 2>&1
DIBS_ACYC_F32=1 python bench.py --no-cpu-baseline > gpurun_out/round6_bench_acyc_f32.json 2>/dev/null
DIBS_ACYC_BF16=1 python bench.py --no-cpu-baseline > gpurun_out/round6_bench_acyc_bf16.json 2>/dev/null
python bench.py --config 2 --no-cpu-baseline > gpurun_out/round6_bench_cfg2.json 2>/dev/null
python bench.py --config 3 --no-cpu-baseline --steady-t 1500 > gpurun_out/round6_bench_cfg3.json 2>/dev/null
python bench.py --config 4 --no-cpu-baseline > gpurun_out/round6_bench_cfg4.json 2>/dev/null
python bench.py --config 5 --no-cpu-baseline --min-seconds 3 > gpurun_out/round6_bench_cfg5.json 2>/dev/null
{ for a in "3 300" "3 1500" "5 300"; do python scripts/gpu_late_breakdown.py $a 5 2>&1 | grep -E "config|estimator"; done; } > gpurun_out/round6_late_breakdown.txt
DIBS_COMM=ipc python bench.py --gpus 2 > gpurun_out/round6_bench_ipc2.json 2>/dev/null
DIBS_COMM=ipc python bench.py --gpus 4 > gpurun_out/round6_bench_ipc4.json 2>/dev/null
python -m pytest tests/test_gpu_posterior.py -q -s 2>&1 | grep -vE "amdgpu.ids" > gpurun_out/round6_posterior_parity.txt
for f in gpurun_out/round6_bench*.json; do python - "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line); print(sys.argv[1], round(j["value"], 1), "steady", j.get("steady_state", {}).get("value"), "roof", (j.get("roofline") or {}).get("frac"))
PY
done
rm -rf gpurun_out/prof_* gpurun_out/pmc_*   # (raw rocprofv3 output: the summaries above are what is kept; gpurun merges back at most 64 MiB)
du -sh gpurun_out
