#!/bin/bash
# usage (GPU box, repo root): scripts/rocprof_steady.sh <tag> [warmup] [steps]
# rocprofv3 kernel trace of bench.py; per-kernel average over the LAST <steps> launches only (steady state of the trajectory)
set -e
TAG=${1:-steady}; W=${2:-300}; K=${3:-100}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o $TAG -- python bench.py --config ${CONFIG:-headline} --no-cpu-baseline --reps 1 --min-seconds 0 --warmup $W --steps $K > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$f" $K > gpurun_out/${TAG}_steady_kernels.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1]))); K = int(sys.argv[2])
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
tot = 0.0
n_steps = sum(len(v) for k, v in by.items() if "k_edge_scores" in k)   # one edge-score launch per step
print(f"{'kernel':44s} {'launches':>8s} {'avg_us(last %d steps)' % K:>22s}")
# bench.py replays the timed window once more with per-kernel events: the last 2K launches cover both passes
for k, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1][-2 * K:])):
    v.sort()
    per_step = len(v) // (n_steps or 1) or 1
    w = v[-2 * K * per_step:] if len(v) >= 2 * K * per_step else v
    avg = sum(e - s for s, e in w) / len(w) / 1e3
    print(f"{k[:44]:44s} {len(v):8d} {avg:22.2f}" + (f"  x{per_step}/step" if per_step > 1 else ""))
    if len(v) >= K: tot += avg * per_step
print(f"sum over one step: {tot:.1f} us")
PY
cat gpurun_out/${TAG}_steady_kernels.txt
grep '"metric"' $OUT/bench.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench:', d['value'], 'steps/s', d['ms_per_step']*1e3, 'us/step')"
