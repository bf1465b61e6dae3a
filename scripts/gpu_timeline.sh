#!/bin/bash
# usage (GPU box, repo root): scripts/gpu_timeline.sh <tag> [args of gpu_quick_bench.py]  -> gpurun_out/<tag>_timeline.txt
# rocprofv3 kernel trace of the headline window; prints the dispatch timeline (start offset, duration, gap to the previous kernel of the step)
# of three consecutive steps in the middle of the timed window.
set -e
TAG=${1:-tl}; shift || true
export TMPDIR=/tmp
rm -rf /tmp/tl_$TAG
rocprofv3 --kernel-trace -d /tmp/tl_$TAG -o tl --output-format csv -- python scripts/gpu_quick_bench.py "$@" > gpurun_out/${TAG}_timeline_run.log 2>&1
F=$(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1)
python - "$F" > gpurun_out/${TAG}_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0][:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")) for r in rows]
# the un-profiled timed repetitions come first: find edge-score launches and take steps in the 3rd repetition
idx = [i for i, e in enumerate(ev) if "k_edge_scores" in e[0]]
start = idx[5 + 2 * 20 + 10]
end = idx[5 + 2 * 20 + 13]
t0 = ev[start][1]
prev_end = t0
for name, s, e, q in ev[start:end]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:6.1f}  q{q}  {name}")
    prev_end = max(prev_end, e)
PY
cat gpurun_out/${TAG}_timeline.txt
