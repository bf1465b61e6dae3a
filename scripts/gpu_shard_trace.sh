#!/bin/bash
# usage (GPU box, repo root): scripts/gpu_shard_trace.sh <R> <overlapped 0|1> <tag>   -> gpurun_out/<tag>_shard_trace.txt
# rocprofv3 kernel + memory-copy trace of the in-engine sharded loop of ONE rank of an R-way run (loopback communicator, scripts/gpu_shard_native.py):
# the dispatch timeline (start offset, duration, queue / stream, gap to the end of the previous entry) of three consecutive steady-state steps.
set -e
R=${1:-4}; OV=${2:-1}; TAG=${3:-shard}
export TMPDIR=/tmp
rm -rf /tmp/st_$TAG
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/st_$TAG -o st --output-format csv -- python scripts/gpu_shard_native.py $R $OV > gpurun_out/${TAG}_shard_trace_run.log 2>&1
K=$(find /tmp/st_$TAG -name "*kernel_trace.csv" | head -1)
C=$(find /tmp/st_$TAG -name "*memory_copy_trace.csv" | head -1)
python - "$K" "$C" > gpurun_out/${TAG}_shard_trace.txt <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44], "q" + r.get("Queue_Id", "?")))
if len(sys.argv) > 2 and sys.argv[2]:
    try:
        for r in csv.DictReader(open(sys.argv[2])):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), "copy"))
    except Exception as ex:
        print("no copy trace:", ex)
ev.sort()
idx = [i for i, e in enumerate(ev) if "k_edge_scores" in e[2]]
s, e_ = idx[len(idx) // 2], idx[len(idx) // 2 + 3]
t0, prev = ev[s][0], ev[s][0]
for a, b, name, q in ev[s:e_]:
    print(f"{(a - t0) / 1e3:9.1f} us  +{(b - a) / 1e3:7.1f} us  gap {(a - prev) / 1e3:7.1f}  {q:>5s}  {name}")
    prev = max(prev, b)
PY
cat gpurun_out/${TAG}_shard_trace.txt
