#!/bin/bash
# per-kernel durations of the last 3 steps of a config run to step t0: scripts/debug/late_kernels.sh <config> <t0>   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_late
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_late -o late -- python $GRAFT_REPO_ROOT/scripts/debug/late_trace.py $1 $2 > /tmp/prof_late.log 2>&1
python3 - <<'PY'
import csv, glob
fs = glob.glob("/tmp/prof_late/**/*kernel_trace.csv", recursive=True)
if not fs:
    print(open("/tmp/prof_late.log").read()[-2000:]); raise SystemExit(1)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_edge_scores" in n]
start = idx[-3]
agg = {}
for r in rows[start:]:
    n = r["Kernel_Name"][:70]; d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += d
for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:16]:
    print(f"{n:72s} {c:4d} x {t / c / 1000:10.1f} us")
PY
