O=gpurun_out/r5q; mkdir -p $O
for i in 1 2 3; do python scripts/gpu_quick_bench.py 2>&1 | tail -n 3 | head -1; DIBS_NO_FLAG_JOIN=1 python scripts/gpu_quick_bench.py 2>&1 | tail -n 3 | head -1; done
bash scripts/gpu_timeline.sh r5q > $O/timeline.log 2>&1; head -9 gpurun_out/r5q_timeline.txt
