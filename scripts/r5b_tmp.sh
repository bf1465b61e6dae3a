O=gpurun_out/r5k; mkdir -p $O
python scripts/gpu_quick_bench.py > $O/quick.txt 2>&1; tail -n 3 $O/quick.txt
bash scripts/gpu_timeline.sh r5k > $O/timeline.log 2>&1; head -8 gpurun_out/r5k_timeline.txt
