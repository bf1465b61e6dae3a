#!/bin/bash
# Round 5, second batch on the GPU box's HOST cores (256 hardware threads): the seeds the container's 8 cores cannot finish in time.
#   config 3 f64 seed 7, config 5 f64 seeds 1..3 -- 64 threads each.  Output: gpurun_out/joint_golden_box2/ (one .npz per seed).
OUT=gpurun_out/joint_golden_box2
mkdir -p $OUT
nproc > $OUT/nproc.txt
G=tests/golden/make_joint_golden.py
( python $G config3 f64 7:8 $OUT 64 > $OUT/log_c3_f64_s7.txt 2>&1 ) &
( python $G config5 f64 1:2 $OUT 64 > $OUT/log_c5_f64_s1.txt 2>&1 ) &
( python $G config5 f64 2:3 $OUT 64 > $OUT/log_c5_f64_s2.txt 2>&1 ) &
( python $G config5 f64 3:4 $OUT 64 > $OUT/log_c5_f64_s3.txt 2>&1 ) &
wait
tail -n 2 $OUT/log_*.txt
