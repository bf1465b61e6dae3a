#!/bin/bash
# Runs tests/golden/make_joint_golden.py on the GPU box's HOST cores (128 threads; ~70-100 min): oracle trajectories of configs 3 / 5.
# One .npz per (config, build, seed) lands in gpurun_out/joint_golden/ (merged back by gpurun); merge with
#   python tests/golden/make_joint_golden.py merge gpurun_out/joint_golden
OUT=gpurun_out/joint_golden
mkdir -p $OUT
nproc > $OUT/nproc.txt
G=tests/golden/make_joint_golden.py
( python $G config3 f64 0:8 $OUT 64 > $OUT/log_c3_f64.txt 2>&1 ) &
( python $G config5 f64 0:4 $OUT 32 > $OUT/log_c5_f64.txt 2>&1 ) &
( python $G config3 f32 0:2 $OUT 32 > $OUT/log_c3_f32.txt 2>&1 ; python $G config5 f32 0:1 $OUT 32 > $OUT/log_c5_f32.txt 2>&1 ) &
wait
tail -n 3 $OUT/log_*.txt
