for ta in 16 8; do echo TA=$ta; DIBS_PHI_TA=$ta python scripts/gpu_quick_bench.py 2>&1 | tail -n 3 | head -2; DIBS_PHI_TA=$ta python scripts/gpu_quick_bench.py 2>&1 | tail -n 3 | head -1; done
