for nw in 2 4 2 4; do echo NW=$nw; DIBS_PHI_NW=$nw python scripts/gpu_quick_bench.py 2>&1 | tail -n 3 | head -2 | cut -c1-200; done
