#!/bin/bash
for cg in 4 2 1; do echo "CG=$cg"; D2T_SAT_CG=$cg timeout -k 10 200 python bench_ops.py --iters 30 2>/dev/null | grep '"roipool"' | grep fwd | cut -c1-120; done
