#!/bin/bash
# Round 3: pooling parity + ops table
out=gpurun_out/r3_pool; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_ps_roipool.py tests/test_roipool.py tests/test_tuned_vs_generic_fuzz.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
timeout -k 10 600 python bench_ops.py --iters 30 2>$out/ops.err | grep -E "roipool" | cut -c1-130 | tee $out/ops.jsonl
