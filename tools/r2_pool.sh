#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_roipool.py tests/test_ps_roipool.py tests/test_tuned_vs_generic_fuzz.py tests/test_graph_capture.py -m gpu -q -x > gpurun_out/pytest_pool.log 2>&1
echo "pytest rc=$?"; tail -n 3 gpurun_out/pytest_pool.log
timeout -k 10 200 python bench_ops.py --iters 30 2>/dev/null | grep -E 'roipool' | grep bwd | cut -c1-100
