#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_roipool.py tests/test_tuned_vs_generic_fuzz.py tests/test_graph_capture.py tests/test_model_graph.py -m gpu -q -x > gpurun_out/pytest_sat.log 2>&1
echo "pytest rc=$?"; tail -n 3 gpurun_out/pytest_sat.log
for m in 0 1; do echo "== INTERLEAVED $m"; D2T_SAT_INTERLEAVED=$m timeout -k 10 200 python bench_ops.py --iters 40 2>/dev/null | grep '"roipool"' | grep fwd | cut -c1-100; done
