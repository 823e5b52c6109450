#!/bin/bash
# Regenerate tests/golden/ (stamped with the reference sources' sha256) and run the new GPU tests.
out=gpurun_out/golden; mkdir -p $out gpurun_out/r3_new
timeout -k 10 600 python tests/golden/make_golden.py $out > gpurun_out/r3_new/golden.log 2>&1; echo "make_golden rc=$?"; tail -2 gpurun_out/r3_new/golden.log
timeout -k 10 900 python -m pytest tests/test_region_filter.py tests/test_model_graph.py -m gpu -x -q > gpurun_out/r3_new/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3_new/pytest.log
