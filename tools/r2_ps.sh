#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_ps_roipool.py tests/test_tuned_vs_generic_fuzz.py tests/test_graph_capture.py -m gpu -q -x -k "ps or capture" > gpurun_out/pytest_ps.log 2>&1
echo "pytest rc=$?"; tail -n 3 gpurun_out/pytest_ps.log
D2T_PS_BWD=sorted timeout -k 10 900 python -m pytest tests/test_ps_roipool.py tests/test_tuned_vs_generic_fuzz.py -m gpu -q -x -k "ps" > gpurun_out/pytest_ps_sorted.log 2>&1
echo "pytest sorted rc=$?"; tail -n 3 gpurun_out/pytest_ps_sorted.log
D2T_PS_BWD=gemm timeout -k 10 200 python tools/ps_scan.py > gpurun_out/ps_scan_gemm.log 2>&1; D2T_PS_BWD=planes timeout -k 10 200 python tools/ps_scan.py > gpurun_out/ps_scan_planes.log 2>&1
D2T_PS_BWD=sorted timeout -k 10 200 python tools/ps_scan.py > gpurun_out/ps_scan_sorted.log 2>&1
paste gpurun_out/ps_scan_planes.log gpurun_out/ps_scan_sorted.log gpurun_out/ps_scan_gemm.log | cut -c1-120
