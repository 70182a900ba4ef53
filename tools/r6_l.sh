export SLAMPP_HIP_DEV=1
SLAMPP_HIP_PLAN_TIMING=1 REPS=2 SETTLE_MS=30 python3 tools/cold_path.py c3 2>&1 | grep "^\[shapes\]\|shapes   \|analyze_ms" | tail -8
SETTLE_MS=30 REPS=6 python3 tools/cold_path.py c3 2>/dev/null
timeout 900 python -m pytest tests/test_sparse_gpu.py tests/test_fullsize_reference_gpu.py -x -q -m gpu > gpurun_out/r6_t.txt 2>&1; grep -n "passed\|failed" gpurun_out/r6_t.txt | tail -2
python3 tools/time_c3.py 2>&1 | grep "factor+solve"
