export SLAMPP_HIP_DEV=1
for v in "A=1" "SLAMPP_HIP_DEV_NO_HUGE_PAGES=1" "A=1"; do
  echo "== $v: a fresh process per system"
  for s in ba:2000:2000000:band ba:1000:500000:venice ba:1000:500000:uniform chain:100000 ba:1000:1000000:band; do env $v python3 -m bench_legs.cold $s 2>/dev/null; done
done
echo "== one process"
SETTLE_MS=30 REPS=4 python3 tools/cold_path.py c1 c2 c3 venice band c5 uniform 2>/dev/null
REPS=5 python3 tools/time_dropin.py c3 venice 2>&1 | grep -o '^[a-z0-9]* \|"hip_cold_ms": [0-9.]*\|"hip_warm_ms_median": [0-9.]*' | tr '\n' ' '; echo
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_schur_gpu.py tests/test_errors_gpu.py tests/test_dropin_gpu.py -x -q -m gpu 2>&1 | tail -2
