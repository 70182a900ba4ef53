#!/bin/bash
# dev helper (round 6): the first call of a process with the handle's streams brought up beside the analysis (default), and before slampp_hip_create returns
export SLAMPP_HIP_DEV=1
for v in "" "SLAMPP_HIP_DEV_NO_BRINGUP_THREAD=1" "SLAMPP_HIP_DEV_NO_BRINGUP_THREAD=1 SLAMPP_HIP_DEV_NO_WARMUP=1"; do
  echo "== ${v:-default}"
  for i in 1 2; do env $v python3 tools/first_launch.py 2>&1 | grep "first handle"; done
  env $v REPS=5 python3 tools/time_dropin.py c3 venice 2>&1 | grep -o '^[a-z0-9]* \|"hip_cold_ms": [0-9.]*\|"hip_warm_ms_median": [0-9.]*\|"ok": [a-z]*' | tr '\n' ' '; echo
  env $v REPS=5 python3 tools/time_dropin.py c3 2>&1 | grep -o '"hip_cold_ms": [0-9.]*'
done
echo "== cold path, default"
python3 tools/cold_path.py c3 venice 2>/dev/null | grep analyze_ms_cold
