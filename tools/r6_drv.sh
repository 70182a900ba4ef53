python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from slam_plus_plus_amd import synth
synth.pose_chain(n=100000).save("/tmp/c3.bin")
PY
eval $(python3 -c "
import sys, os
sys.path.insert(0, os.getcwd())
from oracle import oracle_lib as O
e = O.reference_env()
print('export LD_LIBRARY_PATH=' + e.get('LD_LIBRARY_PATH', ''))
print('export OMP_NUM_THREADS=' + e.get('OMP_NUM_THREADS', '16'))
")
for v in "A=1" "MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=2147483648" "A=1" "MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=2147483648"; do
  echo "===== $v"
  env $v SLAMPP_HIP_PLAN_TIMING=1 oracle/_ref/dropin_driver time /tmp/c3.bin 3 2>&1 | grep "^\[plan\] \(order\|symbolic\|schedule\|pairs\|graph\)\|^\[setup\] \(build_plan\|records\|packages\|panel\|shapes\|uploads\)\|header\]\|hip_cold" | sed 's/"reference_ms.*"hip_cold_ms"/"hip_cold_ms"/' | cut -c1-200 | tail -22
done
