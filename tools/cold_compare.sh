#!/bin/bash
# dev helper: first call of the header-side solver in a fresh process, with and without the staging thread, several times
R=$PWD
python3 - <<'PY'
from slam_plus_plus_amd import synth
synth.pose_chain().save("/tmp/c3.bin")
PY
for i in 1 2 3 4; do
  for v in ahead noahead; do
    if [ $v = noahead ]; then export SLAMPP_HIP_NO_STAGING_AHEAD=1; else unset SLAMPP_HIP_NO_STAGING_AHEAD; fi
    OMP_NUM_THREADS=16 oracle/_ref/dropin_driver time /tmp/c3.bin 2 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', 'cold', d['hip_cold_ms'], 'init', d['hip_runtime_init_ms'], 'warm', d['hip_warm_ms_median'])"
  done
done
