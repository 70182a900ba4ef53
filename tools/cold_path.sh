#!/bin/bash
# dev helper: where the first (cold) call of the header-side solver spends its time, at C3 and C4 -- through gpurun
R=$PWD
mkdir -p $R/gpurun_out
python3 - <<'PY'
from slam_plus_plus_amd import synth
synth.pose_chain().save("/tmp/c3.bin")
synth.ba(1000, 500_000, k=4, mode="band").save("/tmp/c4.bin")
PY
for f in c3 c4; do
  OMP_NUM_THREADS=16 SLAMPP_HIP_PLAN_TIMING=1 oracle/_ref/dropin_driver time /tmp/$f.bin 3 2>&1 | grep -v "^\[setup\] stage\|dense top" | tee $R/gpurun_out/cold_$f.txt
done
