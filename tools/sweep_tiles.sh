#!/bin/bash
# dev helper: tile size of the landmark-major Schur assembly at C4
for p in 16 32 64 128; do
  echo "== SLAMPP_TILE_POINTS=$p"
  SLAMPP_TILE_POINTS=$p SLAMPP_HIP_PLAN_TIMING=1 python tools/time_ba.py 1000 500000 ${1:-band} 2>&1 | grep "tiles\|ms/solve"
done
