#!/bin/bash
# dev helper (round 6): the leaf substitution fetched by the whole wave against a lane fetching its own blocks
R=$PWD
export SLAMPP_HIP_DEV=1
timeout 900 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu 2>&1 | tail -3
for st in 1 0; do
  echo "== SLAMPP_HIP_DEV_SIMT_BWD_STAGED=$st"
  export SLAMPP_HIP_DEV_SIMT_BWD_STAGED=$st
  bash tools/prof_c3.sh 2>&1 | grep -i "factor+solve\|simt\|backward" | head -8
  timeout 300 python3 tools/time_batch.py 2>&1 | tail -4
done
