#!/bin/bash
# dev helper (round 4): the Venice-like and band legs' assembly kernels
for leg in venice band uniform; do
  echo "== $leg"
  bash tools/prof_ba.sh $leg 2>&1 | grep -E "ms/solve|schur_"
done
