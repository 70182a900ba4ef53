#!/bin/bash
# the BA counter passes of tools/profile_round.sh alone (usage: tools/pmc_ba_round.sh <tag>)
set -u
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  for LEG in band uniform venice; do
    rm -rf $OUT/ba_${LEG}_pmc_$C
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/ba_${LEG}_pmc_$C -- python3 $R/bench.py --workload ba --ba-legs $LEG --ba-steps 2 --no-cpu-baseline --ba-solve-only > $OUT/ba_${LEG}_pmc_$C.json 2> $OUT/ba_${LEG}_pmc_$C.err
  done
done
cd $R
python3 tools/parse_pmc.py $OUT/ba_band_pmc_FETCH_SIZE $OUT/ba_band_pmc_WRITE_SIZE $OUT/ba_traffic.json | grep "schur"
python3 tools/parse_pmc.py $OUT/ba_uniform_pmc_FETCH_SIZE $OUT/ba_uniform_pmc_WRITE_SIZE $OUT/ba_uniform_traffic.json | grep "schur"
python3 tools/parse_pmc.py $OUT/ba_venice_pmc_FETCH_SIZE $OUT/ba_venice_pmc_WRITE_SIZE $OUT/ba_venice_traffic.json | grep "schur"
find $OUT -name "*counter_collection.csv" -size +2M -delete
