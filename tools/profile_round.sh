#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun): kernel-trace stats for both workloads and,
# in separate passes, the HBM traffic counters.  Usage: tools/profile_round.sh <tag>   (writes under gpurun_out/<tag>/)
set -u
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace -- python3 $R/bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c3_trace.json 2> $OUT/c3_trace.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba_trace -- python3 $R/bench.py --workload ba --ba-steps 5 --no-cpu-baseline > $OUT/ba_trace.json 2> $OUT/ba_trace.err
# the counter passes run one BA visibility model at a time: the per-kernel averages of a mixed run would blend workloads
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/c3_pmc_$C -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c3_pmc_$C.json 2> $OUT/c3_pmc_$C.err
  for LEG in band uniform venice; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/ba_${LEG}_pmc_$C -- python3 $R/bench.py --workload ba --ba-legs $LEG --ba-steps 2 --no-cpu-baseline --ba-solve-only > $OUT/ba_${LEG}_pmc_$C.json 2> $OUT/ba_${LEG}_pmc_$C.err
  done
done
cd $R
python3 tools/parse_pmc.py $OUT/c3_pmc_FETCH_SIZE $OUT/c3_pmc_WRITE_SIZE $OUT/c3_traffic.json
python3 tools/parse_pmc.py $OUT/ba_band_pmc_FETCH_SIZE $OUT/ba_band_pmc_WRITE_SIZE $OUT/ba_traffic.json
python3 tools/parse_pmc.py $OUT/ba_uniform_pmc_FETCH_SIZE $OUT/ba_uniform_pmc_WRITE_SIZE $OUT/ba_uniform_traffic.json
python3 tools/parse_pmc.py $OUT/ba_venice_pmc_FETCH_SIZE $OUT/ba_venice_pmc_WRITE_SIZE $OUT/ba_venice_traffic.json
for f in $(find $OUT/c3_trace $OUT/ba_trace -name "*kernel_stats.csv"); do echo $f; head -8 $f | cut -c1-150; done
# keep the merged artefacts small: drop the per-dispatch CSVs of the counter passes
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*kernel_trace.csv" -size +4M -delete
