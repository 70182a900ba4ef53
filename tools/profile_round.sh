#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun), ONE WORKLOAD PER FILE -- a per-kernel average over
# mixed workloads is no kernel's number: kernel-trace stats per leg, then, in separate passes without tracing, the HBM
# traffic counters (FETCH_SIZE and WRITE_SIZE do not fit one pass) and the matrix-core counters of the MFMA-bound legs.
# Usage: tools/profile_round.sh <tag>   (writes under gpurun_out/<tag>/; copy the summaries to profiles/)
set -u
TAG=${1:-r06}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# leg name -> bench.py arguments (the solve alone: no host-path / marginals / assembly legs launching the same kernels)
declare -A LEG
LEG[c3]="--workload c3 --steps 20 --warmup 3 --no-cpu-baseline --c3-solve-only"
LEG[ba_venice]="--workload ba --ba-legs venice --ba-steps 5 --no-cpu-baseline --ba-solve-only"
LEG[ba_band]="--workload ba --ba-legs band --ba-steps 5 --no-cpu-baseline --ba-solve-only"
LEG[ba_uniform]="--workload ba --ba-legs uniform --ba-steps 5 --no-cpu-baseline --ba-solve-only"
LEG[c2]="--workload small --small-configs C2 --no-cpu-baseline"
LEG[c1]="--workload small --small-configs C1 --no-cpu-baseline"
for L in c3 ba_venice ba_band ba_uniform c2 c1; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${L}_trace -- python3 $R/bench.py ${LEG[$L]} --full-json gpurun_out/$TAG/${L}_bench_full_under_rocprof.json > $OUT/${L}_bench_under_rocprof.json 2> $OUT/${L}_trace.err
  f=$(find $OUT/${L}_trace -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${L}_kernel_stats.csv
done
# K = 8 value sets of the C3 structure in one pass of launches: which kernels widen and which lengthen
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_batch8_trace -- python3 $R/tools/time_batch.py 8 10 > $OUT/c3_batch8_run.txt 2> $OUT/c3_batch8_trace.err
f=$(find $OUT/c3_batch8_trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/c3_batch8_kernel_stats.csv
for L in c3 ba_venice ba_band ba_uniform; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/${L}_pmc_$C -- python3 $R/bench.py ${LEG[$L]} --full-json gpurun_out/$TAG/scratch_full.json > $OUT/${L}_pmc_$C.json 2> $OUT/${L}_pmc_$C.err
  done
done
# matrix cores: instructions (MOPS) and busy cycles of the MFMA pipe next to the shader's busy cycles, per kernel
for L in ba_uniform ba_venice c2; do
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/${L}_pmc_mfma -- python3 $R/bench.py ${LEG[$L]} --full-json gpurun_out/$TAG/scratch_full.json > $OUT/${L}_pmc_mfma.json 2> $OUT/${L}_pmc_mfma.err
done
cd $R
for L in c3 ba_venice ba_band ba_uniform; do
  python3 tools/parse_pmc.py $OUT/${L}_pmc_FETCH_SIZE $OUT/${L}_pmc_WRITE_SIZE $OUT/${L}_traffic.json > $OUT/${L}_traffic.txt
done
for L in ba_uniform ba_venice c2; do
  python3 tools/parse_mfma.py $OUT/${L}_pmc_mfma $OUT/${L}_kernel_stats.csv $OUT/${L}_mfma.json > $OUT/${L}_mfma.txt
done
for L in c3 ba_venice ba_band ba_uniform c2 c1; do echo "== $L"; head -9 $OUT/${L}_kernel_stats.csv | cut -c1-160; done
# keep the merged artefacts small: the summaries stay, the per-dispatch CSVs go
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
