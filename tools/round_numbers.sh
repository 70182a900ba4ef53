#!/bin/bash
python bench.py --full-json gpurun_out/r06_bench_full.json > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
echo "bench rc=$?"
wc -c gpurun_out/r06_bench.json
export SLAMPP_HIP_DEV=1
SETTLE_MS=30 REPS=5 python3 tools/cold_path.py c1 c2 c3 venice band c5 uniform 1kx1m 2>/dev/null > gpurun_out/r06_cold_path.txt
cat gpurun_out/r06_cold_path.txt
REPS=9 python3 tools/time_dropin.py c3 venice > gpurun_out/r06_dropin.txt 2>&1
cat gpurun_out/r06_dropin.txt
