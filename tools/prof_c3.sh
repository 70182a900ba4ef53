#!/bin/bash
# dev helper: per-kernel durations of the C3 solve (rocprofv3 kernel trace), run through gpurun
R=$PWD
rm -rf $R/gpurun_out/prof_c3; mkdir -p $R/gpurun_out/prof_c3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/tools/time_c3.py > $R/gpurun_out/prof_c3/out.txt 2>&1
cd $R
grep "factor+solve" gpurun_out/prof_c3/out.txt
f=$(find gpurun_out/prof_c3 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:9.1f}us min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f}")
PY
find gpurun_out/prof_c3 -name "*.csv" -delete
