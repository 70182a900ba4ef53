#!/bin/bash
# dev helper: per-kernel durations of the C3 solve (rocprofv3 kernel trace), run through gpurun
R=$PWD
mkdir -p $R/gpurun_out/prof_c3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/tools/time_c3.py "$@" > $R/gpurun_out/prof_c3/out.txt 2>&1
cd $R
tail -2 gpurun_out/prof_c3/out.txt
f=$(find gpurun_out/prof_c3 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:9.1f}us min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f}")
PY
# per-launch sequence of one step: the last 60 kernel launches in time order
t=$(find gpurun_out/prof_c3 -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
rows = rows[-52:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}us grid={r['Grid_Size_X']:>8s} wg={r['Workgroup_Size_X']:>4s} {r['Kernel_Name'][:60]}")
PY
rm -rf gpurun_out/prof_c3/*/*.csv 2>/dev/null; find gpurun_out/prof_c3 -name "*.csv" -size +1M -delete
