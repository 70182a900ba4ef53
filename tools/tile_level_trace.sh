#!/bin/bash
# dev helper: every launch of the reduced camera system's tile levels (Venice-like leg), in launch order, last solve
R=$PWD
OUT=$R/gpurun_out/tile_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tools/time_ba.py 1000 500000 ${1:-venice} > $OUT/run.txt 2>&1
cd $R
grep -E "ms/solve|analyze" $OUT/run.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tile_trace/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last solve: from the last schur_point_inverse / first assembly kernel to the end
starts = [i for i, r in enumerate(rows) if "dense_assemble_kernel" in r["Kernel_Name"]]
i0 = starts[-1]
seq = rows[i0:]
t0 = int(seq[0]["Start_Timestamp"])
prev_end = t0
for r in seq:
    n = r["Kernel_Name"]
    short = n.split("(")[0].replace("slampp::", "").replace("void ", "")[:34]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) if "Grid_Size_X" in r else -1
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} gap {(s - prev_end) / 1e3:5.1f}  {short:34s} wgs {wg}")
    prev_end = e
PY
find $OUT -name "*.csv" -size +1M -delete
