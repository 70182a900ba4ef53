#!/bin/bash
# dev helper: kernel times and HBM traffic of the Lambda assembly kernels (C3-shaped pose graph, C4-shaped BA system)
R=$PWD
OUT=$R/gpurun_out/pmc_asm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in c3 ba; do
  if [ $W = c3 ]; then PROG=$R/tools/time_assembly.py; else PROG=$R/tools/time_assembly_ba.py; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_trace -- python3 $PROG > $OUT/${W}_trace.txt 2>&1
  for SET in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/${W}_$SET -- python3 $PROG > $OUT/${W}_$SET.txt 2>&1
  done
  cd $R
  echo "== $W"
  f=$(find $OUT/${W}_trace -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "assemble" in r["Name"]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
  python3 tools/parse_pmc.py $OUT/${W}_FETCH_SIZE $OUT/${W}_WRITE_SIZE $OUT/${W}_traffic.json | grep assemble
  cp $f $OUT/${W}_kernel_stats.csv
  cd /tmp
done
find $OUT -name "*.csv" -size +1M -delete
