#!/bin/bash
# dev helper: a few SQ / cache counters for the kernels of the C3 step (separate passes, no tracing)
R=$PWD
OUT=$R/gpurun_out/pmc_ba
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $SET | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/$tag -- python3 $R/tools/time_ba.py 1000 500000 band > $OUT/$tag.txt 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/pmc_ba/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("slampp::", "").strip()[:44]
        a = acc[name][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(acc):
    if not any(x in k for x in ("schur_run", "schur_tile_k", "schur_gather")):
        continue
    print(k)
    for c, (n, v) in sorted(acc[k].items()):
        print(f"    {c:32s} per launch {v / n:14.1f}")
PY
find $OUT -name "*.csv" -size +1M -delete
