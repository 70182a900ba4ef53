#!/bin/bash
# dev helper: what the leaf kernels of the C3 solve (factor_simt_kernel, backward_simt_kernel) are bound by, one value set (K = 1,
# tools/time_c3.py) and eight in one pass of launches (tools/time_batch.py 8): address path (TA), L1 (TCP) and issue counters,
# one rocprofv3 --pmc pass per set (no tracing).  Writes gpurun_out/pmc_leaf/summary.txt
R=$PWD
OUT=$R/gpurun_out/pmc_leaf
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for K in 1 8; do
  if [ $K = 1 ]; then CMD="python3 $R/tools/time_c3.py"; else CMD="python3 $R/tools/time_batch.py 8 10"; fi
  i=0
  for SET in "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum" \
             "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
             "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_BUSY_CYCLES SQ_WAVES"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/k${K}_p$i -- $CMD > $OUT/k${K}_p$i.txt 2>&1
  done
done
cd $R
python3 - <<'PY' > gpurun_out/pmc_leaf/summary.txt
import csv, glob, collections
for K in (1, 8):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(f"gpurun_out/pmc_leaf/k{K}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("slampp::", "")
            if "simt" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in sorted(agg):
        a = {c: agg[k][c] / max(cnt[k][c], 1) for c in agg[k]}
        print(f"K = {K}  {k}   (per launch, summed over the chip unless the counter is a maximum)")
        for c in sorted(a):
            print(f"    {c:42s} {a[c]:14.4g}")
PY
find $OUT -name "*.csv" -delete
