#!/bin/bash
# dev helper: where the waves of a BA leg's kernels wait (SQ counters, one rocprofv3 --pmc pass per set; no tracing)
R=$PWD
MODE=${1:-venice}
CMD="python3 $R/tools/time_ba.py 1000 500000 $MODE"
if [ "$MODE" = "c3" ]; then CMD="python3 $R/tools/time_c3.py"; fi
OUT=$R/gpurun_out/pmc_waits
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_WAVES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.txt 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pmc_waits/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("slampp::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    a = {c: agg[k][c] / max(cnt[k][c], 1) for c in agg[k]}
    wc = a.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0: continue
    print(k[:60])
    print("   per launch: " + "  ".join(f"{c[3:]}={a[c]:.3g}" for c in sorted(a)))
    print("   of wave cycles: " + "  ".join(f"{c[3:]}={a[c] / wc:.2f}" for c in sorted(a) if c != "SQ_WAVE_CYCLES" and ("WAIT" in c or "ACTIVE" in c or "CYCLES" in c)))
PY
find $OUT -name "*.csv" -size +2M -delete
