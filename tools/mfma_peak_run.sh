R=$PWD
mkdir -p $R/gpurun_out/r6c
./tools/bin/mfma_f64_peak > $R/gpurun_out/r6c/mfma_f64_peak.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/r6c/mfma_pmc -- $R/tools/bin/mfma_f64_peak > $R/gpurun_out/r6c/mfma_under_pmc.txt 2> $R/gpurun_out/r6c/mfma_pmc.err
cd $R
python3 - <<'PY' > gpurun_out/r6c/mfma_pmc_summary.txt
import csv, glob, collections
rows = collections.OrderedDict()
for f in glob.glob("gpurun_out/r6c/mfma_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mfma_loop" not in r["Kernel_Name"]:
            continue
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"][:60], r.get("Grid_Size", ""), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")))
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, k, g, l), c in sorted(rows.items()):
    busy = c.get("SQ_BUSY_CYCLES", 0); mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    print(f"dispatch {d:4d} grid {g:>8} lds {l:>7} {k}: MFMA_BUSY {mf:.4g} / SQ_BUSY {busy:.4g} = {mf / busy if busy else 0:.3f}; MOPS_F64 {c.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0):.4g}; WAVE_CYCLES {c.get('SQ_WAVE_CYCLES', 0):.4g}")
PY
find gpurun_out/r6c -name "*counter_collection.csv" -delete; find gpurun_out/r6c -name "*agent_info.csv" -delete
