#!/bin/bash
# dev helper: duration of every launch of one dense factorization (n = 6000), in launch order
R=$PWD
OUT=$R/gpurun_out/dense_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/tools/time_dense.py 1000 100000 > $OUT/run.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/dense_trace/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last factorization: find the last run of potrf_diag launches (94 of them)
idx = [i for i, r in enumerate(rows) if "potrf_diag_kernel" in r["Kernel_Name"]]
last = idx[-94:]
i0, i1 = last[0], last[-1]
seq = rows[i0:i1 + 3]
t_begin = int(seq[0]["Start_Timestamp"])
out = []
for r in seq:
    name = "potrf" if "potrf" in r["Kernel_Name"] else "trsm" if "trsm" in r["Kernel_Name"] else "syrk" if "syrk" in r["Kernel_Name"] else r["Kernel_Name"][:20]
    out.append((name, (int(r["Start_Timestamp"]) - t_begin) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // 256 if "Grid_Size_X" in r else -1))
tot = {}
for n, s, d, g in out:
    tot.setdefault(n, [0, 0.0]); tot[n][0] += 1; tot[n][1] += d
print("totals (us):", {k: (v[0], round(v[1], 1)) for k, v in tot.items()}, "span", round(out[-1][1] + out[-1][2], 1))
gaps = sum(out[i + 1][1] - (out[i][1] + out[i][2]) for i in range(len(out) - 1))
print("gaps between launches (us):", round(gaps, 1))
for p in range(0, 24):
    seg = [o for o in out if o[0] == "potrf"][4 * p:4 * p + 4]
    segt = [o for o in out if o[0] == "trsm"][4 * p:4 * p + 4]
    print(f"panel {p:2d}: potrf " + " ".join(f"{d:5.1f}({g:4d})" for _, _, d, g in seg) + "   trsm " + " ".join(f"{d:4.1f}" for _, _, d, g in segt))
PY
find $OUT -name "*.csv" -size +1M -delete
