"""Dev helper: timeline of the dense factorization from a rocprofv3 kernel trace (which kernels overlap, gaps)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last factorization: from the last dense_pad_kernel (or schur_gather) onwards
idx = max(i for i, r in enumerate(rows) if "schur_gather" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
for r in rows[idx:idx + n]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    nm = r["Kernel_Name"].split("(")[0].replace("slampp::", "")[-28:]
    print(f"{nm:30s} q={r.get('Queue_Id','?'):>3s} grid={r.get('Grid_Size','?'):>8s} start={s/1e3:9.2f} end={e/1e3:9.2f} dur={(e-s)/1e3:8.2f}")
