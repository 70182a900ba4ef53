"""Dev helper: per-dispatch durations of one bench step from a rocprofv3 kernel trace CSV."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = collections.OrderedDict()
# take the last full step: find the last factor_subtree dispatch and print everything after it
idx = max(i for i, r in enumerate(rows) if "factor_subtree" in r["Kernel_Name"])
t_prev_end = None
for r in rows[idx:idx + 80]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - t_prev_end) / 1e3 if t_prev_end else 0
    t_prev_end = e
    nm = r["Kernel_Name"].split("(")[0][-40:]
    print(f"{nm:42s} grid={r.get('Grid_Size','?'):>8s} wg={r.get('Workgroup_Size','?'):>5s} dur={(e - s) / 1e3:8.2f}us gap={gap:7.2f}us lds={r.get('LDS_Block_Size','?')} vgpr={r.get('VGPR_Count','?')} sgpr={r.get('SGPR_Count','?')}")
