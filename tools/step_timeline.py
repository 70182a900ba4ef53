"""Dev tool: the dispatches of the last solve step in a rocprofv3 kernel_trace.csv, in order: start, duration, grid, kernel.
Usage: python tools/step_timeline.py <dir with *kernel_trace.csv> <name of the step's first kernel (substring)> [n_rows]"""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import csv, glob, sys

def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    first = sys.argv[2]
    n_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    i0 = starts[-2] if len(starts) > 1 else starts[-1]
    i1 = starts[-1] if len(starts) > 1 else len(rows)
    t0 = int(rows[i0]["Start_Timestamp"])
    print("step: %d dispatches, %.1f us" % (i1 - i0, (int(rows[i1 - 1]["End_Timestamp"]) - t0) * 1e-3))
    for r in rows[i0:i1][:n_rows]:
        print("%8.1f us  +%6.1f us  grid %-8s wg %-4s %s" % ((int(r["Start_Timestamp"]) - t0) * 1e-3,
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"), r["Kernel_Name"][:80]))

if __name__ == "__main__":
    main()
