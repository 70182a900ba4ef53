"""Dev tool: reads a rocprofv3 kernel_trace.csv of a run of the dense (uniform-visibility) BA leg and prints, for the last
dense factorization in it, the dispatches of the chain (potrf_diag_kernel / trsm_kernel) and of the look-ahead stream
(syrk_kernel) on one time axis: start, duration, queue.  Usage: python tools/dense_timeline.py <dir with *kernel_trace.csv> [n_rows]"""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import csv, glob, sys

def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = ("potrf_diag_kernel", "trsm_kernel", "syrk_kernel")
    dense = [r for r in rows if any(n in r["Kernel_Name"] for n in names)]
    # the last factorization: walk back from the end until a gap of more than 200 us between dense kernels
    i = len(dense) - 1
    while i > 0 and int(dense[i]["Start_Timestamp"]) - int(dense[i - 1]["End_Timestamp"]) < 200000:
        i -= 1
    last = dense[i:]
    t0 = int(last[0]["Start_Timestamp"])
    print("factorization: %d dispatches, %.3f ms" % (len(last), (max(int(r["End_Timestamp"]) for r in last) - t0) * 1e-6))
    # per group of 8 chain steps: time in potrf launches, trsm launches, other
    chain = [r for r in last if r.get("Queue_Id") == last[0].get("Queue_Id")]
    step = 0
    acc = {"potrf_diag_kernel": 0.0, "trsm_kernel": 0.0, "syrk_kernel": 0.0}
    t_group = int(chain[0]["Start_Timestamp"])
    for r in chain:
        n = [n for n in names if n in r["Kernel_Name"]][0]
        if n == "potrf_diag_kernel":
            if step and step % 8 == 0:
                print("steps %3d..%3d: %7.1f us  (potrf launches %6.1f, trsm %6.1f, syrk %6.1f)" % (step - 8, step - 1,
                    (int(r["Start_Timestamp"]) - t_group) * 1e-3, acc["potrf_diag_kernel"], acc["trsm_kernel"], acc["syrk_kernel"]))
                acc = {k: 0.0 for k in acc}
                t_group = int(r["Start_Timestamp"])
            step += 1
        acc[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    for r in last[:n_rows]:
        n = [n for n in names if n in r["Kernel_Name"]][0]
        print("%9.1f us  +%7.1f us  q%-3s grid %-7s %s" % ((int(r["Start_Timestamp"]) - t0) * 1e-3,
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3, r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), n))

if __name__ == "__main__":
    main()
