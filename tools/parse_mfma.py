"""Matrix-core counters per kernel from a rocprofv3 --pmc pass (SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES,
SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_INSTS_VALU), next to the kernel durations of the same leg's --kernel-trace --stats run.

usage: parse_mfma.py <dir of the counter pass> <kernel_stats.csv of the leg> <out.json>

Per kernel and launch: the counters as reported (summed over the chip's shader engines / XCDs by rocprofv3), and
  mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES  (share of the shader-busy cycles in which the MFMA pipe of
                     the sampled SQs was busy: the hardware's own utilisation figure),
  mfma_flops       = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 (the counter counts matrix operations in units of 512 flops per
                     the gfx94x derived-metric formula the ROCm 7.2 tables fall back to on gfx950),
  mfma_tflops      = mfma_flops / AverageNs of the kernel in the kernel-trace run, and its share of the 78.6 TFLOP/s fp64
                     matrix peak.
The two utilisation figures are independent: the first needs no unit assumption, the second is checked against the
algorithmic flop count bench.py prices the phase with (README of profiles/)."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import csv, glob, json, re, sys, collections

PEAK_TFLOPS = 78.6


def clean(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()


acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[clean(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
dur = {}
try:
    for r in csv.DictReader(open(sys.argv[2])):
        dur[clean(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]))
except OSError:
    pass
out = {}
for k, c in acc.items():
    per = {n: v[1] / max(v[0], 1) for n, v in c.items()}
    mops = per.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
    if mops <= 0:
        continue
    rec = {"launches": max(v[0] for v in c.values()), "per_launch": per,
           "mfma_busy_frac": per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(per.get("SQ_BUSY_CYCLES", 0.0), 1.0),
           "mfma_flops_per_launch": mops * 512.0}
    if k in dur:
        rec["avg_ns_kernel_trace"] = dur[k][0]
        rec["mfma_tflops"] = mops * 512.0 / dur[k][0] / 1e3
        rec["mfma_frac_of_fp64_peak"] = rec["mfma_tflops"] / PEAK_TFLOPS
    out[k] = rec
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_flops_per_launch"] * kv[1]["launches"]):
    print(f"{k[:64]:64s} n={v['launches']:5d} busy={v['mfma_busy_frac']:.3f} flops/launch={v['mfma_flops_per_launch']:.3e} "
          f"TF/s={v.get('mfma_tflops', float('nan')):.2f}")
