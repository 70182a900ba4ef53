"""Turns rocprofv3 --pmc counter_collection CSVs (one pass per counter) into per-kernel HBM traffic.

usage: parse_pmc.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <out.json>

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming
read, so it is doubled; WRITE_SIZE is exact for streaming stores.  The guide calibrates the factor 2
on 16-B-per-lane streams; these kernels read 8 B per lane, so both the raw and the corrected figure
are kept."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import csv, glob, json, re, sys, collections


def load(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            a = acc[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    nf, vf = fetch.get(k, [0, 0.0])
    nw, vw = write.get(k, [0, 0.0])
    n = max(nf, nw, 1)
    out[k] = {"launches": n, "fetch_KiB_per_launch_raw": vf / max(nf, 1), "write_KiB_per_launch": vw / max(nw, 1),
              "hbm_bytes_per_launch_raw": (vf / max(nf, 1) + vw / max(nw, 1)) * 1024,
              "hbm_bytes_per_launch_corrected": (2 * vf / max(nf, 1) + vw / max(nw, 1)) * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{k[:60]:60s} n={v['launches']:5d} raw={v['hbm_bytes_per_launch_raw']/1e6:10.3f} MB corrected={v['hbm_bytes_per_launch_corrected']/1e6:10.3f} MB")
