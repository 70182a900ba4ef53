"""Dev tool: one line per leg of a bench.py output line (the last line of the file): ms per step and its phases."""
import json, sys

def walk(k, v, ind=0):
    if isinstance(v, dict):
        if "ms_per_step" in v and "phases_ms" in v:
            print(" " * ind + k, "ms", round(v["ms_per_step"], 3), {a: round(b, 3) for a, b in v["phases_ms"].items()})
        for kk, vv in v.items():
            if not kk.startswith("roofline"):
                walk(kk, vv, ind + 1)

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["metric"], "| value", round(d["value"], 1), "| ms/step", round(d["ms_per_step"], 4), "| roofline frac", d.get("roofline", {}).get("frac"))
for k, v in d.items():
    walk(k, v)
for k in ("small_configs", "dropin"):
    if k in d:
        print(k, json.dumps(d[k])[:600])
