"""Dev helper: what a drop-in caller pays -- oracle/_ref/dropin_driver time (the reference's CUberBlockMatrix through
include/slam/LinearSolver_HIP.h) on the bench's systems.  usage: time_dropin.py [c3 venice c1 c2]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from slam_plus_plus_amd import synth
CASES = {"c3": lambda: synth.pose_chain(n=100000), "venice": lambda: synth.ba(1000, 500000, mode="venice", seed=777),
         "c1": lambda: synth.manhattan(3500), "c2": lambda: synth.sphere(50, 50)}
for name in (sys.argv[1:] or ["c3", "venice"]):
    d = bench.dropin_leg(CASES[name](), reps=int(os.environ.get("REPS", "15")))
    print(name, json.dumps({k: d[k] for k in d if k.startswith("hip_") or k in ("ok", "rel_inf", "error")}), flush=True)
