"""Dev helper: option schur_incremental at C4 (update of the reduced system against its full rebuild)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from slam_plus_plus_amd import synth
dev = torch.device("cuda:0")
for mode in ("band", "uniform"):
    lam = synth.ba(1000, 500000, k=4, mode=mode)
    for share in (0.001, 0.004, 0.01, 0.03, 0.1):
        for always in (False, True):
            r = bench.incremental_leg(lam, dev, 0, torch, share=share, reps=3, always=always)
            r.pop("note", None)
            print(mode, share, "always" if always else "when it pays", json.dumps(r), flush=True)
