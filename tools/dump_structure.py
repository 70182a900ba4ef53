"""Dev helper: block structures of the bench's workloads as files for tools/micro/plan_bench.cpp."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_plus_plus_amd import synth
out = sys.argv[1]
os.makedirs(out, exist_ok=True)
for name, lam in (("c1", synth.manhattan(3500)), ("c2", synth.sphere(50, 50)), ("c3", synth.pose_chain(n=100000))):
    with open(os.path.join(out, name + ".bin"), "wb") as f:
        f.write(np.int64(lam.n_bcols).tobytes())
        f.write(np.ascontiguousarray(lam.cumsum, dtype=np.int64).tobytes())
        f.write(np.ascontiguousarray(lam.bcol_ptr, dtype=np.int64).tobytes())
        f.write(np.ascontiguousarray(lam.brow_idx, dtype=np.int32).tobytes())
