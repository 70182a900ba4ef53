"""Dev helper: C1 / C2 look-alikes: GPU warm solve time vs the compiled reference on the same host."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slam_plus_plus_amd import synth
from slam_plus_plus_amd.hip_solver import CLinearSolver_HIP
from oracle import oracle_lib as O
dev = torch.device("cuda:0")
for name, lam in [("C1 manhattan3500 SE(2)", synth.manhattan(3500)), ("C2 sphere2500 SE(3)", synth.sphere(50, 50)), ("10k-pose SE(3) chain", synth.pose_chain(n=10000))]:
    s = CLinearSolver_HIP()
    t0 = time.perf_counter(); s.SymbolicDecomposition_Blocky(lam); an = (time.perf_counter() - t0) * 1e3
    st = s.stats()
    vals = torch.from_numpy(lam.values).to(dev)
    bufs = [torch.from_numpy(lam.rhs).to(dev) for _ in range(21)]
    torch.cuda.synchronize()
    s.factor_solve_device(vals.data_ptr(), bufs[0].data_ptr())
    t0 = time.perf_counter()
    for b in bufs[1:]:
        s.factor_solve_device_async(vals.data_ptr(), b.data_ptr())
    s.sync(); dt = (time.perf_counter() - t0) / 20 * 1e3
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "p.bin"); lam.save(p)
        ref = {k: O.reference_solve(p, k, "-", reps=5)["times_ms"] for k in ("cholmod_auto", "uberblock")}
    print(f"{name}: n={lam.n_scalars} stages={st['n_stages']} tasks={st['n_tasks']} l_nnz={st['l_nnz']} analyze={an:.2f}ms gpu_warm={dt:.3f}ms "
          f"ref cholmod={np.median(ref['cholmod_auto']):.2f}ms ref native warm={np.median(ref['uberblock'][1:]):.2f}ms")
