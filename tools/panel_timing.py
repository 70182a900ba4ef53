"""Dev helper: in-kernel clock samples of the panel kernel on the C4-like band system (SLAMPP_HIP_STAGE_TIMING=1)."""
import os as _os; _os.environ.setdefault("SLAMPP_HIP_DEV", "1")  # development options and knobs are refused without it (csrc/plan.h)
import sys, os
os.environ["SLAMPP_HIP_STAGE_TIMING"] = "1"
sys.argv = [sys.argv[0], "1000", "subtree_size=4"] + sys.argv[1:]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_band_system.py")).read())
