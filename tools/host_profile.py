#!/usr/bin/env python3
"""cProfile of the host side of a small-batch training step (the reference's own --batch 6 is launch-bound): where the Python time goes."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "60", "--warmup", "5", "--batch", os.environ.get("BATCH", "6"), "--no-cpu-baseline"]
import runpy

pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
