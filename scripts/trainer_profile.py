"""where does a BaseTrainer step spend its HOST time?  cProfile over one epoch of the synthetic cfg2 run (bench.through_trainer)"""
import cProfile
import os
import pstats
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import bench  # noqa: E402
import mt3d_amd  # noqa: E402,F401
pr = cProfile.Profile()
pr.enable()
r = bench.through_trainer("cfg2", 2, "bf16", 3, 20)
pr.disable()
print(r)
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
