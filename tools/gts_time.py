import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(tempfile.mkdtemp())
import numpy as np
from blackbird_amd import Blackbird, Connect4
cfg = {"blocks": 4, "filters": 16, "eval": {"dense": 16}, "hasTeacher": False,
       "policy": {"dirichlet": {"alpha": 0.2, "epsilon": 0.3}}, "training": {"optimizer": "adam"}}
model = Blackbird.Model(Connect4.BoardState, "t", {"explorationRate": 0.85, "playLimit": int(sys.argv[1])}, cfg)
n = int(sys.argv[2])
t = time.time(); Blackbird.GenerateTrainingSamples(model, 64, 1.0); print("warm", time.time() - t)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t = time.time(); Blackbird.GenerateTrainingSamples(model, n, 1.0); dt = time.time() - t
pr.disable()
print(f"{n} games at playLimit {sys.argv[1]}: {dt:.2f} s = {n/dt:.0f} games/s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
