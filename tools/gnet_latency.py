"""Latency of one evaluation of a wide network through bb_net_eval (the single-position path FindMove uses)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
F, R = int(sys.argv[1]), int(sys.argv[2])
gi = _lib.game_info(_lib.GAME_CONNECT4)
eng = _lib.Engine(_lib.GAME_CONNECT4, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
eng.load_weights(W.flatten(W.init_weights(gi.C, F, R, 16, gi.A, seed=0)))
st = _lib.game_initial(_lib.GAME_CONNECT4)
for n in (1, 8, 64):
    s = np.repeat(st, n, axis=0)
    eng.net_eval(states=s)
    t = time.perf_counter()
    for _ in range(20):
        eng.net_eval(states=s)
    print(f"F={F} R={R} n={n}: {(time.perf_counter() - t) / 20 * 1e3:.2f} ms per bb_net_eval call")
