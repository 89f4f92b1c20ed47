import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from blackbird_amd import _lib, weights as W
for game in (1, 0):
    gi = _lib.game_info(game)
    H, Wd, _ = _lib.GRID[game]
    eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
    eng.load_weights(W.flatten(W.init_weights(gi.C, 16, 4, 16, gi.A, seed=2, perturb=True)))
    rng = np.random.RandomState(1)
    n = 3000
    cells = rng.randint(0, 3, size=(n, H, Wd))
    b = np.zeros((n, H, Wd, 2), dtype=np.int8); b[..., 0] = cells == 1; b[..., 1] = cells == 2
    st = _lib.pack_grid(game, b, rng.randint(1, 3, n))
    v, l, p = eng.net_eval(states=st)          # PW max
    v2, l2, p2 = eng.net_eval(states=st[:1500])  # PW 2
    v1 = np.zeros(64, np.float32); l1 = np.zeros((64, gi.A), np.float32)
    for i in range(64):
        a = eng.net_eval(states=st[i:i + 1]); v1[i] = a[0][0]; l1[i] = a[1][0]
    print(game, "PWmax vs PW1: value", np.array_equal(v[:64], v1), "logits", np.array_equal(l[:64], l1), np.abs(l[:64] - l1).max(),
          "| PW2 vs PW1:", np.array_equal(v2[:64], v1), np.array_equal(l2[:64], l1), np.abs(l2[:64] - l1).max())
    bad = np.nonzero((l[:64] != l1).any(1))[0]
    print("  mismatching positions (PWmax):", bad[:20], " (PW2):", np.nonzero((l2[:64] != l1).any(1))[0][:20])
