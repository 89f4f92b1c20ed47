import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from blackbird_amd import _lib, weights as W
if os.environ.get("BB_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["BB_LIB"])
from oracle import orc
for game, og in ((1, 1), (0, 0)):
    gi = _lib.game_info(game)
    H, Wd, _ = _lib.GRID[game]
    flat = W.flatten(W.init_weights(gi.C, 16, 4, 16, gi.A, seed=2, perturb=True))
    eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
    eng.load_weights(flat)
    rng = np.random.RandomState(1)
    n = 3000
    cells = rng.randint(0, 3, size=(n, H, Wd))
    b = np.zeros((n, H, Wd, 2), dtype=np.int8); b[..., 0] = cells == 1; b[..., 1] = cells == 2
    st = _lib.pack_grid(game, b, rng.randint(1, 3, n))
    planes = _lib.game_encode(game, st)
    ov, ol, op = orc.net_forward(orc.NetWeights(gi.H, gi.W, gi.C, 16, 4, 16, gi.A, flat), planes[:200])
    for cnt in (3000, 1500, 200):
        v, l, p = eng.net_eval(states=st[:cnt])
        print(game, cnt, "max |logit diff| vs oracle", np.abs(l[:200] - ol).max(), "bit-identical frac", np.mean(l[:200] == ol))
