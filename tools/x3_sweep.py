import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from blackbird_amd import _lib, weights as W
from oracle import orc
game = _lib.GAME_CONNECT4
gi = _lib.game_info(game)
worst = (0, 0, 0)
for seed in range(8):
    for perturb in (False, True):
        w = W.init_weights(gi.C, 16, 4, 16, gi.A, seed=100 + seed, perturb=perturb)
        flat = W.flatten(w)
        eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
        eng.load_weights(flat)
        rng = np.random.RandomState(seed)
        n = 400
        cells = rng.randint(0, 3, size=(n, 6, 7))
        b = np.zeros((n, 6, 7, 2), dtype=np.int8); b[..., 0] = cells == 1; b[..., 1] = cells == 2
        st = _lib.pack_grid(game, b, rng.randint(1, 3, n))
        planes = _lib.game_encode(game, st)
        v, l, p = eng.net_eval(states=st)
        ov, ol, op = orc.net_forward(orc.NetWeights(gi.H, gi.W, gi.C, 16, 4, 16, gi.A, flat), planes)
        e = (np.abs(v - ov).max(), (np.abs(l - ol) / np.maximum(1, np.abs(ol))).max(), np.abs(p - op).max())
        worst = tuple(max(a, b_) for a, b_ in zip(worst, e))
        eng.close()
print("worst over 16 networks x 400 positions: value %.2e logits(rel) %.2e policy %.2e" % worst)
