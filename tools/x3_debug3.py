import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
game = _lib.GAME_CONNECT4
gi = _lib.game_info(game)
n = 4
b = np.zeros((n, 6, 7, 2), dtype=np.int8)
st = _lib.pack_grid(game, b, np.ones(n, dtype=np.int64))   # empty boards, player 1: plane 2 = +1 everywhere
planes = _lib.game_encode(game, st)
K = 'resTower/conv_block/conv/kernel'
for tap in range(9):
    w = W.init_weights(gi.C, 16, 0, 16, gi.A, seed=11)
    w[K][:] = 0
    w[K][tap // 3, tap % 3, 2, 0] = 1 + 2.0 ** -10 + 2.0 ** -20
    w['resTower/conv_block/conv/bias'][:] = 0
    for s, v in (('gamma', 1), ('beta', 0), ('moving_mean', 0), ('moving_variance', 1 - 1e-3)):
        w['resTower/conv_block/batch_norm/' + s][:] = v
    eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
    eng.load_weights(W.flatten(w))
    v1, l1, p1 = eng.net_eval(planes=planes)
    inside = (tap // 3 >= 1) and (tap % 3 >= 1)
    print(f"tap {tap}: got {v1[0]:.9f} expect {(1 + 2.0**-10 + 2.0**-20) if inside else 0:.9f}")
    eng.close()
