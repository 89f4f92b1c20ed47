"""Kernel-tuning aid: logits / value / policy of the engine's 16-filter network against the oracle's float32 restatement
(max abs / relative difference), Connect4 and TicTacToe, random and perturbed weights.  F32=1 selects the float32
MFMA path (bb_config.net_form) for comparison."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
from oracle import orc
FORM = _lib.NET_FORM_F32 if os.environ.get("F32") == "1" else _lib.NET_FORM_AUTO
for game in (_lib.GAME_CONNECT4, _lib.GAME_TICTACTOE):
    gi = _lib.game_info(game)
    for perturb in (False, True):
        for R in (4, 1, 0):
            w = W.init_weights(gi.C, 16, R, 16, gi.A, seed=11, perturb=perturb)
            flat = W.flatten(w)
            eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET, net_form=FORM)
            eng.load_weights(flat)
            rng = np.random.RandomState(3)
            n = 203
            H, Wd, _ = _lib.GRID[game]
            cells = rng.randint(0, 3, size=(n, H, Wd))
            b = np.zeros((n, H, Wd, 2), dtype=np.int8); b[..., 0] = cells == 1; b[..., 1] = cells == 2
            st = _lib.pack_grid(game, b, rng.randint(1, 3, n))
            planes = _lib.game_encode(game, st)
            v1, l1, p1 = eng.net_eval(states=st)
            v2, l2, p2 = eng.net_eval(planes=planes)
            ov, ol, op = orc.net_forward(orc.NetWeights(gi.H, gi.W, gi.C, 16, R, 16, gi.A, flat), planes)
            same = np.array_equal(v1, v2) and np.array_equal(l1, l2)
            print(f"game {game} perturb {perturb} R {R}: value {np.abs(v1-ov).max():.2e} logits abs {np.abs(l1-ol).max():.2e} "
                  f"rel {(np.abs(l1-ol)/np.maximum(1,np.abs(ol))).max():.2e} policy {np.abs(p1-op).max():.2e} | logit scale {np.abs(ol).max():.2f} | states==planes {same}")
            eng.close()
# DragonChess (17 input planes, 4032-wide head)
game = _lib.GAME_DRAGONCHESS
for R in (4, 1):
    w = W.init_weights(17, 16, R, 16, 4032, seed=13, perturb=True)
    flat = W.flatten(w)
    eng = _lib.Engine(game, n_slots=2, sims_per_move=2, evaluator=_lib.EVAL_NET, max_plies=8, net_form=FORM)
    eng.load_weights(flat)
    rng = np.random.RandomState(5)
    n = 11
    boards = np.zeros((n, 8, 8), dtype=np.int8)
    for i in range(n):
        m = rng.rand(8, 8) < 0.35
        boards[i][m] = rng.choice([-6, -5, -4, -3, -2, -1, 1, 2], m.sum())
    st = _lib.pack_dc(boards, rng.randint(1, 3, n), rng.randint(0, 3, n), rng.randint(0, 2, (n, 4)))
    planes = _lib.game_encode(game, st)
    v1, l1, p1 = eng.net_eval(states=st)
    v2, l2, p2 = eng.net_eval(planes=planes)
    ov, ol, op = orc.net_forward(orc.NetWeights(8, 8, 17, 16, R, 16, 4032, flat), planes)
    print(f"DragonChess R {R} form {eng.net_form()}: value {np.abs(v1-ov).max():.2e} logits rel {(np.abs(l1-ol)/np.maximum(1,np.abs(ol))).max():.2e} policy {np.abs(p1-op).max():.2e} | states==planes {np.array_equal(l1,l2) and np.array_equal(v1,v2)}")
    eng.close()
