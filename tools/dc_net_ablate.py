"""DragonChess network kernel alone on a 1024-leaf mailbox, with parts switched off (tuning aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackbird_amd import _lib, weights as W
game = _lib.GAME_DRAGONCHESS
eng = _lib.Engine(game, n_slots=1024, sims_per_move=16, evaluator=_lib.EVAL_NET, noise_on=True, max_games=1024, max_plies=64)
eng.load_weights(W.flatten(W.init_weights(17, 16, 4, 16, 4032, seed=0)))
eng.selfplay_begin(1024, 1.0)
eng.selfplay_step(2)
eng.synchronize()
for name, noise, abl in [("full", 1, 0), ("no heads", 0, 1), ("no tower", 0, 2), ("no tower, no heads", 0, 3), ("full", 1, 0)]:
    ms = eng.timing_net(iters=100, noise=noise, ablate=abl)
    print(f"{name:22s} {ms * 1e3:8.1f} us/launch   {2694976 * 1024 / ms / 1e9:7.1f} TFLOP/s-equivalent")
