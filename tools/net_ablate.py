"""Kernel-tuning aid: time the network kernel alone on a 4096-leaf mailbox, with parts switched off."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W

game = _lib.GAME_CONNECT4
eng = _lib.Engine(game, n_slots=4096, sims_per_move=64, evaluator=_lib.EVAL_NET, noise_on=True, max_games=4096 * 4)
eng.load_weights(W.flatten(W.init_weights(3, 16, 4, 16, 7, seed=0)))
eng.selfplay_begin(4096 * 4, 1.0)
eng.selfplay_step(6)  # realistic mid-game leaves in the mailbox
eng.synchronize()
for name, noise, abl in [("full+noise", 1, 0), ("noise w/o beta draw", 1, 8), ("full", 0, 0), ("no heads", 0, 1), ("no tower", 0, 2),
                         ("no tower, no heads", 0, 3), ("full+noise", 1, 0)]:
    ms = eng.timing_net(iters=200, noise=noise, ablate=abl)
    print(f"{name:22s} {ms * 1e3:8.1f} us/launch   {1588700 * 4096 / ms / 1e9:7.1f} TFLOP/s-equivalent")
