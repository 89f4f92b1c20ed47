import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackbird_amd import _lib
game = _lib.GAME_CONNECT4
slots = 4096
eng = _lib.Engine(game, n_slots=slots, sims_per_move=800, evaluator=_lib.EVAL_HASH, max_games=slots * 8)
eng.selfplay_begin(slots * 8, 1.0)
eng.selfplay_step(3)
eng.synchronize()
