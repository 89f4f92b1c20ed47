"""Protocol check: work-queue kernel with the hash evaluator computed in its network waves (Q_HASH build) must equal
round-based asynchronous self-play with the hash evaluator."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
_lib.LIB_PATH = os.environ['BB_LIB']

def run(queue, slots, sims, games):
    os.environ["BB_MEGA_QUEUE"] = "1" if queue else "0"
    os.environ["BB_MEGA"] = "1" if queue else "0"
    os.environ["BB_QUEUE_LIMIT_S"] = "5"
    eng = _lib.Engine(_lib.GAME_CONNECT4, n_slots=slots, sims_per_move=sims, evaluator=_lib.EVAL_NET if queue else _lib.EVAL_HASH, noise_on=False, max_games=games)
    if queue:
        eng.load_weights(W.flatten(W.init_weights(3, 16, 4, 16, 7, seed=3)))
    eng.selfplay_begin(games, 1.0)
    for _ in range(400):
        eng.selfplay_step(1)
        if eng.selfplay_done()[0]:
            break
    rec, offs, win = eng.fetch_examples()
    return rec, offs, win, eng.counters()

slots, sims, games = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
a = run(False, slots, sims, games)
b = run(True, slots, sims, games)
print("rounds", a[3]); print("queue ", b[3])
if len(a[0]) == len(b[0]) and a[0].tobytes() == b[0].tobytes():
    print("IDENTICAL", len(a[0]), "examples")
else:
    n = min(len(a[0]), len(b[0]))
    for i in range(n):
        if a[0][i].tobytes() != b[0][i].tobytes():
            g = int(np.searchsorted(a[1], i, "right") - 1)
            print("first mismatch at example", i, "game", g, "ply", i - a[1][g])
            print(a[0][i]); print(b[0][i])
            break
    sys.exit(1)
