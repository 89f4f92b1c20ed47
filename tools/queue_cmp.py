"""Self-play under the phase kernel and under the work-queue kernel must give identical examples."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
if os.environ.get("BB_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["BB_LIB"])
if os.environ.get('BB_LIB'):
    _lib.LIB_PATH = os.environ['BB_LIB']

def run(queue, slots, sims, games, noise):
    os.environ["BB_MEGA_QUEUE"] = os.environ.get("QMODE", "1") if queue else "0"
    os.environ["BB_QUEUE_LIMIT_S"] = "5"
    eng = _lib.Engine(_lib.GAME_CONNECT4, n_slots=slots, sims_per_move=sims, evaluator=_lib.EVAL_NET, noise_on=noise, max_games=games)
    eng.load_weights(W.flatten(W.init_weights(3, 16, 4, 16, 7, seed=3)))
    eng.selfplay_begin(games, 1.0)
    for _ in range(400):
        eng.selfplay_step(1)
        if eng.selfplay_done()[0]:
            break
    rec, offs, win = eng.fetch_examples()
    return rec, offs, win, eng.counters()

slots, sims, games = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
noise = len(sys.argv) > 4 and sys.argv[4] == "noise"
a = run(False, slots, sims, games, noise)
b = run(True, slots, sims, games, noise)
print("phase", a[3]); print("queue", b[3])
print("offs equal", np.array_equal(a[1], b[1]), "winners equal", np.array_equal(a[2], b[2]))
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
