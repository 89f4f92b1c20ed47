import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from blackbird_amd import _lib, weights as W
def run(env, game, n_slots, n_games, sims, noise):
    for k in ("BB_MEGA", "BB_ASYNC"): os.environ.pop(k, None)
    os.environ.update(env)
    gi = _lib.game_info(game)
    flat = W.flatten(W.init_weights(gi.C, 16, 4, 16, gi.A, seed=21, perturb=True))
    eng = _lib.Engine(game, n_slots=n_slots, sims_per_move=sims, evaluator=_lib.EVAL_NET, seed=5, max_games=n_games, noise_on=noise, alpha=0.2, epsilon=0.3)
    eng.load_weights(flat)
    mode = eng.selfplay_mode()
    eng.selfplay_begin(n_games, 1.0)
    while not eng.selfplay_done()[0]:
        eng.selfplay_step(2)
    rec, offs, win = eng.fetch_examples()
    eng.close()
    return rec, offs, win, mode
for game, ns, ng, sims in ((1, 16, 40, 24), (1, 5, 10, 24), (0, 37, 60, 40)):
    for noise in (True, False):
        a = run({}, game, ns, ng, sims, noise)
        for name, env in (("rounds", {"BB_MEGA": "0"}), ("lockstep", {"BB_MEGA": "0", "BB_ASYNC": "0"})):
            b = run(env, game, ns, ng, sims, noise)
            same = np.array_equal(a[1], b[1]) and a[0].tobytes() == b[0].tobytes()
            print(game, ns, noise, name, b[3], "same" if same else "DIFFERENT", len(a[0]), len(b[0]))
