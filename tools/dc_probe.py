"""BASELINE configs[3] sanity: DragonChess, 400 sims/move, 1024 concurrent games, ply cap 512 (node+edge pools ~160 GB)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackbird_amd import _lib, weights as W
if os.environ.get("BB_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["BB_LIB"])
game = _lib.GAME_DRAGONCHESS
t = time.time()
eng = _lib.Engine(game, n_slots=1024, sims_per_move=400, evaluator=_lib.EVAL_NET, noise_on=True, max_games=2048, max_plies=int(os.environ.get("DC_MAX_PLIES","512")))
eng.load_weights(W.flatten(W.init_weights(17, 16, 4, 16, 4032, seed=0)))
print("create %.1fs" % (time.time() - t))
eng.selfplay_begin(2048, 1.0)
eng.selfplay_step(1); eng.synchronize()
for plies in (2, 4):
    eng.reset_counters()
    t = time.perf_counter()
    eng.selfplay_step(plies); eng.synchronize()
    dt = time.perf_counter() - t
    c = eng.counters()
    print(f"{plies} plies: {dt / plies * 1e3:.1f} ms/ply, {c['sims'] / dt / 1e6:.2f} M sims/s, depth {c['sum_depth'] / max(c['sims'], 1):.2f}, overflow {c['overflow']}, nodes {c['nodes']}")
