"""Tuning aid: tree-kernel time vs games-per-wave (hash evaluator => evaluator cost ~5 us)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackbird_amd import _lib
game = _lib.GAME_CONNECT4
for gpw in (8, 4, 2, 1):
    os.environ["BB_TREE_GPW"] = str(gpw)
    slots = 4096
    eng = _lib.Engine(game, n_slots=slots, sims_per_move=800, evaluator=_lib.EVAL_HASH, max_games=slots * 8)
    eng.selfplay_begin(slots * 8, 1.0)
    eng.selfplay_step(2)
    eng.synchronize()
    t = time.perf_counter()
    eng.selfplay_step(3)
    eng.synchronize()
    dt = time.perf_counter() - t
    c = eng.counters()
    print(f"gpw {gpw}: {dt / 3 / 800 * 1e6:.1f} us per sim-step (tree + ~5 us hash), depth {c['sum_depth'] / c['sims']:.2f}, overflow {c['overflow']}")
    eng.close()
