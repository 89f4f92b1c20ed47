import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackbird_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbb_stamps.so")
game = _lib.GAME_CONNECT4
slots = 4096
eng = _lib.Engine(game, n_slots=slots, sims_per_move=800, evaluator=_lib.EVAL_HASH, max_games=slots * 8)
eng.selfplay_begin(slots * 8, 1.0)
L = _lib.lib()
L.bb_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
import numpy as np
for ply in range(4):
    eng.selfplay_step(1)
    st = np.zeros(8, dtype=np.uint64)
    L.bb_debug_stamps(eng.h, st.ctypes.data)
    c = eng.counters()
    w = float(st[4])
    print(f"   per game-sim: load-wait ticks {st[5]/max(float(st[7]),1):.0f} over {st[6]/max(float(st[7]),1):.2f} loop iterations")
    print(f"ply {ply}: per wave-step ticks apply {st[0]/w:.0f} fence {st[1]/w:.0f} select {st[2]/w:.0f}; mean depth so far {c['sum_depth']/c['sims']:.2f}")
