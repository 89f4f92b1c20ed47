import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbb_stamps.so")
game = _lib.GAME_DRAGONCHESS
eng = _lib.Engine(game, n_slots=1024, sims_per_move=400, evaluator=_lib.EVAL_NET, noise_on=True, max_games=2048, max_plies=512)
eng.load_weights(W.flatten(W.init_weights(17, 16, 4, 16, 4032, seed=0)))
eng.selfplay_begin(2048, 1.0)
eng.selfplay_step(2); eng.synchronize()
L = _lib.lib(); L.bb_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
st = np.zeros(16, dtype=np.uint64); L.bb_debug_stamps(eng.h, st.ctypes.data)
eng.selfplay_step(2); eng.synchronize()
L.bb_debug_stamps(eng.h, st.ctypes.data); st = st.astype(np.float64)
n = st[4]
print(f"per game-step cycles: apply {st[0]/n:.0f} (expansions {st[8]/n:.2f} per step: gather+noise {st[5]/max(st[8],1):.0f}, np_sum {st[6]/max(st[8],1):.0f}, edge write {st[7]/max(st[8],1):.0f}), select {st[2]/n:.0f}; move generation {st[9]/max(st[8],1):.0f}")
print(f"select per game-step: levels {st[1]/n:.2f}; first loads {st[11]/n:.0f}, node rows {st[12]/n:.0f} ({st[12]/max(st[1],1):.0f} per level), edge rows {st[13]/n:.0f} ({st[13]/max(st[1],1):.0f} per level), PUCT + argmax {st[14]/n:.0f} ({st[14]/max(st[1],1):.0f} per level), child creation / leaf state / tail {st[15]/n:.0f}")
print(f"apply: entry -> first round of loads complete {st[10]/n:.0f}, second round {st[3]/n:.0f}")
if eng.selfplay_mode() == 5:
    print(f"one wave per game (mega_dc.hip.h): in the first line 'apply' is then the tree phases of a simulation (apply + select) = {st[0]/n:.0f} cycles and 'select' the network evaluation = {st[2]/n:.0f} cycles")
