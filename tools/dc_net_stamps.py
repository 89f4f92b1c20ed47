"""Cycles per section of the network evaluation inside the DragonChess one-wave-per-game kernel.
Needs the diagnostic build: hipcc ... -DBB_STAMPS -DBB_STAMPS_NET -shared -o tools/libbb_stamps_net.so blackbird_amd/csrc/engine.hip"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbb_stamps_net.so")
game = _lib.GAME_DRAGONCHESS
eng = _lib.Engine(game, n_slots=1024, sims_per_move=400, evaluator=_lib.EVAL_NET, noise_on=True, max_games=2048, max_plies=512)
eng.load_weights(W.flatten(W.init_weights(17, 16, 4, 16, 4032, seed=0)))
eng.selfplay_begin(2048, 1.0)
eng.selfplay_step(1); eng.synchronize()
L = _lib.lib()
L.bb_debug_net_stamps.argtypes = [C.c_void_p, C.c_void_p]
L.bb_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
ns = np.zeros(8, dtype=np.uint64); L.bb_debug_net_stamps(eng.h, ns.ctypes.data)
st = np.zeros(16, dtype=np.uint64); L.bb_debug_stamps(eng.h, st.ctypes.data)
eng.selfplay_step(2); eng.synchronize()
L.bb_debug_net_stamps(eng.h, ns.ctypes.data); L.bb_debug_stamps(eng.h, st.ctypes.data)
n = float(st[4])
names = ["prologue (state, zero, planes, addressing)", "first conv", "tower", "1x1 head convs", "head tail (all of it)", "heads", "  of which dense_1 + R0/R1 + value (from tail entry)", "7"]
for i, nm in enumerate(names):
    if ns[i]:
        print(f"{nm}: {ns[i] / n:.0f} cycles per evaluation")
print(f"network total {st[2] / n:.0f}, tree phases {st[0] / n:.0f} (stamps add overhead)")
