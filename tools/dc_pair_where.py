"""Where the hardware places the eight waves of a k_dc_selfplay_pair workgroup (HW_ID of each wave of workgroup 3).
Needs the diagnostic build tools/libbb_stamps.so (-DBB_STAMPS)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbb_stamps.so")
eng = _lib.Engine(_lib.GAME_DRAGONCHESS, n_slots=64, sims_per_move=8, evaluator=_lib.EVAL_NET, noise_on=True, max_games=64, max_plies=16)
eng.load_weights(W.flatten(W.init_weights(17, 16, 4, 16, 4032, seed=0)))
eng.selfplay_begin(64, 1.0)
eng.selfplay_step(1); eng.synchronize()
L = _lib.lib(); L.bb_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
st = np.zeros(16, dtype=np.uint64); L.bb_debug_stamps(eng.h, st.ctypes.data)
for w in range(8):
    v = int(st[w]) & 0xffff
    print(f"wave {w}: HW_ID low16 = {v:#06x}  wave_id {v & 15}  simd {(v >> 4) & 3}  pipe {(v >> 6) & 3}  cu {(v >> 8) & 15}")
