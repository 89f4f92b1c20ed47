"""Kernel-tuning aid (diagnostic build -DBB_STAMPS -DBB_STAMPS_LIGHT): cycles per tree level of async_game, in whichever
self-play structure LAUNCH selects (LAUNCH=2: asynchronous rounds -- separate tree launches, no network waves on the SIMD)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
_lib.LIB_PATH = os.path.abspath(os.environ.get("BB_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbb_stamps_deep.so")))
game = _lib.GAME_CONNECT4
n = int(os.environ.get("SLOTS", "4096"))
eng = _lib.Engine(game, n_slots=n, sims_per_move=800, evaluator=_lib.EVAL_NET, noise_on=True, max_games=n * 12, launch=int(os.environ.get("LAUNCH", "0")))
eng.load_weights(W.flatten(W.init_weights(3, 16, 4, 16, 7, seed=0)))
eng.selfplay_begin(n * 12, 1.0)
eng.set_sims_per_move(32); eng.selfplay_step(48); eng.set_sims_per_move(800)
eng.selfplay_step(2)
L = _lib.lib(); L.bb_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
st = np.zeros(16, dtype=np.uint64); L.bb_debug_stamps(eng.h, st.ctypes.data)
import time
eng.synchronize(); t0 = time.time()
eng.selfplay_step(3); eng.synchronize(); dt = time.time() - t0
L.bb_debug_stamps(eng.h, st.ctypes.data); st = st.astype(np.float64)
print(f"mode {eng.selfplay_mode()}, {n} slots: 3 steps in {dt*1e3:.1f} ms; longest game of each call: {st[6]/max(st[7],1):.0f} cycles per level in its descent loops (row-load wait {st[11]/max(st[7],1):.0f}, PUCT + argmax {st[12]/max(st[7],1):.0f})")
