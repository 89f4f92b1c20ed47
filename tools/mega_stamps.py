import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbb_stamps.so")
game = _lib.GAME_CONNECT4
eng = _lib.Engine(game, n_slots=4096, sims_per_move=800, evaluator=_lib.EVAL_NET, noise_on=True, max_games=4096 * 12)
eng.load_weights(W.flatten(W.init_weights(3, 16, 4, 16, 7, seed=0)))
eng.selfplay_begin(4096 * 12, 1.0)
eng.set_sims_per_move(32); eng.selfplay_step(48); eng.set_sims_per_move(800)
eng.selfplay_step(2)
L = _lib.lib(); L.bb_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
st = np.zeros(16, dtype=np.uint64); L.bb_debug_stamps(eng.h, st.ctypes.data)
eng.selfplay_step(3)
L.bb_debug_stamps(eng.h, st.ctypes.data)
phases = 2 * 3 * 800
print(f"tree levels per game-visit {st[7]/(4096*3*800):.2f}, load-wait cycles per level {st[6]/max(float(st[7]),1):.0f}")
v=float(st[12]); print(f"per game-visit cycles: total {st[11]/v:.0f} apply {st[8]/v:.0f} cached-backups {st[9]/v:.0f} move {st[10]/v:.0f} load-wait {st[6]/v:.0f}")
print(f"per phase cycles: net work {st[0]/st[4]/phases:.0f} of {st[1]/st[4]/phases:.0f}; tree work {st[2]/st[5]/phases:.0f} of {st[3]/st[5]/phases:.0f}")
