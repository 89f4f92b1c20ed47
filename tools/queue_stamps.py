import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
_lib.LIB_PATH = os.path.abspath(os.environ.get("BB_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbb_stamps.so")))
game = _lib.GAME_CONNECT4
eng = _lib.Engine(game, n_slots=4096, sims_per_move=800, evaluator=_lib.EVAL_NET, noise_on=True, max_games=4096 * 12)
eng.load_weights(W.flatten(W.init_weights(3, 16, 4, 16, 7, seed=0)))
eng.selfplay_begin(4096 * 12, 1.0)
eng.set_sims_per_move(32); eng.selfplay_step(48); eng.set_sims_per_move(800)
eng.selfplay_step(2)
L = _lib.lib(); L.bb_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
st = np.zeros(16, dtype=np.uint64); L.bb_debug_stamps(eng.h, st.ctypes.data)
import time
ns = np.zeros(8, dtype=np.uint64); L.bb_debug_net_stamps.argtypes = [C.c_void_p, C.c_void_p]; L.bb_debug_net_stamps(eng.h, ns.ctypes.data)
eng.synchronize(); t0 = time.time()
eng.selfplay_step(3); eng.synchronize(); dt = time.time() - t0
L.bb_debug_stamps(eng.h, st.ctypes.data)
st = st.astype(np.float64)
print(f"3 plies in {dt*1e3:.1f} ms")
print(f"wave lifetimes (mean over waves, shader cycles / 2.4e6 = ms): network {st[1]/max(st[4],1)/2.4e6:.1f} ms, tree {st[3]/max(st[5],1)/2.4e6:.1f} ms of a {dt*1e3:.1f} ms run"); print(f"net waves: busy {st[0]/st[1]:.3f}; cycles per evaluation {st[0]/st[15]:.0f}; evaluations {st[15]:.0f}")
print(f"tree waves: busy {st[2]/st[3]:.3f}; cycles per async call {st[2]/st[13]:.0f}; games per call {st[14]/st[13]:.2f}; calls {st[13]:.0f}")
v = st[12]
if v > 0 and os.environ.get("LIGHT"):
    print(f"LIGHT: async_game {st[11]/v:.0f} cycles per game-visit; longest game of each call: {st[6]/max(st[7],1):.0f} cycles per level in its descent loops")
elif v > 0 and os.environ.get("QMODE", "1") == "1":
    print(f"DEEP per game-visit cycles: total {st[11]/v:.0f} apply {st[8]/v:.0f} cached-backups {st[9]/v:.0f} move {st[10]/v:.0f} load-wait {st[6]/v:.0f}; levels/visit {st[7]/v:.2f}")
clk = 2.4e3  # shader cycles per us (approx); wall clock ticks are 100 MHz
print(f"per evaluation: queue wait {st[10]/st[15]/100:.1f} us, network {st[0]/st[15]/clk:.1f} us; result pick-up wait {st[8]/max(st[9],1)/100:.1f} us; tree call {st[2]/st[13]/clk:.1f} us")
L.bb_debug_net_stamps(eng.h, ns.ctypes.data); ns = ns.astype(np.float64)
if os.environ.get("NETSTAMPS"):
    e = st[15]
    print("network wave, cycles per evaluation: prologue %.0f, first conv %.0f, tower %.0f, head convs %.0f, value/policy/noise %.0f, apply (expand + backup) %.0f"
          % tuple(ns[i] / e for i in range(6)))
if os.environ.get("QMODE") == "2":
    passes = st[11]
    ns = ns / passes
    print("team passes %.0f, positions per pass %.2f, cycles per pass %.0f" % (passes, st[15] / passes, st[0] / passes))
    print("per pass, cycles (mean over the 4 waves where summed): prologue %.0f, first conv %.0f, layer compute %.0f, sync waits %.0f | heads: value wave %.0f, policy wave %.0f, noise+idle waves %.0f | mix+store %.0f"
          % (ns[0] / 4, ns[1] / 4, ns[3] / 4, ns[2] / 4, ns[4], ns[5], ns[6], ns[7] / 4))
