"""Step two engines (BB_SHADOW_MASK=0 and =MASK) launch by launch and report the first per-slot divergence."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbb_stamps.so")
NS, SIMS, NG = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
np.set_printoptions(linewidth=250)
names = ["sims_left", "game_lid", "ply", "root_N", "pend_leaf", "n_nodes", "root", "resume_cur", "sim_serial", "path_len"]
def mk(mask):
    os.environ["BB_MEGA_QUEUE"] = "1"; os.environ["BB_SHADOW_MASK"] = mask; os.environ["BB_QUEUE_LIMIT_S"] = "5"
    eng = _lib.Engine(_lib.GAME_CONNECT4, n_slots=NS, sims_per_move=SIMS, evaluator=_lib.EVAL_NET, noise_on=True, max_games=NG)
    eng.load_weights(W.flatten(W.init_weights(3, 16, 4, 16, 7, seed=3)))
    eng.selfplay_begin(NG, 1.0)
    return eng
L = None
def snap(eng):
    global L
    L = _lib.lib(); L.bb_debug_slot_i32.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    out = {}
    for i, n in enumerate(names):
        a = np.zeros(NS, dtype=np.int32); L.bb_debug_slot_i32(eng.h, i, a.ctypes.data); out[n] = a
    return out
a, b = mk("0"), mk(sys.argv[1])
for step in range(400):
    os.environ['BB_SHADOW_MASK'] = '0'; a.selfplay_step(1); a.synchronize()
    os.environ['BB_SHADOW_MASK'] = sys.argv[1]; b.selfplay_step(1); b.synchronize()
    sa, sb = snap(a), snap(b)
    bad = [n for n in names if not np.array_equal(sa[n], sb[n])]
    if bad:
        print("first divergence after launch", step, "in", bad)
        for n in names:
            print("%-10s ref %s" % (n, sa[n])); print("%-10s got %s" % ("", sb[n]))
        sys.exit(1)
print("no divergence in 400 launches; done:", a.selfplay_done(), b.selfplay_done())
ra, rb = a.fetch_examples(), b.fetch_examples()
print("examples equal:", len(ra[0]) == len(rb[0]) and ra[0].tobytes() == rb[0].tobytes(), len(ra[0]), len(rb[0]))
if ra[0].tobytes() != rb[0].tobytes():
    for i in range(min(len(ra[0]), len(rb[0]))):
        if ra[0][i].tobytes() != rb[0][i].tobytes():
            print(i, ra[0][i]); print(i, rb[0][i]); break

