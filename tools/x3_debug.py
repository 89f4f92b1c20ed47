import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
from oracle import orc
game = _lib.GAME_CONNECT4
gi = _lib.game_info(game)
rng = np.random.RandomState(3)
n = 8
cells = rng.randint(0, 3, size=(n, 6, 7))
b = np.zeros((n, 6, 7, 2), dtype=np.int8); b[..., 0] = cells == 1; b[..., 1] = cells == 2
st = _lib.pack_grid(game, b, rng.randint(1, 3, n))
planes = _lib.game_encode(game, st)
def run(name, mod, R=0):
    w = W.init_weights(gi.C, 16, R, 16, gi.A, seed=11, perturb=True)
    mod(w)
    flat = W.flatten(w)
    eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
    eng.load_weights(flat)
    v1, l1, p1 = eng.net_eval(planes=planes)
    v2, l2, p2 = eng.net_eval(planes=planes)
    v3, l3, p3 = eng.net_eval(states=st)
    ov, ol, op = orc.net_forward(orc.NetWeights(gi.H, gi.W, gi.C, 16, R, 16, gi.A, flat), planes)
    print(f"{name:28s} value err {np.abs(v1-ov).max():.2e} logits err {np.abs(l1-ol).max():.2e} repeatable {np.array_equal(l1,l2)} states==planes {np.array_equal(l1,l3)}")
    eng.close()
K = 'resTower/conv_block/conv/kernel'
def zero(w): w[K][:] = 0
def only(tap, c=None):
    def f(w):
        k = w[K].copy(); w[K][:] = 0
        if c is None: w[K][tap // 3, tap % 3] = k[tap // 3, tap % 3]
        else: w[K][tap // 3, tap % 3, c] = k[tap // 3, tap % 3, c]
    return f
run("zero kernel", zero)
for tap in range(9): run(f"only tap {tap}", only(tap))
for c in range(3): run(f"tap 4 channel {c}", only(4, c))
run("full", lambda w: None)
run("full R=1", lambda w: None, R=1)
