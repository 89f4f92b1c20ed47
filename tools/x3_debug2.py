import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
game = _lib.GAME_CONNECT4
gi = _lib.game_info(game)
rng = np.random.RandomState(3)
n = 8
cells = rng.randint(0, 3, size=(n, 6, 7))
b = np.zeros((n, 6, 7, 2), dtype=np.int8); b[..., 0] = cells == 1; b[..., 1] = cells == 2
st = _lib.pack_grid(game, b, rng.randint(1, 3, n))
planes = _lib.game_encode(game, st)
w = W.init_weights(gi.C, 16, 0, 16, gi.A, seed=11, perturb=True)
flat = W.flatten(w)
eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
eng.load_weights(flat)
v1, l1, p1 = eng.net_eval(planes=planes)
print("values", v1)
# expected first conv output, channel 0 of pixel 0 (y=0,x=0), float64
k = w['resTower/conv_block/conv/kernel'].astype(np.float64); bias = w['resTower/conv_block/conv/bias'].astype(np.float64)
g_, be, mu, var = [w['resTower/conv_block/batch_norm/' + s].astype(np.float64) for s in ('gamma', 'beta', 'moving_mean', 'moving_variance')]
x = planes.reshape(n, 6, 7, 3).astype(np.float64)
xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
out = np.zeros(n)
for i in range(n):
    acc = bias[0]
    for dy in range(3):
        for dx in range(3):
            acc += (xp[i, 0 + dy, 0 + dx] * k[dy, dx, :, 0]).sum()
    out[i] = max((acc - mu[0]) / np.sqrt(var[0] + 1e-3) * g_[0] + be[0], 0)
print("expect", out)
from oracle import orc
ov, ol, op = orc.net_forward(orc.NetWeights(gi.H, gi.W, gi.C, 16, 0, 16, gi.A, flat), planes)
np.set_printoptions(linewidth=200, precision=5)
print("GPU logits\n", l1[:4]); print("oracle logits\n", ol[:4]); print("GPU value", v1, "\noracle", ov)
v2, l2, p2 = eng.net_eval(planes=planes)
print("second call logits\n", l2[:4])
