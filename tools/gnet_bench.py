"""Throughput of the general-filter network path (BASELINE configs[4] shape by default): evaluations/s and f32 MFMA TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blackbird_amd import _lib, weights as W
if os.environ.get("BB_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["BB_LIB"])
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
gi = _lib.game_info(_lib.GAME_CONNECT4)
eng = _lib.Engine(_lib.GAME_CONNECT4, n_slots=n, sims_per_move=2, evaluator=_lib.EVAL_NET, noise_on=True)
eng.load_weights(W.flatten(W.init_weights(gi.C, F, R, 16, gi.A, seed=0)))
ms = eng.timing_net(1, True, 0)  # warm-up
ms = eng.timing_net(3, True, 0)
HW = 42
flops = 2 * HW * 9 * gi.C * F + 4 * R * HW * 9 * F * F + 6 * HW * F + 4 * gi.A + 4 * 16
print(f"F={F} R={R} n={n}: {ms:.2f} ms per batch, {n / ms * 1e3:.0f} evals/s, {n * flops / ms / 1e9:.1f} TFLOP/s algorithmic (peak 157.3)")
