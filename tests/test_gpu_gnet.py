"""-m gpu: the general-filter-count network path (csrc/gnet.hip.h: one implicit-GEMM launch per conv layer) vs the
oracle's float32 restatement, tolerance 1e-5 (BASELINE.json north_star), and vs the fused F=16 kernels, which it
must reproduce bit for bit when forced onto a 16-filter network (bb_config.general_net)."""

import numpy as np
import pytest

from blackbird_amd import _lib, weights as W
from .test_gpu_net import boards_for

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _check(orc, game, F, R, n, seed, perturb=True, form=_lib.NET_FORM_AUTO):
    gi = _lib.game_info(game)
    flat = W.flatten(W.init_weights(gi.C, F, R, 16, gi.A, seed=seed, perturb=perturb))
    eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET, net_form=form)
    eng.load_weights(flat)
    assert eng.net_form() == (1 if form == _lib.NET_FORM_F32 or R == 0 else 3)
    rng = np.random.RandomState(n + F)
    b, pl = boards_for(game, rng, n)
    st = _lib.pack_grid(game, b, pl)
    planes = _lib.game_encode(game, st)
    v1, l1, p1 = eng.net_eval(states=st)
    v2, l2, p2 = eng.net_eval(planes=planes)
    assert np.array_equal(v1, v2) and np.array_equal(l1, l2) and np.array_equal(p1, p2)
    ov, ol, op = orc.net_forward(orc.NetWeights(gi.H, gi.W, gi.C, F, R, 16, gi.A, flat), planes)
    assert np.max(np.abs(v1 - ov)) <= TOL
    assert np.max(np.abs(l1 - ol) / np.maximum(1.0, np.abs(ol))) <= TOL
    assert np.max(np.abs(p1 - op)) <= TOL
    if eng.net_form() == 1 and form == _lib.NET_FORM_F32:
        assert np.mean(l1 == ol) > 0.99  # float32-MFMA layers: the K order of the oracle's fmaf chains
    perm = rng.permutation(n)  # batch invariance, also across workgroup boundaries
    v3, l3, p3 = eng.net_eval(states=st[perm])
    assert np.array_equal(v3, v1[perm]) and np.array_equal(l3, l1[perm]) and np.array_equal(p3, p1[perm])
    eng.close()


@pytest.mark.parametrize("game", [_lib.GAME_CONNECT4, _lib.GAME_TICTACTOE])
@pytest.mark.parametrize("F,R,n", [(32, 2, 1), (32, 2, 37), (64, 3, 9), (48, 1, 130), (128, 2, 21)])
def test_general_filters_vs_oracle(orc, game, F, R, n):
    _check(orc, game, F, R, n, seed=21)
    if (F, n) in ((32, 37), (64, 9)):  # the float32-MFMA layers (BB_NET_FORM_F32): the oracle's own K order
        _check(orc, game, F, R, n, seed=21, form=_lib.NET_FORM_F32)


def test_c5_shape_vs_oracle(orc):
    """BASELINE configs[4]: 20 blocks x 256 filters on Connect4 (a few positions: the oracle needs ~2 GFLOP each)."""
    _check(orc, _lib.GAME_CONNECT4, 256, 20, 3, seed=5, perturb=False)


def test_general_path_equals_fused_path_at_16_filters():
    """The general-filter kernels and the fused float32-MFMA tower are the same k-ordered fmaf chains: identical bits.  The
    default 16-filter path of the dense games runs on the bf16 matrix pipe with three-way split operands (net_x3.hip.h):
    float32 products, another summation order -- within the 1e-5 of the north star, not bit-identical."""
    game = _lib.GAME_CONNECT4
    gi = _lib.game_info(game)
    flat = W.flatten(W.init_weights(gi.C, 16, 4, 16, gi.A, seed=3, perturb=True))
    rng = np.random.RandomState(0)
    b, pl = boards_for(game, rng, 77)
    st = _lib.pack_grid(game, b, pl)
    outs = {}
    for name, gnet, form in (("fused f32", False, _lib.NET_FORM_F32), ("general", True, _lib.NET_FORM_F32), ("fused x3", False, _lib.NET_FORM_SPLIT)):
        eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET, noise_on=True, general_net=gnet, net_form=form)
        eng.load_weights(flat)
        outs[name] = eng.net_eval(states=st)
        eng.close()
    for a, b2 in zip(outs["fused f32"], outs["general"]):
        assert np.array_equal(a, b2)
    v, l, p = outs["fused f32"]
    v3, l3, p3 = outs["fused x3"]
    assert np.max(np.abs(v3 - v)) <= 1e-5 and np.max(np.abs(p3 - p)) <= 1e-5
    assert np.max(np.abs(l3 - l) / np.maximum(1.0, np.abs(l))) <= 1e-5


def test_dragonchess_general_filters(orc):
    game = _lib.GAME_DRAGONCHESS
    flat = W.flatten(W.init_weights(17, 32, 2, 16, 4032, seed=13, perturb=True))
    eng = _lib.Engine(game, n_slots=2, sims_per_move=2, evaluator=_lib.EVAL_NET, max_plies=8)
    eng.load_weights(flat)
    rng = np.random.RandomState(5)
    n = 7
    boards = np.zeros((n, 8, 8), dtype=np.int8)
    for i in range(n):
        m = rng.rand(8, 8) < 0.35
        boards[i][m] = rng.choice([-6, -5, -4, -3, -2, -1, 1, 2], m.sum())
    st = _lib.pack_dc(boards, rng.randint(1, 3, n), rng.randint(0, 3, n), rng.randint(0, 2, (n, 4)))
    planes = _lib.game_encode(game, st)
    v1, l1, p1 = eng.net_eval(states=st)
    ov, ol, op = orc.net_forward(orc.NetWeights(8, 8, 17, 32, 2, 16, 4032, flat), planes)
    assert np.max(np.abs(v1 - ov)) <= TOL
    assert np.max(np.abs(l1 - ol) / np.maximum(1.0, np.abs(ol))) <= TOL
    assert np.max(np.abs(p1 - op)) <= TOL and np.allclose(p1.sum(1), 1.0, atol=1e-4)
    eng.close()


def test_selfplay_with_general_network_matches_oracle_tree(orc):
    """Asynchronous self-play driven by a 32-filter network == the oracle's sequential search fed with the same
    network values (taken from the GPU so that only the search is compared)."""
    from .test_gpu_net import _selfplay_vs_oracle_tree
    _selfplay_vs_oracle_tree(orc, filters=32, blocks=2)


@pytest.mark.parametrize("F,R,n", [(32, 2, 1500), (128, 2, 300)])
def test_large_batch_kernels_equal_small_batch_kernels(orc, F, R, n):
    """Big batches run the throughput launch (4 positions x 2 filter blocks per wave), small batches the fine-grained latency launch (1 position x 1 filter block per wave): the same K order,
    so the same bits -- and both within 1e-5 of the oracle."""
    game = _lib.GAME_CONNECT4
    gi = _lib.game_info(game)
    flat = W.flatten(W.init_weights(gi.C, F, R, 16, gi.A, seed=9, perturb=True))
    eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
    eng.load_weights(flat)
    rng = np.random.RandomState(F)
    b, pl = boards_for(game, rng, n)
    st = _lib.pack_grid(game, b, pl)
    v, l, p = eng.net_eval(states=st)                      # n * F/16 > 2048: throughput kernels
    for lo in range(0, n, 50):                             # <= 50 positions a call: latency launch
        v2, l2, p2 = eng.net_eval(states=st[lo:lo + 50])
        assert np.array_equal(v2, v[lo:lo + 50]) and np.array_equal(l2, l[lo:lo + 50]) and np.array_equal(p2, p[lo:lo + 50])
    k = 12
    ov, ol, op = orc.net_forward(orc.NetWeights(gi.H, gi.W, gi.C, F, R, 16, gi.A, flat), _lib.game_encode(game, st[:k]))
    assert np.max(np.abs(v[:k] - ov)) <= TOL and np.max(np.abs(p[:k] - op)) <= TOL
    assert np.max(np.abs(l[:k] - ol) / np.maximum(1.0, np.abs(ol))) <= TOL
    eng.close()


def test_wide_network_rounds_on_two_streams(orc):
    """A wide network runs self-play as asynchronous rounds; with >= 512 slots two slot-range views are pipelined on two
    streams, both run the general path at once and must not share activation scratch: self-play still equals the oracle's search."""
    from .test_gpu_net import _selfplay_vs_oracle_tree
    _selfplay_vs_oracle_tree(orc, filters=32, blocks=1, n_slots=512, n_games=24, sims=16)
