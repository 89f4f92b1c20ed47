"""-m gpu: the fused MFMA tower + heads (through the C ABI) vs the oracle's float32 restatement.
Tolerance 1e-5 on value / logits / policy (BASELINE.json north_star).  The float32-MFMA tower (bb_config.net_form =
BB_NET_FORM_F32) is bit-identical to the oracle's k-ordered fmaf chains; the default tower of every 16-filter network (Connect4,
TicTacToe and DragonChess) runs on the bf16 matrix pipe with three-way split operands (exact float32 products, the MFMA's own
summation order): measured 3e-6 relative on logits, 6e-7 on values."""
import numpy as np
import pytest

from blackbird_amd import _lib, weights as W

pytestmark = pytest.mark.gpu
TOL = 1e-5


def boards_for(game, rng, n):
    H, Wd, _ = _lib.GRID[game]
    cells = rng.randint(0, 3, size=(n, H, Wd))
    b = np.zeros((n, H, Wd, 2), dtype=np.int8)
    b[..., 0] = cells == 1
    b[..., 1] = cells == 2
    pl = rng.randint(1, 3, n)
    return b, pl


@pytest.mark.parametrize("game,og", [(_lib.GAME_CONNECT4, 0), (_lib.GAME_TICTACTOE, 1)])
@pytest.mark.parametrize("perturb", [False, True])
@pytest.mark.parametrize("n", [1, 5, 16, 203])
@pytest.mark.parametrize("x3", ["1", "0"])
def test_net_vs_oracle(orc, game, og, perturb, n, x3):
    gi = _lib.game_info(game)
    w = W.init_weights(gi.C, 16, 4, 16, gi.A, seed=11, perturb=perturb)
    flat = W.flatten(w)
    eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET,
                      net_form=_lib.NET_FORM_AUTO if x3 == "1" else _lib.NET_FORM_F32)
    eng.load_weights(flat)
    assert eng.net_form() == (2 if x3 == "1" else 0)
    rng = np.random.RandomState(n)
    b, pl = boards_for(game, rng, n)
    st = _lib.pack_grid(game, b, pl)
    planes = _lib.game_encode(game, st)
    v1, l1, p1 = eng.net_eval(states=st)
    v2, l2, p2 = eng.net_eval(planes=planes)
    # packed-state and AsInputArray inputs are the same computation
    assert np.array_equal(v1, v2) and np.array_equal(l1, l2) and np.array_equal(p1, p2)
    ov, ol, op = orc.net_forward(orc.NetWeights(gi.H, gi.W, gi.C, 16, 4, 16, gi.A, flat), planes)
    assert np.max(np.abs(v1 - ov)) <= TOL
    assert np.max(np.abs(l1 - ol) / np.maximum(1.0, np.abs(ol))) <= TOL
    assert np.max(np.abs(p1 - op)) <= TOL
    print(f"bit-exact logits: {np.mean(l1 == ol):.3f}, value: {np.mean(v1 == ov):.3f}")
    if x3 == "0":
        assert np.mean(l1 == ol) > 0.99  # k-ordered f32 MFMA chain == oracle fmaf chain
    # batch invariance: a position's outputs do not depend on where it sits in the batch
    perm = rng.permutation(n)
    v3, l3, p3 = eng.net_eval(states=st[perm])
    assert np.array_equal(v3, v1[perm]) and np.array_equal(l3, l1[perm]) and np.array_equal(p3, p1[perm])
    eng.close()


@pytest.mark.parametrize("form", [_lib.NET_FORM_AUTO, _lib.NET_FORM_F32])
def test_net_dc_vs_oracle(orc, form):
    """DragonChess network: 17 input planes, 8x8, 4032-wide policy head -- in the default split-operand form (1e-5 of the oracle)
    and in the float32-MFMA form (tower + value head are the oracle's fmaf chains)."""
    game = _lib.GAME_DRAGONCHESS
    w = W.init_weights(17, 16, 2, 16, 4032, seed=13, perturb=True)
    flat = W.flatten(w)
    eng = _lib.Engine(game, n_slots=2, sims_per_move=2, evaluator=_lib.EVAL_NET, max_plies=8, net_form=form)
    eng.load_weights(flat)
    assert eng.net_form() == (2 if form == _lib.NET_FORM_AUTO else 0)
    rng = np.random.RandomState(5)
    n = 11
    boards = np.zeros((n, 8, 8), dtype=np.int8)
    for i in range(n):
        m = rng.rand(8, 8) < 0.35
        boards[i][m] = rng.choice([-6, -5, -4, -3, -2, -1, 1, 2], m.sum())
    st = _lib.pack_dc(boards, rng.randint(1, 3, n), rng.randint(0, 3, n), rng.randint(0, 2, (n, 4)))
    planes = _lib.game_encode(game, st)
    v1, l1, p1 = eng.net_eval(states=st)
    v2, l2, p2 = eng.net_eval(planes=planes)
    assert np.array_equal(v1, v2) and np.array_equal(l1, l2) and np.array_equal(p1, p2)
    ov, ol, op = orc.net_forward(orc.NetWeights(8, 8, 17, 16, 2, 16, 4032, flat), planes)
    assert np.max(np.abs(v1 - ov)) <= TOL
    assert np.max(np.abs(l1 - ol) / np.maximum(1.0, np.abs(ol))) <= TOL
    assert np.max(np.abs(p1 - op)) <= TOL and np.allclose(p1.sum(1), 1.0, atol=1e-4)
    if form == _lib.NET_FORM_F32:
        assert np.array_equal(v1, ov) or np.max(np.abs(v1 - ov)) <= 2e-7  # tower + value head are the oracle's fmaf chains
    eng.close()


def test_noise_distribution():
    """Prior noise is Beta(alpha, 1-alpha) per action, mixed with weight eps and renormalised."""
    game = _lib.GAME_CONNECT4
    gi = _lib.game_info(game)
    w = W.init_weights(gi.C, 16, 4, 16, gi.A, seed=2)
    eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET, alpha=0.2, epsilon=0.3, seed=8)
    eng.load_weights(W.flatten(w))
    n = 20000
    st = np.repeat(_lib.game_initial(game), n, axis=0)
    _v, _l, clean = eng.net_eval(states=st)
    _v, _l, noisy = eng.net_eval(states=st, noise=True)
    assert np.allclose(noisy.sum(1), 1.0, atol=1e-5)
    # undo the renormalisation: q = (1-eps) p + eps x, noisy = q / sum(q)  =>  sum(q) = (1-eps) + eps*sum(x)
    # E[x] = alpha, Var[x] = alpha(1-alpha)/2 for Beta(alpha, 1-alpha)
    # estimate x from ratios: noisy_a / noisy_b is not linear, so test moments of sum(q)-free quantity:
    q_over = noisy / noisy.sum(1, keepdims=True)
    assert np.all(q_over > 0)
    # Monte-Carlo expectation of q_a/sum(q) computed on the host with numpy's Beta sampler
    rng = np.random.RandomState(0)
    x = rng.beta(0.2, 0.8, size=(200000, gi.A))
    q = 0.7 * clean[0][None, :] + 0.3 * x
    ref = (q / q.sum(1, keepdims=True))
    assert np.allclose(noisy.mean(0), ref.mean(0), atol=4e-3)
    assert np.allclose(noisy.std(0), ref.std(0), atol=6e-3)
    eng.close()


def test_tree_with_net_matches_oracle_tree_on_gpu_values(orc):
    """End to end with the real network: the oracle's tree search, fed the GPU network's outputs
    through its callback evaluator, must reproduce the engine's visit counts exactly."""
    _selfplay_vs_oracle_tree(orc, filters=16, blocks=4)


@pytest.mark.parametrize("game,og,n_slots,n_games,max_plies", [(_lib.GAME_TICTACTOE, 1, 21, 50, 10), (_lib.GAME_CONNECT4, 0, 19, 45, 43)])
def test_persistent_kernel_ragged_slots_and_slot_reuse(orc, game, og, n_slots, n_games, max_plies):
    """The default persistent kernel with a slot count that is not a multiple of its 16-game workgroups and more games
    than slots (every slot plays several games): still the oracle's search, game by game."""
    _selfplay_vs_oracle_tree(orc, filters=16, blocks=2, game=game, og=og, n_slots=n_slots, n_games=n_games, sims=24,
                             max_plies=max_plies)


def _selfplay_vs_oracle_tree(orc, filters, blocks, game=_lib.GAME_CONNECT4, og=0, n_slots=8, n_games=6, sims=40, max_plies=43):
    gi = _lib.game_info(game)
    w = W.init_weights(gi.C, filters, blocks, 16, gi.A, seed=21)
    flat = W.flatten(w)
    eng = _lib.Engine(game, n_slots=n_slots, sims_per_move=sims, evaluator=_lib.EVAL_NET, seed=17, max_games=n_games)
    eng.load_weights(flat)
    eng.selfplay_begin(n_games, 1.0)
    while not eng.selfplay_done()[0]:
        eng.selfplay_step(2)
    rec, offs, win = eng.fetch_examples()
    ev = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
    ev.load_weights(flat)

    def cb(_ctx, stp, vp, pp):
        planes = orc.encode(og, stp.contents)
        v, _l, p = ev.net_eval(planes=planes)
        vp[0] = float(v[0])
        if pp:
            for a in range(gi.A):
                pp[a] = float(p[0, a])

    cfg = orc.make_cfg(og, evaluator=orc.EVAL_CALLBACK, seed=17, cb=orc.EVAL_CB(cb))
    for gidx in range(n_games):
        o = orc.selfplay_game(cfg, gidx, 1.0, sims, max_plies - 1)
        r = rec[offs[gidx]:offs[gidx + 1]]
        assert len(r) == o["n"] and win[gidx] == o["winner"], gidx
        tot = np.maximum(r["total"].astype(np.float64), 1.0)[:, None]
        assert np.array_equal(r["visits"][:, :gi.A] / tot, o["pi"]), gidx
    eng.close()
    ev.close()


def test_empty_and_single_position_batches():
    """n = 0 is a no-op; n = 1 (the FindMove case) takes the one-position-per-wave launch and equals the same position
    inside a large batch bit for bit."""
    game = _lib.GAME_CONNECT4
    gi = _lib.game_info(game)
    eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
    eng.load_weights(W.flatten(W.init_weights(gi.C, 16, 4, 16, gi.A, seed=2)))
    rng = np.random.RandomState(1)
    b, pl = boards_for(game, rng, 3000)
    st = _lib.pack_grid(game, b, pl)
    v0, l0, p0 = eng.net_eval(states=st[:0])
    assert v0.shape == (0,) and l0.shape == (0, gi.A) and p0.shape == (0, gi.A)
    v, l, p = eng.net_eval(states=st)            # 3000 positions: 4 per wave
    for i in (0, 1234, 2999):
        v1, l1, p1 = eng.net_eval(states=st[i:i + 1])   # 1 per wave
        assert v1[0] == v[i] and np.array_equal(l1[0], l[i]) and np.array_equal(p1[0], p[i])
    v2, l2, p2 = eng.net_eval(states=st[:1500])  # 2 per wave
    assert np.array_equal(v2, v[:1500]) and np.array_equal(l2, l[:1500]) and np.array_equal(p2, p[:1500])
    eng.close()


@pytest.mark.parametrize("game,og", [(_lib.GAME_CONNECT4, 0), (_lib.GAME_TICTACTOE, 1)])
def test_every_positions_per_wave_variant_matches_oracle(orc, game, og):
    """The fused kernel is instantiated for 1, 2 and the LDS-filling number of positions per wave (12 for TicTacToe, 4 for
    Connect4), picked by batch size; the tile counts (1, 2, 7 / 3, 6, 11 tiles of 16 pixels) go through different operand
    schedules.  All of them against the oracle, logits bit-identical.  (Round 2 found a compiler reordering across the
    cross-lane LDS hand-over between layers that only struck the one-tile variant: net.hip.h wave_lds_handover.)
    These are the float32-MFMA kernels (BB_NET_FORM_F32); the bf16-pipe form has one position per wave at every batch size."""
    gi = _lib.game_info(game)
    flat = W.flatten(W.init_weights(gi.C, 16, 4, 16, gi.A, seed=2, perturb=True))
    eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET, net_form=_lib.NET_FORM_F32)
    eng.load_weights(flat)
    rng = np.random.RandomState(1)
    b, pl = boards_for(game, rng, 3000)
    st = _lib.pack_grid(game, b, pl)
    planes = _lib.game_encode(game, st)
    ov, ol, op = orc.net_forward(orc.NetWeights(gi.H, gi.W, gi.C, 16, 4, 16, gi.A, flat), planes[:300])
    for cnt in (3000, 1500, 300, 1):
        v, l, p = eng.net_eval(states=st[:cnt])
        k = min(cnt, 300)
        assert np.array_equal(l[:k], ol[:k]), (cnt, float(np.abs(l[:k] - ol[:k]).max()))
        assert np.max(np.abs(v[:k] - ov[:k])) <= TOL and np.max(np.abs(p[:k] - op[:k])) <= TOL
    eng.close()


@pytest.mark.parametrize("game", [_lib.GAME_CONNECT4, _lib.GAME_TICTACTOE])
def test_split_operand_tower_every_tap_and_plane_contributes(game):
    """The bf16-pipe tower (net_x3.hip.h) forms a float32 product from three bf16 planes per operand.  A weight 1 + 2^-10 + 2^-20
    has one bit in each plane; placed on one tap of the first conv (a network without residual blocks, identity batch norm) the
    value head sees it only if that tap's slice carried all three planes.  The first version lost plane 2 of every K = 32 slice:
    a K = 32 MFMA directly followed by a K = 16 MFMA on the same accumulator reads its SrcC too early on gfx950 / ROCm 7.2
    (the tower uses one MFMA kind now and tap 8 is three plane-concatenated K = 32 products: net_x3.hip.h).  Checked through the public outputs: the value of an empty board equals the float64 statement within
    1e-6 only when nothing is lost (a lost plane 2 costs 1e-3)."""
    gi = _lib.game_info(game)
    H, Wd, _ = _lib.GRID[game]
    st = _lib.pack_grid(game, np.zeros((1, H, Wd, 2), dtype=np.int8), np.ones(1, dtype=np.int64))  # empty board, player 1: plane 2 = +1
    planes = _lib.game_encode(game, st)
    K = "resTower/conv_block/conv/kernel"
    wv = 1 + 2.0 ** -10 + 2.0 ** -20
    for tap in range(9):
        w = W.init_weights(gi.C, 16, 0, 16, gi.A, seed=11)
        w[K][:] = 0
        w[K][tap // 3, tap % 3, 2, 0] = wv
        w["resTower/conv_block/conv/bias"][:] = 0
        for s, v in (("gamma", 1), ("beta", 0), ("moving_mean", 0), ("moving_variance", 1 - 1e-3)):
            w["resTower/conv_block/batch_norm/" + s][:] = v
        eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
        eng.load_weights(W.flatten(w))
        assert eng.net_form() == 2
        v, l, p = eng.net_eval(planes=planes)
        eng.close()
        # float64 statement: channel 0 of the first conv = wv where the tap's neighbour is on the board, pooled value head after it
        dy, dx = tap // 3 - 1, tap % 3 - 1
        inside = sum(1 for y in range(H) for x in range(Wd) if 0 <= y + dy < H and 0 <= x + dx < Wd)
        x0 = np.zeros(16)
        vk = w["value/convolution/kernel"].astype(np.float64).reshape(16)
        g_, be, mu, var = [w["value/batch_norm/" + s].astype(np.float64)[0] for s in ("gamma", "beta", "moving_mean", "moving_variance")]
        pre_on = wv * vk[0] + float(w["value/convolution/bias"][0])
        pre_off = float(w["value/convolution/bias"][0])
        bn = lambda a: max((a - mu) / np.sqrt(var + 1e-3) * g_ + be, 0.0)
        R = inside * bn(pre_on) + (H * Wd - inside) * bn(pre_off)
        d1k = w["value/dense_1/kernel"].astype(np.float64).reshape(-1); d1b = w["value/dense_1/bias"].astype(np.float64)
        hdn = np.maximum(R * d1k + H * Wd * d1b, 0.0)
        expect = np.tanh(hdn @ w["value/dense_2/kernel"].astype(np.float64).reshape(-1) + float(w["value/dense_2/bias"][0]))
        assert abs(float(v[0]) - expect) <= 1e-6, (tap, float(v[0]), expect)
        del x0


# ---- the split-operand arithmetic over its envelope (bb_config.net_form AUTO / SPLIT) -------------------------------------
def _forward64(w, planes):
    """The graph of NetworkFactory.py:22-183 in float64 numpy (NHWC, SAME padding, batch-norm epsilon 1e-3): an independent
    statement of what both the float32 oracle and the GPU forms approximate.  Returns value, logits."""
    x = planes.astype(np.float64)

    def conv(t, k, b):
        kh = k.shape[0] // 2
        tp = np.pad(t, ((0, 0), (kh, kh), (kh, kh), (0, 0)))
        out = np.zeros(t.shape[:3] + (k.shape[3],))
        for dy in range(k.shape[0]):
            for dx in range(k.shape[1]):
                out += np.einsum("nhwc,cf->nhwf", tp[:, dy:dy + t.shape[1], dx:dx + t.shape[2], :], k[dy, dx].astype(np.float64))
        return out + b.astype(np.float64)

    def bn(t, p):
        g, b, m, v = (w[f"{p}/{n}"].astype(np.float64) for n in ("gamma", "beta", "moving_mean", "moving_variance"))
        return g * (t - m) / np.sqrt(v + 1e-3) + b

    relu = lambda t: np.maximum(t, 0.0)
    t = relu(bn(conv(x, w["resTower/conv_block/conv/kernel"], w["resTower/conv_block/conv/bias"]), "resTower/conv_block/batch_norm"))
    i = 0
    while f"resTower/block_{i}/conv_1/kernel" in w:
        u = relu(bn(conv(t, w[f"resTower/block_{i}/conv_1/kernel"], w[f"resTower/block_{i}/conv_1/bias"]), f"resTower/block_{i}/batch_norm_1"))
        u = bn(conv(u, w[f"resTower/block_{i}/conv_2/kernel"], w[f"resTower/block_{i}/conv_2/bias"]), f"resTower/block_{i}/batch_norm_2")
        t = relu(u + t)
        i += 1
    v = relu(bn(conv(t, w["value/convolution/kernel"], w["value/convolution/bias"]), "value/batch_norm"))
    v = (v @ w["value/dense_1/kernel"].astype(np.float64) + w["value/dense_1/bias"]).sum(axis=(1, 2))
    v = np.tanh(relu(v) @ w["value/dense_2/kernel"].astype(np.float64) + w["value/dense_2/bias"])[:, 0]
    p = relu(bn(conv(t, w["policy/convolution/kernel"], w["policy/convolution/bias"]), "policy/batch_norm"))
    logits = (p @ w["policy/policy/kernel"].astype(np.float64) + w["policy/policy/bias"]).sum(axis=(1, 2))
    return v, logits


def _layer_names(blocks):
    names = [("resTower/conv_block/conv", "resTower/conv_block/batch_norm")]
    for i in range(blocks):
        names += [(f"resTower/block_{i}/conv_{j}", f"resTower/block_{i}/batch_norm_{j}") for j in (1, 2)]
    return names


def _rescaled(w, layer, k, blocks):
    """Kernel, bias and moving mean of one conv layer times 2^k, its batch-norm gamma divided by 2^k: the network function is
    unchanged in exact arithmetic, and every scaling is a power of two -- exact in float32 and in every bf16 plane."""
    conv, bnp = _layer_names(blocks)[layer]
    out = {n: a.copy() for n, a in w.items()}
    s = np.float32(2.0) ** k
    out[conv + "/kernel"] = (w[conv + "/kernel"] * s).astype(np.float32)
    out[conv + "/bias"] = (w[conv + "/bias"] * s).astype(np.float32)
    out[bnp + "/moving_mean"] = (w[bnp + "/moving_mean"] * s).astype(np.float32)
    out[bnp + "/gamma"] = (w[bnp + "/gamma"] / s).astype(np.float32)
    return out


@pytest.mark.parametrize("game", [_lib.GAME_CONNECT4, _lib.GAME_DRAGONCHESS])
def test_split_operand_envelope(orc, game):
    """The default arithmetic (float32 operands as three exact bf16 planes, six bf16 MFMA products, float32 accumulation) over the
    range a trained network can reach, against the float32 oracle (north star: 1e-5) AND against a float64 statement of the
    graph -- the split form must be as close to the exact result as a float32 fmaf chain is:
      * one layer's kernel x 2^k with its batch-norm gamma / 2^k (k = -20 .. 20): pre-activations 1e-6 .. 1e6 times the usual,
      * moving_variance down to 1e-8 (scale 31.6 gamma), gamma up to 64: activations beyond 1e4,
      * kernels small enough that the third bf16 plane of a weight is a bf16 subnormal (2^-115): the hardware's treatment of
        subnormal MFMA inputs shows here and nowhere in a realistic network."""
    gi = _lib.game_info(game)
    R = 4 if game == _lib.GAME_CONNECT4 else 2
    w = W.init_weights(gi.C, 16, R, 16, gi.A, seed=31, perturb=True)
    rng = np.random.RandomState(3)
    if game == _lib.GAME_CONNECT4:
        b, pl = boards_for(game, rng, 24)
        st = _lib.pack_grid(game, b, pl)
    else:
        boards = np.zeros((8, 8, 8), dtype=np.int8)
        for i in range(8):
            m = rng.rand(8, 8) < 0.35
            boards[i][m] = rng.choice([-6, -5, -4, -3, -2, -1, 1, 2], m.sum())
        st = _lib.pack_dc(boards, rng.randint(1, 3, 8), rng.randint(0, 3, 8), rng.randint(0, 2, (8, 4)))
    planes = _lib.game_encode(game, st)

    def run(wts, what, tol=TOL, c64=4.0):
        flat = W.flatten(wts)
        eng = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET, max_plies=8, net_form=_lib.NET_FORM_SPLIT)
        eng.load_weights(flat)
        assert eng.net_form() == 2
        v, l, p = eng.net_eval(states=st)
        eng.close()
        ov, ol, op = orc.net_forward(orc.NetWeights(gi.H, gi.W, gi.C, 16, R, 16, gi.A, flat), planes)
        v64, l64 = _forward64(wts, planes)
        den = np.maximum(1.0, np.abs(l64))
        e_gpu, e_orc = np.max(np.abs(l - l64) / den), np.max(np.abs(ol - l64) / den)
        print(f"{what}: |logit| <= {np.abs(l64).max():.3g}; vs oracle {np.max(np.abs(l - ol) / den):.2e}; vs float64: GPU {e_gpu:.2e}, oracle {e_orc:.2e}")
        # two float32 forms cannot agree better than each agrees with the exact result: where float32 arithmetic itself is
        # > 1e-5 away from float64 (logits ~1e5) the bound between them follows it
        tol_l = max(tol, 1.25 * (e_gpu + e_orc))
        assert np.max(np.abs(l - ol) / den) <= tol_l and np.max(np.abs(v - ov)) <= tol_l and np.max(np.abs(p - op)) <= tol_l, what
        assert e_gpu <= max(c64 * e_orc, 2e-6), what  # as accurate as float32 arithmetic itself
        return v, l, p

    base = run(w, "base")
    for layer, k in [(0, -20), (1, 20), (2, -12), (3, 12), (2 * R, -20), (2 * R, 20), (1, -6), (2, 6)]:
        got = run(_rescaled(w, layer, k, R), f"layer {layer} x 2^{k}")
        # powers of two scale every plane and every product exactly: the SAME bits as the unscaled network
        assert np.array_equal(got[1], base[1]) and np.array_equal(got[0], base[0]), (layer, k)
    # tiny variances and large gammas: large activations
    big = {n: a.copy() for n, a in w.items()}
    big["resTower/conv_block/batch_norm/moving_variance"][:] = 1e-8
    big["resTower/conv_block/batch_norm/gamma"] *= 64.0
    big["resTower/block_0/batch_norm_1/moving_variance"][:] = 1e-8
    # (logits ~1e5 here: float32 arithmetic itself -- the oracle's fmaf chains -- is 2e-5 .. 5e-5 away from the float64 result;
    # what is asserted is that the split form is as close to exact as the float32 one)
    run(big, "variance 1e-8, gamma x 64")
    # third plane of the weights subnormal in bf16 (|w| ~ 2^-3 x 2^-112, third plane ~ 2^-131 < 2^-126)
    run(_rescaled(w, 1, -112, R), "layer 1 x 2^-112 (third plane subnormal)", tol=1e-4, c64=64.0)
