"""-m gpu: HIP game kernels (through the C ABI) vs the reference golden vectors and vs the oracle."""
import os

import numpy as np
import pytest

from blackbird_amd import _lib

pytestmark = pytest.mark.gpu
KEYS = {"c4": (_lib.GAME_CONNECT4, 0), "ttt": (_lib.GAME_TICTACTOE, 1)}


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _pack(game, g):
    H, W, _ = _lib.GRID[game]
    return _lib.pack_grid(game, g["board"].reshape(-1, H, W, 2), g["player"], g["prev"])


def _legal_dense(g, n, A):
    out = np.zeros((n, A), dtype=np.uint8)
    for i in range(n):
        out[i, g["legal_idx"][g["legal_off"][i]:g["legal_off"][i + 1]]] = 1
    return out


@pytest.mark.parametrize("key", ["c4", "ttt"])
def test_golden_random_boards(golden_dir, key):
    game, _ = KEYS[key]
    g = _load(golden_dir, f"boards_{key}.npz")
    st = _pack(game, g)
    n = st.shape[0]
    A = _lib.game_info(game).A
    assert np.array_equal(_lib.game_legal(game, st), _legal_dense(g, n, A))
    assert np.array_equal(_lib.game_winner(game, st), g["win_none"].astype(np.int8))
    assert np.array_equal(_lib.game_encode(game, st).reshape(n, -1), g["enc"])
    for a in range(A):
        acts = np.full(n, a, dtype=np.int32)
        assert np.array_equal(_lib.game_winner(game, st, acts), g["win_prev"][:, a].astype(np.int8)), a
        _s2, status = _lib.game_apply(game, st, acts)
        assert np.array_equal(status == 0, g["apply_ok"][:, a] == 1), a
        assert np.array_equal(_s2[status != 0], st[status != 0])  # ValueError leaves the board alone


@pytest.mark.parametrize("key", ["c4", "ttt"])
def test_golden_playouts(golden_dir, key):
    game, _ = KEYS[key]
    g = _load(golden_dir, f"playouts_{key}.npz")
    st = _pack(game, g)
    n = st.shape[0]
    A = _lib.game_info(game).A
    assert np.array_equal(_lib.game_legal(game, st), _legal_dense(g, n, A))
    assert np.array_equal(_lib.game_winner(game, st), g["win_none"].astype(np.int8))
    assert np.array_equal(_lib.game_encode(game, st).reshape(n, -1), g["enc"])
    mv = np.where(g["action"] >= 0)[0]
    # the reference applies moves to Copy(), which drops PreviousPlayer for these two games
    src = st[mv].copy()
    src[:, 0] &= ~(np.uint64(3) << np.uint64(58))
    nxt, status = _lib.game_apply(game, src, g["action"][mv])
    assert (status == 0).all()
    assert np.array_equal(nxt, st[mv + 1])  # same packed state the golden trajectory reaches
    assert np.array_equal(_lib.game_winner(game, nxt, g["action"][mv]), g["win_prev"][mv].astype(np.int8))


@pytest.mark.parametrize("key", ["c4", "ttt"])
def test_vs_oracle_large(orc, key):
    """200k arbitrary boards: every kernel bit-exact against the oracle."""
    game, og = KEYS[key]
    H, W, _ = _lib.GRID[game]
    A = _lib.game_info(game).A
    rng = np.random.RandomState(7)
    n = 20000
    fill = rng.uniform(0.0, 1.0, size=(n, 1, 1))
    r = rng.uniform(size=(n, H, W))
    cells = np.where(r < fill / 2, 1, np.where(r < fill, 2, 0))
    boards = np.zeros((n, H, W, 2), dtype=np.int8)
    boards[..., 0] = cells == 1
    boards[..., 1] = cells == 2
    players = rng.randint(1, 3, n)
    st = _lib.pack_grid(game, boards, players)
    legal = _lib.game_legal(game, st)
    wn = _lib.game_winner(game, st)
    enc = _lib.game_encode(game, st)
    acts = rng.randint(0, A, n).astype(np.int32)
    wp = _lib.game_winner(game, st, acts)
    nxt, status = _lib.game_apply(game, st, acts)
    nb, npl, npv = _lib.unpack_grid(game, nxt)
    for i in range(0, n, 7):  # oracle is per-board ctypes: sample every 7th
        s = orc.state_from_arrays(og, boards[i], players[i])
        assert np.array_equal(orc.legal(og, s), legal[i].astype(np.float64))
        w = orc.winner(og, s)
        assert (-1 if w is None else w) == wn[i]
        w = orc.winner(og, s, int(acts[i]))
        assert (-1 if w is None else w) == wp[i]
        assert np.array_equal(orc.encode(og, s)[0], enc[i])
        t = s.copy()
        rc = orc.apply(og, t, int(acts[i]))
        assert (rc == 0) == (status[i] == 0)
        if rc == 0:
            assert np.array_equal(np.frombuffer(bytes(t.b)[:H * W * 2], dtype=np.int8).reshape(H, W, 2), nb[i])
            assert t.player == npl[i] and t.prev == npv[i]


def test_full_size_properties():
    """BASELINE config size (4096 x 800-sim leaves ~ 3.3M boards): size-independent properties."""
    game = _lib.GAME_CONNECT4
    n = 1 << 20
    rng = np.random.RandomState(3)
    st = np.repeat(_lib.game_initial(game), n, axis=0)
    plies = np.zeros(n, dtype=np.int32)
    alive = np.ones(n, dtype=bool)
    for _ in range(42):
        legal = _lib.game_legal(game, st)
        assert ((legal.sum(1) > 0) | ~alive).all()
        pick = (rng.rand(n, 7) * legal).argmax(1).astype(np.int32)
        nxt, status = _lib.game_apply(game, st, pick)
        assert (status[alive] == 0).all()
        w = _lib.game_winner(game, nxt, pick)
        wfull = _lib.game_winner(game, nxt)
        # Winner(prevAction) == Winner(None) on positions reached by legal play
        assert np.array_equal(w[alive], wfull[alive])
        st = np.where(alive[:, None], nxt, st)
        plies += alive
        alive &= w < 0
        if not alive.any():
            break
    assert not alive.any()
    b, _p, _v = _lib.unpack_grid(game, st)
    assert np.array_equal(b.sum((1, 2, 3)), plies)  # one stone per ply
    assert (b.sum(3) <= 1).all()


# ---- DragonChess ---------------------------------------------------------------------------------------
def _pack_dc(g):
    return _lib.pack_dc(g["board"].reshape(-1, 8, 8), g["player"], g["prev"], g["castle"])


def test_dc_golden_random_boards(golden_dir):
    game = _lib.GAME_DRAGONCHESS
    g = _load(golden_dir, "boards_dc.npz")
    st = _pack_dc(g)
    n = st.shape[0]
    assert np.array_equal(_lib.game_legal(game, st), _legal_dense(g, n, 4032))
    assert np.array_equal(_lib.game_winner(game, st), g["win_none"].astype(np.int8))
    assert np.array_equal(_lib.game_encode(game, st).reshape(n, -1), g["enc"])
    ap = g["apply_ok"]  # [n][32][72]: action, ok, player, prev, castle[4], board[64]
    for k in range(ap.shape[1]):
        acts = ap[:, k, 0].astype(np.int32)
        sel = acts >= 0
        nxt, status = _lib.game_apply(game, st[sel], acts[sel])
        assert np.array_equal(status == 0, ap[sel, k, 1] == 1)
        ok = status == 0
        b, p, pv, cs = _lib.unpack_dc(nxt[ok])
        want = ap[sel][ok][:, k]
        assert np.array_equal(p, want[:, 2]) and np.array_equal(pv, want[:, 3])
        assert np.array_equal(cs, want[:, 4:8]) and np.array_equal(b.reshape(-1, 64), want[:, 8:72])
        assert np.array_equal(nxt[~ok], st[sel][~ok])


def test_dc_golden_playouts(golden_dir):
    game = _lib.GAME_DRAGONCHESS
    g = _load(golden_dir, "playouts_dc.npz")
    st = _pack_dc(g)
    n = st.shape[0]
    assert np.array_equal(_lib.game_legal(game, st), _legal_dense(g, n, 4032))
    assert np.array_equal(_lib.game_winner(game, st), g["win_none"].astype(np.int8))
    assert np.array_equal(_lib.game_encode(game, st).reshape(n, -1), g["enc"])
    mv = np.where(g["action"] >= 0)[0]
    nxt, status = _lib.game_apply(game, st[mv], g["action"][mv])
    assert (status == 0).all() and np.array_equal(nxt, st[mv + 1])  # incl. W,W,B turn order and castle flags
    assert np.array_equal(_lib.game_winner(game, nxt, g["action"][mv]), g["win_prev"][mv].astype(np.int8))
    # initial position and the start-position move list
    init = _lib.game_initial(game)
    assert np.array_equal(init, st[:1])
    la = np.where(_lib.game_legal(game, init)[0] == 1)[0]
    assert [(a // 63, (a % 63) + ((a % 63) >= a // 63)) for a in la] == \
        [(4, 3), (4, 5), (11, 19), (11, 27), (12, 20), (12, 28), (13, 21), (13, 29)]


def test_dc_random_games_vs_oracle(orc):
    """2048 random DragonChess games advanced in lock-step on the GPU; every 16th game is replayed on the oracle."""
    game = _lib.GAME_DRAGONCHESS
    n = 2048
    rng = np.random.RandomState(17)
    st = np.repeat(_lib.game_initial(game), n, axis=0)
    alive = np.ones(n, dtype=bool)
    ost = {i: orc.new_state(2) for i in range(0, n, 16)}
    for ply in range(60):
        legal = _lib.game_legal(game, st)
        cnt = legal.sum(1)
        assert (cnt[alive] > 0).all()
        pick = (rng.rand(n, 4032).astype(np.float32) * legal).argmax(1).astype(np.int32)
        nxt, status = _lib.game_apply(game, st, pick)
        assert (status[alive] == 0).all()
        for i, s in ost.items():
            if not alive[i]:
                continue
            assert np.array_equal(np.where(orc.legal(2, s) == 1)[0], np.where(legal[i] == 1)[0])
            assert orc.apply(2, s, int(pick[i])) == 0
            b, p, pv, cs = _lib.unpack_dc(nxt[i:i + 1])
            assert bytes(s.b)[:64] == b.tobytes() and s.player == p[0] and s.prev == pv[0] and list(s.castle) == list(cs[0])
        w = _lib.game_winner(game, nxt)
        st = np.where(alive[:, None], nxt, st)
        alive &= w < 0
    b, _p, _pv, _cs = _lib.unpack_dc(st)
    assert ((b == 1).sum((1, 2)) <= 1).all() and ((b == -1).sum((1, 2)) <= 1).all()
