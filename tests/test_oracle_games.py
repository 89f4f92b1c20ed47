"""Oracle (oracle/orc_games.c) vs golden vectors generated from the reference's GameState classes
(tests/make_golden.py parts 'playouts' and 'boards')."""
import os

import numpy as np
import pytest

KEYS = {"c4": 0, "ttt": 1, "dc": 2}


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _state(orc, game, g, i):
    return orc.state_from_arrays(game, g["board"][i], g["player"][i], g["prev"][i] or None,
                                 g["castle"][i])


@pytest.mark.parametrize("key", ["c4", "ttt", "dc"])
def test_playouts(orc, golden_dir, key):
    game = KEYS[key]
    g = _load(golden_dir, f"playouts_{key}.npz")
    gs = g["game_start"]
    n_checked = 0
    for gi in range(len(gs) - 1):
        st = orc.new_state(game)
        for i in range(gs[gi], gs[gi + 1]):
            # state reached by the oracle's own ApplyAction equals the reference's
            ref = _state(orc, game, g, i)
            assert bytes(st.b) == bytes(ref.b) and st.player == ref.player
            assert st.prev == ref.prev
            assert list(st.castle) == list(ref.castle)
            la = orc.legal(game, st)
            want = g["legal_idx"][g["legal_off"][i]:g["legal_off"][i + 1]]
            assert np.array_equal(np.where(la == 1)[0], want)
            assert set(np.unique(la)) <= {0.0, 1.0}
            assert np.array_equal(orc.encode(game, st).ravel(), g["enc"][i])
            w = orc.winner(game, st)
            assert (-1 if w is None else w) == g["win_none"][i]
            a = int(g["action"][i])
            if a < 0:
                break
            nxt = st.copy()
            if game != 2:
                nxt.prev = 0  # Copy() drops PreviousPlayer (Connect4.py:24-28)
            assert orc.apply(game, nxt, a) == 0
            wp = orc.winner(game, nxt, a)
            assert (-1 if wp is None else wp) == g["win_prev"][i]
            st = nxt
            n_checked += 1
    assert n_checked > 100


@pytest.mark.parametrize("key", ["c4", "ttt"])
def test_random_boards_grid(orc, golden_dir, key):
    game = KEYS[key]
    g = _load(golden_dir, f"boards_{key}.npz")
    A = orc.dims(game)[3]
    for i in range(len(g["player"])):
        st = _state(orc, game, g, i)
        la = orc.legal(game, st)
        want = g["legal_idx"][g["legal_off"][i]:g["legal_off"][i + 1]]
        assert np.array_equal(np.where(la == 1)[0], want)
        w = orc.winner(game, st)
        assert (-1 if w is None else w) == g["win_none"][i], i
        for a in range(A):
            wp = orc.winner(game, st, a)
            assert (-1 if wp is None else wp) == g["win_prev"][i][a], (i, a)
            t = st.copy()
            assert (orc.apply(game, t, a) == 0) == bool(g["apply_ok"][i][a])
        assert np.array_equal(orc.encode(game, st).ravel(), g["enc"][i])


def test_random_boards_dc(orc, golden_dir):
    game = 2
    g = _load(golden_dir, "boards_dc.npz")
    for i in range(len(g["player"])):
        st = _state(orc, game, g, i)
        la = orc.legal(game, st)
        want = g["legal_idx"][g["legal_off"][i]:g["legal_off"][i + 1]]
        assert np.array_equal(np.where(la == 1)[0], want), i
        w = orc.winner(game, st)
        assert (-1 if w is None else w) == g["win_none"][i]
        assert np.array_equal(orc.encode(game, st).ravel(), g["enc"][i])
        for row in g["apply_ok"][i]:
            a, ok = int(row[0]), int(row[1])
            if a < 0:
                continue
            t = st.copy()
            rc = orc.apply(game, t, a)
            assert (rc == 0) == bool(ok), (i, a)
            if ok:
                assert t.player == row[2] and t.prev == row[3]
                assert list(t.castle) == [int(x) for x in row[4:8]]
                assert list(t.b[:64]) == [int(x) for x in row[8:72]]


def test_dc_action_index_formula(orc):
    # DragonChess.py:26-34: enumeration order == sq1*63 + sq2 - (sq2 > sq1)
    idx = 0
    for s1 in range(64):
        for s2 in range(64):
            if s1 == s2:
                continue
            assert idx == s1 * 63 + s2 - (s2 > s1)
            idx += 1
    assert idx == 4032
    # start position: 8 legal moves (SURVEY 8a edge cases)
    st = orc.new_state(2)
    la = np.where(orc.legal(2, st) == 1)[0]
    moves = [(a // 63, (a % 63) + ((a % 63) >= a // 63)) for a in la]
    assert moves == [(4, 3), (4, 5), (11, 19), (11, 27), (12, 20), (12, 28), (13, 21), (13, 29)]


def test_np_sum_matches_numpy(orc):
    rng = np.random.default_rng(5)
    import ctypes as C
    for n in (1, 7, 8, 9, 15, 16, 17, 64, 127, 128, 129, 1000, 4032):
        for _ in range(50):
            a = rng.random(n) * rng.choice([1e-3, 1.0, 1e3], n)
            if n == 4032:
                a = a * (rng.random(n) < 0.01)
            got = orc.lib().orc_np_sum(a.ctypes.data, n)
            assert got == float(np.sum(a)), n
