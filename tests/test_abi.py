"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/blackbird_hip.h declares; compute entry points fail loudly without a GPU (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from blackbird_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "blackbird_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bb_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    L = C.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/blackbird_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == syms, "python binding list and header disagree"


def test_game_info_and_layouts():
    gi = _lib.game_info(_lib.GAME_CONNECT4)
    assert (gi.H, gi.W, gi.C, gi.A, gi.S, gi.state_bytes, gi.dense) == (6, 7, 3, 7, 8, 16, 1)
    gi = _lib.game_info(_lib.GAME_TICTACTOE)
    assert (gi.H, gi.W, gi.C, gi.A, gi.S, gi.state_bytes) == (3, 3, 3, 9, 16, 16)
    assert _lib.example_dtype(_lib.GAME_CONNECT4).itemsize == 64
    # initial position is host-computable (no kernel): empty board, player 1, no previous player
    s = _lib.game_initial(_lib.GAME_CONNECT4)
    b, p, pv = _lib.unpack_grid(_lib.GAME_CONNECT4, s)
    assert b.sum() == 0 and p[0] == 1 and pv[0] == 0


def test_pack_roundtrip():
    rng = np.random.RandomState(0)
    for game, (H, W, _s) in _lib.GRID.items():
        cells = rng.randint(0, 3, size=(50, H, W))
        boards = np.zeros((50, H, W, 2), dtype=np.int8)
        boards[..., 0] = cells == 1
        boards[..., 1] = cells == 2
        pl = rng.randint(1, 3, 50)
        pv = rng.randint(0, 3, 50)
        b2, p2, v2 = _lib.unpack_grid(game, _lib.pack_grid(game, boards, pl, pv))
        assert np.array_equal(b2, boards) and np.array_equal(p2, pl) and np.array_equal(v2, pv)


def test_no_cpu_fallback():
    """Without a GPU the product must raise, never silently compute on the host."""
    if _lib.lib().bb_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_lib.BlackbirdHipError):
        _lib.Engine(_lib.GAME_CONNECT4, 4, 8, evaluator=_lib.EVAL_HASH)
    st = _lib.game_initial(_lib.GAME_CONNECT4)
    with pytest.raises(_lib.BlackbirdHipError):
        _lib.game_legal(_lib.GAME_CONNECT4, st)


def test_product_does_not_use_oracle():
    """The oracle is test infrastructure: nothing under blackbird_amd/ may import, load or link it."""
    pkg = os.path.join(ROOT, "blackbird_amd")
    pat = re.compile(r"(import\s+oracle|from\s+oracle|liborc|orc\.h|oracle/)")
    for dp, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", "Makefile")):
                src = open(os.path.join(dp, f)).read()
                assert not pat.search(src), os.path.join(dp, f)


def test_empty_batches_are_noops():
    """n == 0 through the batched entry points: BB_OK and empty outputs, without touching a GPU."""
    import numpy as np
    from blackbird_amd import _lib
    for game in (_lib.GAME_CONNECT4, _lib.GAME_TICTACTOE, _lib.GAME_DRAGONCHESS):
        gi = _lib.game_info(game)
        st0 = _lib.game_initial(game)[:0] if _lib.lib().bb_device_count() > 0 else np.zeros((0, gi.state_bytes // np.dtype(_lib.STATE_DTYPE[game]).itemsize), dtype=_lib.STATE_DTYPE[game])
        assert _lib.game_legal(game, st0).shape == (0, gi.A)
        assert _lib.game_encode(game, st0).shape == (0, gi.H, gi.W, gi.C)
        assert _lib.game_winner(game, st0).shape == (0,)
        ns, status = _lib.game_apply(game, st0, np.zeros(0, np.int32))
        assert ns.shape[0] == 0 and status.shape == (0,)
