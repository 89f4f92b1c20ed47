"""-m gpu: the Python mirror of the reference API (GameState / MCTS / Network / Blackbird) driven the
way the reference's own code drives it, checked against golden vectors from the reference."""
import glob
import os

import numpy as np
import pytest

from blackbird_amd import _lib, Connect4, TicTacToe
from blackbird_amd import weights as W
from blackbird_amd.Blackbird import ExampleState
from blackbird_amd.DynamicMCTS import DynamicMCTS
from blackbird_amd.FixedMCTS import FixedMCTS
from blackbird_amd.GameState import GameState

pytestmark = pytest.mark.gpu
GAMES = {"c4": Connect4.BoardState, "ttt": TicTacToe.BoardState}


def make_state(cls, board, player, prev):
    s = cls()
    s.Board = np.array(board, dtype=np.int8).reshape(s.Board.shape)
    s.Player = int(player)
    s.PreviousPlayer = int(prev) or None
    return s


@pytest.mark.parametrize("key", ["c4", "ttt"])
def test_gamestate_interface(golden_dir, key):
    cls = GAMES[key]
    g = np.load(os.path.join(golden_dir, f"playouts_{key}.npz"), allow_pickle=False)
    gs = g["game_start"]
    for gi in range(0, len(gs) - 1, 9):
        s = cls()
        assert isinstance(s, GameState) and s.PreviousPlayer is None and s.Player == 1
        for i in range(gs[gi], gs[gi + 1]):
            la = s.LegalActions()
            assert la.dtype == np.float64
            assert np.array_equal(np.where(la == 1)[0], g["legal_idx"][g["legal_off"][i]:g["legal_off"][i + 1]])
            x = s.AsInputArray()
            assert x.dtype == np.int8 and x.shape[0] == 1 and np.array_equal(x.ravel(), g["enc"][i])
            w = s.Winner()
            assert (-1 if w is None else int(w)) == g["win_none"][i]
            a = int(g["action"][i])
            if a < 0:
                break
            t = s.Copy()
            assert t.PreviousPlayer is None and t == s and hash(t) == hash(s)
            t.ApplyAction(a)
            wp = t.Winner(a)
            assert (-1 if wp is None else int(wp)) == g["win_prev"][i]
            assert t.PreviousPlayer == s.Player and t.Player == 3 - s.Player
            s = t
    # illegal move -> ValueError('Tried to make an illegal move.')
    s = cls()
    if key == "c4":
        for _ in range(6):
            s.ApplyAction(0)
    else:
        s.ApplyAction(0)
    with pytest.raises(ValueError, match="illegal move"):
        s.ApplyAction(0)
    assert type(cls().Winner()) is type(None)


class HashSearch(DynamicMCTS):
    """DynamicMCTS on the validation evaluator (what the golden fixtures were generated with)."""
    _EVALUATOR = _lib.EVAL_HASH
    salt = 0

    def _make_engine(self, game_id, n_slots, sims, **kw):
        return _lib.Engine(game_id, n_slots=n_slots, sims_per_move=max(int(sims), 1), mcts_kind=self._KIND,
                           max_depth=self._max_depth(), evaluator=_lib.EVAL_HASH, hash_salt=self.salt,
                           c_puct=float(self.ExplorationRate), **kw)


class HashFixed(FixedMCTS):
    _EVALUATOR = _lib.EVAL_HASH
    salt = 0
    _make_engine = HashSearch._make_engine


FILES = [f for f in sorted(os.path.basename(p) for p in glob.glob(
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mcts_*.npz")))
    if f.split("_")[1] in GAMES and "s800" not in f]


@pytest.mark.parametrize("fname", FILES)
def test_findmove_api_golden(golden_dir, fname, monkeypatch):
    g = np.load(os.path.join(golden_dir, fname), allow_pickle=False)
    cls = GAMES[fname.split("_")[1]]
    sims, seed, salt, max_depth, fixed, reuse = [int(x) for x in g["meta"]]
    c, temp = [float(x) for x in g["cfg"]]
    gs = g["game_start"]
    for gi in range(len(gs) - 1):
        if fixed:
            m = HashFixed(maxDepth=max_depth, explorationRate=c, playLimit=sims)
        else:
            m = HashSearch(explorationRate=c, playLimit=sims)
        m.salt = salt + gi
        state = cls()
        m.DropRoot()
        for i in range(gs[gi], gs[gi + 1]):
            assert state == make_state(cls, g["board"][i], g["player"][i], g["prev"][i])
            monkeypatch.setattr(np.random, "random_sample", lambda *a, _u=float(g["u"][i]): _u)
            nxt, v, prob = m.FindMove(state, temp)
            assert float(v) == g["v"][i]
            assert np.array_equal(prob, g["prob"][i])
            assert np.array_equal(m.Root.ChildPlays(), g["plays"][i])
            assert np.array_equal(m.Root.ChildWinRates(), g["winrates"][i])
            assert m.Root.Plays == g["root_plays"][i]
            want = state.Copy()
            want.ApplyAction(int(g["action"][i]))
            assert nxt == want and nxt.PreviousPlayer == state.Player
            state = nxt
            if reuse:
                m.MoveRoot(state)
            else:
                m.DropRoot()


@pytest.mark.parametrize("key", ["c4", "ttt"])
def test_resetroot_and_children_golden(golden_dir, key):
    """MCTS.ResetRoot (MCTS.py:214-225) and the Node graph below the root (MCTS.py:7-98), against the reference's own run
    (tests/make_golden.py gen_resetroot): FindMove + MoveRoot a few times at temp 0, ResetRoot -- the top-most ancestor's
    Plays are the sum of ALL simulations (the reference's _backProp walks through the ancestors above the current root) -- then
    down the played line through Children, then another FindMove on the tree ResetRoot left."""
    g = np.load(os.path.join(golden_dir, f"resetroot_{key}.npz"), allow_pickle=False)
    cls = GAMES[key]
    sims, moves, sims_after, salt = [int(x) for x in g["meta"]]
    m = HashSearch(explorationRate=0.85, playLimit=sims)
    m.salt = salt
    s = cls()
    for k in range(moves):
        nxt, _v, _p = m.FindMove(s, 0)
        want = s.Copy()
        want.ApplyAction(int(g["actions"][k]))
        assert nxt == want
        s = nxt
        m.MoveRoot(s)
    m.ResetRoot()
    node = m.Root
    assert node.State == cls() and node.Parent is None
    for depth in range(moves + 1):
        assert node.Plays == g[f"plays_{depth}"], depth
        assert np.float32(node.Value) == np.float32(g[f"value_{depth}"]), depth
        assert np.array_equal(node.ChildPlays(), g[f"child_plays_{depth}"]), depth
        assert np.array_equal(node.ChildWinRates(), g[f"child_winrates_{depth}"]), depth
        assert np.array_equal(node.LegalActions, g[f"legal_{depth}"]), depth
        assert [c is None for c in node.Children] == list(g[f"children_none_{depth}"]), depth
        if depth < moves:
            child = node.Children[int(g["actions"][depth])]
            assert child.Parent is node
            node = child
    nxt, v, prob = m.FindMove(cls(), 0, playLimit=sims_after)
    assert m.Root.Plays == g["after_plays"] and float(v) == float(g["after_v"])
    assert np.array_equal(m.Root.ChildPlays(), g["after_child_plays"])
    assert np.array_equal(m.Root.ChildWinRates(), g["after_child_winrates"])
    assert np.array_equal(prob, g["after_prob"])
    want = cls()
    want.ApplyAction(int(g["after_action"]))
    assert nxt == want
    # ResetRoot on a tree that never moved, and on no tree at all, changes nothing
    m.ResetRoot()
    assert m.Root.Plays == g["after_plays"]
    fresh = HashSearch(explorationRate=0.85, playLimit=4)
    fresh.ResetRoot()
    assert fresh.Root is None


def test_findmove_errors():
    m = HashSearch(explorationRate=0.85)
    s = Connect4.BoardState()
    with pytest.raises(TypeError):
        m.FindMove("not a state")
    with pytest.raises(ValueError, match="stop time"):
        m.FindMove(s)
    with pytest.raises(ValueError, match="NaN"):
        m.FindMove(s, 1.0, playLimit=1)  # one simulation on a fresh root
    m.FindMove(s, 1.0, playLimit=8)
    other = s.Copy()
    other.ApplyAction(3)
    with pytest.raises(AssertionError):
        m.FindMove(other, 1.0, playLimit=8)  # tree root is still `s`
    with pytest.raises(ValueError):
        FixedMCTS(maxDepth=0, explorationRate=1.0, playLimit=5)
    # time-limited search
    f = FixedMCTS(maxDepth=3, explorationRate=0.85, timeLimit=0.05)
    nxt, v, p = f.FindMove(Connect4.BoardState(), 0.1)
    assert abs(p.sum() - 1.0) < 1e-12 and f.Root.Plays >= 16


@pytest.mark.parametrize("key,shape", [("c4", (1, 6, 7, 3)), ("ttt", (1, 3, 3, 3))])
def test_selfplay_blobs_match_reference(golden_dir, key, shape):
    """The reference's GenerateTrainingSamples output, byte for byte: replay its games on the engine
    with the uniforms its np.random.choice calls consumed, serialise with our wire encoder."""
    cls = GAMES[key]
    game = cls.GAME_ID
    g = np.load(os.path.join(golden_dir, f"selfplay_{key}.npz"), allow_pickle=False)
    sims, seed, salt, n_games = [int(x) for x in g["meta"]]
    offs = g["game_off"]
    A = cls.LegalMoves
    blobs, pos = [], 0
    for ln in g["blob_len"]:
        blobs.append(g["blob"][pos:pos + ln].tobytes())
        pos += ln
    ucur = 0
    for gi in range(n_games):  # the reference plays its games one after another with one RNG stream
        eng = _lib.Engine(game, n_slots=1, sims_per_move=sims, evaluator=_lib.EVAL_HASH, hash_salt=salt, c_puct=0.85)
        eng.set_roots(_lib.game_initial(game))
        history = []
        winner = -1
        while winner < 0:
            root = eng.root_states()
            planes = _lib.game_encode(game, root)
            _b, player, _pv = _lib.unpack_grid(game, root)
            eng.run_sims(sims)
            out = eng.sample_moves(1.0, u=g["u"][ucur:ucur + 1])
            ucur += 1
            plays = out["child_plays"][0, :A].astype(np.float64)
            history.append(ExampleState(None, plays / plays.sum(), planes, player=int(player[0])))
            eng.move_roots(out["action"])
            winner = int(_lib.game_winner(game, eng.root_states())[0])
        root = eng.root_states()
        _b, player, _pv = _lib.unpack_grid(game, root)
        history.append(ExampleState(None, np.zeros(A), _lib.game_encode(game, root), player=int(player[0])))
        for ex in history:
            ex.MctsEval = 0 if winner == 0 else (1 if ex.Player == winner else -1)
        mine = [ex.SerializeState() for ex in history]
        assert mine == blobs[offs[gi]:offs[gi + 1]], gi
        eng.close()
    assert ucur == len(g["u"])


def test_generate_training_samples_drop_in(tmp_path, monkeypatch, orc):
    """Blackbird.GenerateTrainingSamples(model, nGames, temp) with a network-backed Model: one
    PutGames per game, examples well formed; ValueError for nGames <= 0."""
    monkeypatch.chdir(tmp_path)
    from blackbird_amd import Blackbird
    cfg = {"blocks": 2, "filters": 16, "eval": {"dense": 16}, "hasTeacher": False,
           "policy": {"dirichlet": {"alpha": 0.2, "epsilon": 0.3}}, "training": {"optimizer": "adam"}}
    model = Blackbird.Model(Connect4.BoardState, "drop", {"explorationRate": 0.85, "playLimit": 24}, cfg)
    with pytest.raises(ValueError, match="positive integer"):
        Blackbird.GenerateTrainingSamples(model, 0, 1.0)
    Blackbird.GenerateTrainingSamples(model, 5, 1.0)
    blobs = model.Conn.GetGames(model.Name, model.Version)
    exs = [ExampleState.FromSerialized(b) for b in blobs]
    terminal = [e for e in exs if not e.MctsPolicy.any()]
    assert len(terminal) == 5  # one terminal example (pi == 0) per game
    for e in exs:
        assert e.Board.shape == (1, 6, 7, 3) and e.MctsEval[0] in (-1.0, 0.0, 1.0)
        assert e.MctsPolicy.shape == (7,) and (abs(e.MctsPolicy.sum() - 1) < 1e-12 or not e.MctsPolicy.any())
    # Network API on the same model: value in (-1,1), policy sums to 1, noise differs call to call
    x = Connect4.BoardState().AsInputArray()
    v = model.getEvaluation(x)
    p1, p2 = model.getPolicy(x), model.getPolicy(x)
    assert v.dtype == np.float32 and -1 < v < 1 and abs(p1.sum() - 1) < 1e-5 and not np.array_equal(p1, p2)
    # FindMove / MoveRoot loop exactly as the reference's self-play loop drives it
    state = Connect4.BoardState()
    model.DropRoot()
    nxt, v, pi = model.FindMove(state, 1.0)
    assert abs(pi.sum() - 1) < 1e-12 and model.Root.Plays == 24
    model.MoveRoot(nxt)
    nxt2, v2, pi2 = model.FindMove(nxt, 1.0)
    assert model.Root.Plays > 24 - 1  # tree reuse: playLimit is added to the kept child's visits
    # train() moves the weights and the engine sees the new ones (f1 row)
    before = model.getEvaluation(x)
    Blackbird.TrainWithExamples(model, batchSize=8, learningRate=1e-2)
    assert model.Version == 2 and model.getEvaluation(x) != before
    model.Conn.Close()


def test_dragonchess_interface(golden_dir):
    """DragonChess.BoardState mirror: W,W,B turn order, castle flags, 17-plane encoding, edge cases of SURVEY 8a."""
    from blackbird_amd import DragonChess
    cls = DragonChess.BoardState
    g = np.load(os.path.join(golden_dir, "playouts_dc.npz"), allow_pickle=False)
    gs = g["game_start"]
    s = cls()
    assert s.Player == 1 and s.PreviousPlayer is None and s.LegalMoves == 4032 and s.GameType == "DragonChess"
    assert [cls.int_to_move[a] for a in np.where(s.LegalActions() == 1)[0]] == \
        ["4 3", "4 5", "11 19", "11 27", "12 20", "12 28", "13 21", "13 29"]
    for i in range(gs[0], gs[0] + 30):
        assert np.array_equal(s.board.astype(np.int8).ravel(), g["board"][i])
        assert s.Player == g["player"][i] and (s.PreviousPlayer or 0) == g["prev"][i]
        assert np.array_equal(s.AsInputArray().ravel(), g["enc"][i])
        assert np.array_equal(np.where(s.LegalActions() == 1)[0], g["legal_idx"][g["legal_off"][i]:g["legal_off"][i + 1]])
        w = s.Winner()
        assert (-1 if w is None else w) == g["win_none"][i]
        t = s.Copy()
        assert t == s and t.PreviousPlayer == s.PreviousPlayer  # DragonChess.Copy keeps PreviousPlayer
        t.ApplyAction(int(g["action"][i]))
        s = t
    # after White's 11->27, 12->28 it is Black's turn with 20 legal moves
    s = cls()
    s.ApplyAction(cls.move_to_int["11 27"])
    assert s.Player == 1 and s.PreviousPlayer == 1
    s.ApplyAction(cls.move_to_int["12 28"])
    assert s.Player == 2 and int(s.LegalActions().sum()) == 20
    with pytest.raises(ValueError, match="illegal move"):
        s.ApplyAction(cls.move_to_int["0 1"])
    # K x k is legal and wins
    k = cls()
    k.board[:] = 0
    k.board[3, 3] = 1
    k.board[4, 4] = -1
    assert k.LegalActions()[cls.move_to_int["27 36"]] == 1
    k.ApplyAction(cls.move_to_int["27 36"])
    assert k.Winner() == 1


def test_generate_training_samples_dragonchess(tmp_path, monkeypatch):
    """The drop-in path for the wide game: compact child lists scattered to the 4032-wide pi, 17-plane boards, the
    int8-wrapped policy dim of the reference's blobs (4032 -> -64, Blackbird.py:76-79), one PutGames per game."""
    monkeypatch.chdir(tmp_path)
    from blackbird_amd import Blackbird, DragonChess, proto_wire
    cfg = {"blocks": 1, "filters": 16, "eval": {"dense": 16}, "hasTeacher": False,
           "policy": {"dirichlet": {"alpha": 0.2, "epsilon": 0.3}}, "training": {"optimizer": "adam"}}
    model = Blackbird.Model(DragonChess.BoardState, "dc", {"explorationRate": 0.85, "playLimit": 6}, cfg)
    Blackbird.GenerateTrainingSamples(model, 3, 1.0)
    blobs = model.Conn.GetGames(model.Name, model.Version)
    assert len(blobs) >= 3 * 2
    n_term = 0
    for bts in blobs:
        f = proto_wire.decode_state(bts)
        assert np.frombuffer(f["boardDims"], dtype=np.int8).tolist() == [1, 8, 8, 17]
        assert np.frombuffer(f["policyDims"], dtype=np.int8).tolist() == [np.int64(4032).astype(np.int8)]
        pi = np.frombuffer(f["mctsPolicy"], dtype=np.float64)
        board = np.frombuffer(f["boardEncoding"], dtype=np.int8)
        assert pi.shape == (4032,) and board.shape == (8 * 8 * 17,)
        if pi.any():
            # visited children: at most playLimit new ones on top of those the re-used subtree brought along (its root had at
            # most playLimit - 1 visits in the previous search, the first of which expanded it)
            assert abs(pi.sum() - 1.0) < 1e-12 and (pi >= 0).all() and np.count_nonzero(pi) <= 2 * 6 - 2
        else:
            n_term += 1
    assert n_term == 3  # one terminal example (pi = 0) per game
