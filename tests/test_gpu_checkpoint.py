"""-m gpu: rows f4 (checkpoint interchange, Network.py:86-112) and the self-play entry's bookkeeping that the round-1
review found untested: saveModel -> loadModel -> identical weights and logits, LastVersion() loading what was saved,
fresh random streams on every GenerateTrainingSamples call, more games than slots for the wide game, and the capacity
errors of bb_create / bb_fit_slots."""
import os

import numpy as np
import pytest

from blackbird_amd import Blackbird, Connect4, DragonChess, _lib
from blackbird_amd import weights as W

pytestmark = pytest.mark.gpu
CFG = {"blocks": 2, "filters": 16, "eval": {"dense": 16}, "hasTeacher": False,
       "policy": {"dirichlet": {"alpha": 0.2, "epsilon": 0.3}}, "training": {"optimizer": "adam"}}


def _positions(n=9, seed=4):
    rng = np.random.RandomState(seed)
    cells = rng.randint(0, 3, size=(n, 6, 7))
    b = np.zeros((n, 6, 7, 3), dtype=np.int8)
    b[..., 0] = cells == 1
    b[..., 1] = cells == 2
    b[..., 2] = rng.choice([-1, 1], size=(n, 1, 1))
    return b


def _logits(weights, planes):
    eng = _lib.Engine(_lib.GAME_CONNECT4, n_slots=2, sims_per_move=2, evaluator=_lib.EVAL_NET)
    eng.load_weights(W.flatten(weights))
    out = eng.net_eval(planes=planes)
    eng.close()
    return out


def test_save_load_and_last_version_round_trip(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    model = Blackbird.Model(Connect4.BoardState, "ckpt", {"explorationRate": 0.85, "playLimit": 8}, CFG)
    assert os.path.isfile(os.path.join("blackbird_models", "ckpt_1", "best.npz"))  # Network.__init__ saves a new model
    w_init = {k: v.copy() for k, v in model._weights.items()}
    assert sorted(w_init) == sorted(W.variable_names(2))  # the reference graph's variable names
    planes = _positions()
    base = _logits(w_init, planes)
    # one training step moves the weights in memory; the file on disk still holds the initial ones
    rng = np.random.RandomState(0)
    model.train(planes[:8], rng.choice([-1.0, 1.0], 8), rng.dirichlet(np.ones(7), 8), 1e-2)
    w_trained = {k: v.copy() for k, v in model._weights.items()}
    assert any(not np.array_equal(w_trained[k], w_init[k]) for k in w_init)
    # LastVersion(): a second Model built from the same arguments -> loads blackbird_models/ckpt_1 = the INITIAL weights
    old = model.LastVersion()
    assert old.Version == model.Version and old.Name == model.Name
    for k in w_init:
        assert np.array_equal(old._weights[k], w_init[k]) and old._weights[k].dtype == np.float32, k
    x = planes[3:4]
    v_old, p_clean = old.getEvaluation(x), _logits(w_init, x)
    assert v_old == p_clean[0][0] == base[0][3]
    # saveModel(name) -> loadModel(name): bit-identical variables, bit-identical network outputs
    model.saveModel(model._saveName)
    other = Blackbird.Model(Connect4.BoardState, "ckpt", {"explorationRate": 0.85, "playLimit": 8}, CFG)
    for k in w_trained:
        assert np.array_equal(other._weights[k], w_trained[k]), k
    a, b = _logits(w_trained, planes), _logits(other._weights, planes)
    for s, t in zip(a, b):
        assert np.array_equal(s, t)
    assert not np.array_equal(a[1], base[1])
    assert other.getEvaluation(x) == a[0][3] == model.getEvaluation(x)
    # loadModel of a name that was never saved reports False and leaves the weights alone (Network.py:103-106)
    assert other.loadModel("no_such_model") is False
    for k in w_trained:
        assert np.array_equal(other._weights[k], w_trained[k])
    # a saved name loads over the current weights (ckpt_1 now holds the trained ones) and drops the stale trainer
    old._trainer = object()
    assert old.loadModel("ckpt_1") is True and old._trainer is None
    for k in w_trained:
        assert np.array_equal(old._weights[k], w_trained[k])
    assert old.getEvaluation(x) == a[0][3]  # ... and the engine behind getEvaluation has them too
    for m in (model, old, other):
        m.Conn.Close()


def _games(conn, name, version):
    return conn.GetGames(name, version)


def test_generate_training_samples_draws_fresh_streams(tmp_path, monkeypatch):
    """a second call with unchanged weights must not replay the first one's games."""
    monkeypatch.chdir(tmp_path)
    np.random.seed(11)
    model = Blackbird.Model(Connect4.BoardState, "fresh", {"explorationRate": 0.85, "playLimit": 16}, CFG)
    Blackbird.GenerateTrainingSamples(model, 6, 1.0)
    first = list(_games(model.Conn, model.Name, model.Version))
    eng = model._batch_engine
    Blackbird.GenerateTrainingSamples(model, 6, 1.0)
    assert model._batch_engine is eng  # the engine is reused ...
    both = list(_games(model.Conn, model.Name, model.Version))
    second = both[len(first):]
    assert len(second) >= 6 * 8 and second != first[:len(second)]  # ... but the games are new ones
    assert set(second) != set(first)
    assert model._games_played == 12
    model.Conn.Close()


def test_more_dragonchess_games_than_slots_and_slot_cap(tmp_path, monkeypatch):
    """the number of concurrent slots is capped by what the device holds; extra games queue on the slots."""
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(Blackbird, "MAX_CONCURRENT_GAMES", 3)
    cfg = dict(CFG, blocks=1)
    model = Blackbird.Model(DragonChess.BoardState, "dcq", {"explorationRate": 0.85, "playLimit": 4}, cfg)
    Blackbird.GenerateTrainingSamples(model, 7, 1.0)
    assert model._batch_engine.n_slots == 3
    blobs = model.Conn.GetGames(model.Name, model.Version)
    from blackbird_amd import proto_wire
    terminal = sum(1 for b in blobs if not np.frombuffer(proto_wire.decode_state(b)["mctsPolicy"], dtype=np.float64).any())
    assert terminal == 7  # every one of the 7 games was played to its end (or to the ply cap) on 3 slots
    model.Conn.Close()


def test_capacity_is_reported_not_crashed():
    game = _lib.GAME_DRAGONCHESS
    # BASELINE configs[3] fits one MI355X ...
    fit, per_slot = _lib.fit_slots(game, 1024, 400)
    assert fit == 1024 and 100e6 < per_slot < 250e6
    # ... ten times the slots do not: bb_fit_slots says how many would, bb_create refuses before allocating anything
    fit, _ = _lib.fit_slots(game, 10240, 400)
    assert 1024 <= fit < 10240
    with pytest.raises(_lib.BlackbirdHipError, match="do not fit"):
        _lib.Engine(game, n_slots=10240, sims_per_move=400, evaluator=_lib.EVAL_HASH)
    with pytest.raises(ValueError):
        _lib.fit_slots(game, 0, 400)
    # and a slot that cannot exist at all is BB_ERR_CAPACITY too
    with pytest.raises(_lib.BlackbirdHipError, match="not even one"):
        _lib.fit_slots(game, 4, 400, max_games=1 << 20)  # a 0.5 TB example store


def test_time_limited_batched_selfplay(tmp_path, monkeypatch):
    """mcts.timeLimit without playLimit (MCTS.py:173-182): every move all games are searched together for the budget; the
    examples are the reference's: per ply (planes, visit distribution over legal moves, side to move), a terminal example with
    pi = 0, z by the winner, one PutGames per game."""
    monkeypatch.chdir(tmp_path)
    from blackbird_amd import TicTacToe, proto_wire
    model = Blackbird.Model(TicTacToe.BoardState, "timed", {"explorationRate": 0.85, "timeLimit": 0.02}, dict(CFG, blocks=1))
    assert model.PlayLimit is None
    np.random.seed(5)
    Blackbird.GenerateTrainingSamples(model, 5, 1.0)
    blobs = model.Conn.GetGames(model.Name, model.Version)
    exs = [Blackbird.ExampleState.FromSerialized(b) for b in blobs]
    terminal = [e for e in exs if not e.MctsPolicy.any()]
    assert len(terminal) == 5 and 5 * 6 <= len(exs) <= 5 * 10
    for e in exs:
        assert e.Board.shape == (1, 3, 3, 3) and e.MctsEval[0] in (-1.0, 0.0, 1.0)
        if e.MctsPolicy.any():
            assert abs(e.MctsPolicy.sum() - 1.0) < 1e-12
            empty = (e.Board[0, :, :, 0] == 0) & (e.Board[0, :, :, 1] == 0)
            assert (e.MctsPolicy.reshape(3, 3)[~empty] == 0).all()   # visits only on legal (empty) cells
    assert model._games_played == 5
    model.Conn.Close()


def test_dragonchess_getpolicy_mixes_noise():
    """Network.getPolicy for the wide game must mix the graph's Beta noise like the other games do."""
    game = _lib.GAME_DRAGONCHESS
    eng = _lib.Engine(game, n_slots=2, sims_per_move=2, evaluator=_lib.EVAL_NET, alpha=0.2, epsilon=0.3, seed=4, max_plies=8)
    eng.load_weights(W.flatten(W.init_weights(17, 16, 1, 16, 4032, seed=1)))
    st = _lib.game_initial(game)
    clean = eng.net_eval(states=st)[2][0].astype(np.float64)
    n1 = eng.net_eval(states=st, noise=1)[2][0].astype(np.float64)
    n2 = eng.net_eval(states=st, noise=2)[2][0].astype(np.float64)
    assert abs(clean.sum() - 1) < 1e-4 and abs(n1.sum() - 1) < 1e-4 and not np.array_equal(n1, n2)
    # undo the mix: q = (1-eps) p + eps x, noisy = q / sum(q); the x are Beta(0.2, 0.8) draws: mean 0.2, in (0, 1)
    x = (n1 * ((1 - 0.3) + 0.3 * 4032 * 0.2) - 0.7 * clean) / 0.3        # with sum(q) ~ 0.7 + 0.3 * A * alpha
    assert 0.15 < x.mean() < 0.25 and (x > -0.05).all()
    eng.close()
