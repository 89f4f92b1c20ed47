"""-m gpu: the batched arena (blackbird_amd/arena.py) against the serial TestModels mirror
(Blackbird.py:177-216 semantics: FindMove on my turns, MoveRoot by both sides after every move)."""
import random

import numpy as np
import pytest

from blackbird_amd import Blackbird, Connect4, TicTacToe
from blackbird_amd.RandomMCTS import RandomMCTS
from blackbird_amd.arena import TestModelsBatched

pytestmark = pytest.mark.gpu


def _model(game, name, tmp_seed, play_limit, eps=0.0):
    cfg = {"blocks": 2, "filters": 16, "eval": {"dense": 16}, "hasTeacher": False,
           "policy": {"dirichlet": {"alpha": 0.2, "epsilon": eps}}, "training": {"optimizer": "adam"}}
    np.random.seed(tmp_seed)  # weight initialisation draws from numpy's stream
    return Blackbird.Model(game, name, {"explorationRate": 0.85, "playLimit": play_limit}, cfg)


def _serial(model1, model2, temp, first):
    """TestModels with the coin flips replaced by `first` (same loop as Blackbird.TestModels)."""
    out = []
    for f in first:
        model1ToMove = bool(f)
        model1Player = 1 if model1ToMove else 2
        model1.DropRoot()
        model2.DropRoot()
        state = model1.Game()
        winner = None
        while winner is None:
            (state, *_) = (model1 if model1ToMove else model2).FindMove(state, temp)
            model1.MoveRoot(state)
            model2.MoveRoot(state)
            model1ToMove = not model1ToMove
            winner = state.Winner()
        out.append(1 if winner == model1Player else (0 if winner == 0 else -1))
    return np.array(out)


@pytest.mark.parametrize("game", [Connect4.BoardState, TicTacToe.BoardState])
def test_batched_arena_equals_serial_when_deterministic(tmp_path, monkeypatch, game):
    """temp = 0 and epsilon = 0: no random number is consumed, so every game must end exactly as in the serial loop."""
    monkeypatch.chdir(tmp_path)
    m1, m2 = _model(game, "a", 1, 24), _model(game, "b", 2, 24)
    first = np.array([True, False, True, True, False, False, True, False])
    serial = _serial(m1, m2, 0, first)
    batched = TestModelsBatched(m1, m2, 0, len(first), first=first)
    assert np.array_equal(serial, batched)
    assert set(np.unique(batched)) <= {-1, 0, 1}


def test_batched_arena_against_random_player_and_coin(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    m1 = _model(Connect4.BoardState, "a", 3, 48, eps=0.3)
    rnd = RandomMCTS()
    random.seed(7)
    np.random.seed(7)
    res = TestModelsBatched(m1, rnd, 0.5, 64)
    assert res.shape == (64,) and set(np.unique(res)) <= {-1, 0, 1}
    assert (res == 1).sum() > (res == -1).sum()  # 48 simulations per move beat uniformly random moves
    with pytest.raises(ValueError):
        TestModelsBatched(m1, rnd, 0.5, 0)
