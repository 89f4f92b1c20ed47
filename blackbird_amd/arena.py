"""Batched head-to-head evaluation: Blackbird.TestModels (Blackbird.py:177-216) for many games at once.

The reference plays one game at a time: a coin decides who starts, the player to move calls FindMove, BOTH players
call MoveRoot with the new state (tree reuse), until `state.Winner()` is set; the result is +1 / 0 / -1 from
model1's point of view.  Here all `numTests` games advance one ply per iteration: each searcher owns one engine with one
slot per game, searches exactly the slots where it is to move (`bb_run_sims_masked`: the other slots' trees are not
touched, as in the reference, where a model only searches on its own turns), and follows every move with MoveRoot.

Per game the sequence (FindMove on my turns, MoveRoot after every move) is the reference's; what differs from a serial
run is only the order in which random numbers are consumed across games, so results are identical for deterministic
settings (temp = 0, no prior noise) and identically distributed otherwise (tests/test_gpu_arena.py)."""
import random
from time import time

import numpy as np

from . import _lib
from .RandomMCTS import RandomMCTS


class _Searcher(object):
    """One side of the arena: an MCTS / FixedMCTS / Model object turned into an engine with one slot per game."""

    def __init__(self, player, game_id, n_games, sims, seconds):
        self.player = player
        self.random = isinstance(player, RandomMCTS)
        self.engine = None
        self.sims = None if sims is None else int(sims)   # simulations per move (playLimit), or
        self.seconds = seconds                             # wall-clock budget per move (timeLimit) when sims is None
        if not self.random:
            plies = {_lib.GAME_CONNECT4: 43, _lib.GAME_TICTACTOE: 10}.get(game_id, 64)
            cap = player._MAX_NODES
            if self.sims is not None:
                cap = min(cap, self.sims * plies * player._max_depth() + 64)
            self.engine = player._make_engine(game_id, n_games, self.sims or 64, node_capacity=cap)
            player._after_engine_created(self.engine)  # a Model loads its weights here
            self.primed = np.zeros(n_games, dtype=bool)

    def search(self, mask):
        """_runMCTS (MCTS.py:284-303) on the slots of `mask`: playLimit simulations, or simulations in chunks until the
        move's wall-clock budget is spent (at least one chunk, so the root is always expanded)."""
        if self.sims is not None:
            self.engine.run_sims(self.sims, mask=mask)
            return
        end = time() + self.seconds
        while True:
            self.engine.run_sims(16, mask=mask)
            self.engine.synchronize()
            if time() >= end:
                break

    def close(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None


def TestModelsBatched(model1, model2, temp, numTests, playLimit=None, first=None, uniforms=None):
    """Play `numTests` games of model1 against model2 concurrently; returns an int array of +1 / 0 / -1 (model1's
    wins / draws / losses), one entry per game, in game order.

    playLimit  simulations per move for both sides (default: each side's own PlayLimit; a side that only has a
               TimeLimit searches all its games together until that many seconds have passed, every move)
    first      optional bool array: model1 moves first in game i (default: `random.choice([True, False])` per game,
               drawn in game order like the reference does at the top of each game)
    uniforms   optional callable n -> float64[n] supplying np.random.choice's uniforms (default np.random.random_sample)"""
    if numTests <= 0:
        raise ValueError('Use a positive integer for number of tests.')
    game = model1.Game
    game_id = game.GAME_ID
    info = _lib.game_info(game_id)
    if first is None:
        first = np.array([random.choice([True, False]) for _ in range(numTests)], dtype=bool)
    first = np.asarray(first, dtype=bool)
    draw = uniforms if uniforms is not None else np.random.random_sample
    sides = []
    for m in (model1, model2):
        # FindMove's stop rule (MCTS.py:173-182): playLimit simulations and / or timeLimit seconds per move
        sims = playLimit if playLimit is not None else getattr(m, 'PlayLimit', None)
        seconds = getattr(m, 'TimeLimit', None)
        if sims is None and seconds is None and not isinstance(m, RandomMCTS):
            raise ValueError('Not enough information to decide a stop time.')
        sides.append(_Searcher(m, game_id, numTests, sims, seconds))
    try:
        states = np.repeat(_lib.game_initial(game_id), numTests, axis=0)   # packed boards, host side
        alive = np.ones(numTests, dtype=bool)
        result = np.zeros(numTests, dtype=np.int32)
        to_move1 = first.copy()
        model1_player = np.where(first, 1, 2)
        while alive.any():
            actions = np.full(numTests, -1, dtype=np.int32)
            for k, side in enumerate(sides):
                mine = alive & (to_move1 if k == 0 else ~to_move1)
                if not mine.any():
                    continue
                idx = np.nonzero(mine)[0]
                if side.random:                                   # RandomMCTS.FindMove: a uniformly random legal move
                    legal = _lib.game_legal(game_id, states[idx])
                    for i, row in zip(idx, legal):
                        actions[i] = int(np.random.choice(np.nonzero(row)[0]))
                    continue
                eng = side.engine
                fresh = idx[~side.primed[idx]]                    # FindMove's `Root is None` branch
                if len(fresh):
                    eng.set_roots(states[fresh], slots=fresh, game_ids=fresh)
                    side.primed[fresh] = True
                side.search(mine)
                if eng.counters()['overflow']:
                    raise _lib.BlackbirdHipError('search tree outgrew the node pool')
                u = np.zeros(numTests, dtype=np.float64)
                if temp != 0:
                    u[idx] = draw(len(idx))
                out = eng.sample_moves(temp, u if temp != 0 else None)
                if (out['action'][idx] < 0).any():
                    raise ValueError('probabilities contain NaN')
                actions[idx] = out['action'][idx]
            # apply the moves, then both sides follow with MoveRoot (Blackbird.py:198-200)
            idx = np.nonzero(alive)[0]
            new_states, status = _lib.game_apply(game_id, states[idx], actions[idx])
            if (status != 0).any():
                raise ValueError('Tried to make an illegal move.')
            states[idx] = new_states
            for side in sides:
                if side.random:
                    continue
                mv = np.where(alive & side.primed, actions, -1).astype(np.int32)
                side.engine.move_roots(mv)
            winners = _lib.game_winner(game_id, states[idx])   # state.Winner(): full scan, as Blackbird.py:202
            to_move1 = ~to_move1
            for i, w in zip(idx, winners):
                if w >= 0:                                         # Winner() is not None
                    alive[i] = False
                    result[i] = 0 if w == 0 else (1 if w == model1_player[i] else -1)
        return result
    finally:
        for side in sides:
            side.close()
