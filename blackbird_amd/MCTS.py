"""Tree search front end: the reference's MCTS API (/root/reference/src/MCTS.py:101-225) on top of
the HIP node-pool engine.  FindMove = prime/verify the root (host), `playLimit` simulations
(k_tree_step + evaluator kernels), root statistics + move choice (k_sample); MoveRoot/DropRoot map to
bb_move_roots / bb_set_roots.  No search arithmetic happens on the host.

Differences a user can observe (documented in DESIGN.md):
  * `Root` is a read-only view of the engine's node rows, not a linked Node graph: `Root.Children[i]` builds the child's view
    on demand (bb_node_view; as deep as the caller walks) and `Parent` is only set on views reached that way;
  * GetPriors/SampleValue are not per-call Python hooks: the evaluator is chosen per class
    (base/Fixed: priors = ones + random rollouts, MCTS.py:346-383; Model: the network);
  * DragonChess: ResetRoot re-primes the tree at the position the search started from (statistics are dropped) and Children
    stay None -- the dense-action games keep the reference's behaviour (bb_reset_roots: the root goes back to its top-most
    ancestor with every simulation's statistics, MCTS.py:214-225).
"""
from time import time

import numpy as np

from . import _lib
from .GameState import GameState


def _seed_from_numpy():
    """Engine RNG seed derived from numpy's global generator WITHOUT advancing it, so that
    np.random.seed() makes rollouts reproducible and FindMove consumes exactly the one uniform
    np.random.choice consumes in the reference."""
    key = np.random.get_state()[1]
    return int(key[0]) ^ (int(key[1]) << 16) ^ int(np.random.get_state()[2])


class _Children(object):
    """Node.Children of an expanded node (MCTS.py:31-41, :122-139): None at illegal actions, a Node at every legal one.  Built
    on demand: a child that a simulation has reached is a view of its engine row, one that none has reached yet is what the
    reference's eager AddChildren would hold -- the position after the move, no plays, no children."""

    def __init__(self, parent, rows):
        self._parent, self._rows, self._cache = parent, rows, {}

    def __len__(self):
        return len(self._parent.LegalActions)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        if self._parent.LegalActions[i] != 1:
            return None
        if i not in self._cache:
            p, r = self._parent, self._rows
            plays, value = int(r['plays'][i]), np.float32(r['value'][i])
            if r['child'][i] >= 0 and p._engine is not None:
                c = Node._from_engine(p._engine, p._cls, int(r['child'][i]) & 0x3FFFFFFF, plays, value)
            else:
                st = p.State.Copy()
                st.ApplyAction(i)
                c = Node(st, st.LegalActions(), plays, value, np.zeros(len(self)), np.zeros(len(self)), False)
            c.Parent = p
            self._cache[i] = c
        return self._cache[i]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Node(object):
    """Read-only view of one node's statistics with the reference Node's accessors (MCTS.py:7-98)."""

    def __init__(self, state, legalActions, plays, value, childPlays, childValues, expanded):
        self.State = state
        self.LegalActions = np.array(legalActions)
        self.Plays = plays
        self.Value = value
        self.Children = [None] * len(legalActions) if expanded else None
        self.Parent = None
        self._engine = self._cls = None
        self._childPlays = np.asarray(childPlays, dtype=np.float64)
        n = np.asarray(childPlays, dtype=np.float32)
        with np.errstate(divide='ignore', invalid='ignore'):
            wr = np.where(n > 0, np.asarray(childValues, dtype=np.float32) / np.maximum(n, 1), 0)
        self._childWinRates = wr.astype(np.float64)

    @classmethod
    def _from_engine(cls, engine, state_cls, node, plays, value, slot=0):
        """The node at pool index `node` (-1: the root) of the engine's slot; Plays / Value are its parent's record of it."""
        r = engine.node_view(slot, node)
        A = engine.info.A
        state = state_cls._from_packed(r['state'])
        expanded = bool(r['flags'] & 1)
        legal = np.array([(r['legal_mask'] >> a) & 1 for a in range(A)], dtype=np.float64) if expanded else state.LegalActions()
        n = cls(state, legal, plays, value, r['plays'][:A] * legal, r['value'][:A] * legal, expanded)
        n._engine, n._cls = engine, state_cls
        if expanded:
            n.Children = _Children(n, r)
        return n

    def WinRate(self):
        return self.Value / self.Plays if self.Plays > 0 else 0

    def ChildProbability(self):
        allPlays = self._childPlays.sum()
        return self._childPlays / allPlays if allPlays > 0 else np.zeros(len(self._childPlays))

    def ChildWinRates(self):
        return self._childWinRates

    def ChildPlays(self):
        return self._childPlays


class MCTS(object):
    """Base class (MCTS.py:101-120).  Evaluator: uniform priors + random rollouts."""
    _KIND = _lib.MCTS_DYNAMIC
    _EVALUATOR = _lib.EVAL_ROLLOUT
    _MAX_NODES = 1 << 20

    def __init__(self, explorationRate, timeLimit=None, playLimit=None, **kwargs):
        self.TimeLimit = timeLimit
        self.PlayLimit = playLimit
        self.ExplorationRate = explorationRate
        self.Root = None
        self._engine = None
        self._root_state = None   # host mirror of the engine's root position
        self._first_state = None
        self._root_sims = 0       # simulations run since the root was primed
        self._moves = 0
        self._anc_depth = 0       # MoveRoot steps since the tree was primed

    # ---- engine plumbing ---------------------------------------------------------------------------
    def _max_depth(self):
        return 1

    def _make_engine(self, game_id, n_slots, sims, **kw):
        return _lib.Engine(game_id, n_slots=n_slots, sims_per_move=max(int(sims), 1), mcts_kind=self._KIND,
                           max_depth=self._max_depth(), evaluator=self._EVALUATOR, c_puct=float(self.ExplorationRate),
                           seed=_seed_from_numpy(), **kw)

    def _ensure_engine(self, state):
        if self._engine is None or self._engine.game != state.GAME_ID:
            plies = {_lib.GAME_CONNECT4: 43, _lib.GAME_TICTACTOE: 10}.get(state.GAME_ID, 64)
            cap = self._MAX_NODES if self.PlayLimit is None else min(self._MAX_NODES,
                                                                     int(self.PlayLimit) * plies * self._max_depth() + 64)
            self._engine = self._make_engine(state.GAME_ID, 1, self.PlayLimit or 64, node_capacity=cap,
                                             track_ancestors=state.GAME_ID != _lib.GAME_DRAGONCHESS)
            self._after_engine_created(self._engine)
        return self._engine

    def _after_engine_created(self, engine):
        pass

    # ---- reference API ---------------------------------------------------------------------------------
    def DropRoot(self):
        """MCTS.py:141-144"""
        self.Root = None
        self._root_state = None

    def FindMove(self, state, temp=0.1, moveTime=None, playLimit=None):
        """MCTS.py:146-199.  Returns (next state, Root.WinRate(), Root.ChildProbability())."""
        if not isinstance(state, GameState):
            raise TypeError('State not of type GameState')
        endTime = None
        if moveTime is None:
            moveTime = self.TimeLimit
        if moveTime is not None:
            endTime = time() + moveTime
        if playLimit is None:
            playLimit = self.PlayLimit
        if endTime is None and playLimit is None:
            raise ValueError('Not enough information to decide a stop time.')

        eng = self._ensure_engine(state)
        if self._root_state is None:
            eng.set_roots(state._packed(), slots=[0], game_ids=[self._moves])
            self._root_state = state
            self._first_state = state   # the top-most ancestor of whatever this tree grows into (ResetRoot)
            self._anc_depth = 0
            self._root_sims = 0
        assert self._root_state == state, 'Primed for the correct input state.'

        # _runMCTS (MCTS.py:284-303): stop on either limit; keep going while the root is unexpanded
        done = 0
        while True:
            if playLimit is not None and done >= playLimit:
                break
            if endTime is not None and time() >= endTime and self._root_sims > 0:
                break
            chunk = 16 if endTime is not None else playLimit - done
            if playLimit is not None:
                chunk = min(chunk, playLimit - done)
            eng.run_sims(chunk)
            done += chunk
            self._root_sims += chunk
        if eng.counters()['overflow']:
            raise _lib.BlackbirdHipError('search tree outgrew the node pool; raise node capacity')

        u = np.array([np.random.random_sample()]) if temp != 0 else None  # np.random.choice's one draw
        out = eng.sample_moves(temp, u)
        A = eng.info.A
        action = int(out['action'][0])
        self.Root = self._root_view(self._root_state, state.LegalActions(), out, True)
        if action < 0:
            raise ValueError('probabilities contain NaN')
        nextState = state.Copy()
        nextState.ApplyAction(action)
        winrate = out['root_winrate'][0] if out['root_plays'][0] > 0 else 0
        return (nextState, winrate, self.Root.ChildProbability())

    def MoveRoot(self, state):
        """MCTS.py:201-212 / _moveRoot :260-282"""
        if self._root_state is None:
            return
        if self._root_sims == 0:  # Root.Children is None -> Root = None
            self.DropRoot()
            return
        legal = np.where(self._root_state.LegalActions() == 1)[0]
        for a in legal:
            child = self._root_state.Copy()
            child.ApplyAction(int(a))
            if child == state:
                self._engine.move_roots([int(a)])
                self._root_state = child
                self._moves += 1
                self._anc_depth += 1
                out = self._engine.sample_moves(0.0)
                A = self._engine.info.A
                self._root_sims = int(out['root_plays'][0])
                self.Root = self._root_view(child, child.LegalActions(), out, out['action'][0] != -3)
                return

    def _root_view(self, state, legal, out, expanded):
        """Root as a Node view from bb_sample_moves' outputs (+ lazy Children for the dense-action games)."""
        A = self._engine.info.A
        plays = int(out['root_plays'][0])
        value = np.float32(out['root_winrate'][0]) * np.float32(max(plays, 1))
        root = Node(state, legal, plays, value, out['child_plays'][0, :A], out['child_value'][0, :A], expanded)
        if expanded and self._engine.info.dense:
            root._engine, root._cls = self._engine, type(state)
            root.Children = _Children(root, self._engine.node_view(0, -1))
        return root

    def ResetRoot(self):
        """MCTS.py:214-225: Root goes back to its top-most ancestor -- the first position this tree searched -- and keeps
        everything the simulations since have added (the reference's _backProp recurses through the ancestors above the
        current root, :252-258; the engine does the same for engines created with track_ancestors)."""
        if self._root_state is None or self.Root is None:
            return
        if self._engine is None or not self._engine.info.dense:   # DragonChess: statistics are dropped (module docstring)
            first = self._first_state
            self.DropRoot()
            self._first_state = first
            return
        self._engine.reset_roots()
        if self._anc_depth:
            self._root_state = self._first_state
            self._anc_depth = 0
        out = self._engine.sample_moves(0.0)
        self._root_sims = int(out['root_plays'][0])
        self.Root = self._root_view(self._root_state, self._root_state.LegalActions(), out, out['action'][0] != -3)

    def _applyAction(self, state, action):
        s = state.Copy()
        s.ApplyAction(action)
        return s

    def GetPriors(self, state):
        """MCTS.py:346-358"""
        return np.array([1] * len(state.LegalActions()))

    def __getstate__(self):
        d = self.__dict__.copy()
        d['_engine'] = None
        return d
