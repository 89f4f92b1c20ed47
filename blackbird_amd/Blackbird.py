"""Orchestration layer: mirror of /root/reference/src/Blackbird.py.

GenerateTrainingSamples keeps the reference's signature and side effects (one Conn.PutGames per
game, examples in ply order plus the terminal example, Blackbird.py:219-268) but plays all games
concurrently on the GPU: every tree, every leaf evaluation and every move runs in the HIP engine
(bb_selfplay_*); the host only turns finished example records into protobuf blobs.
"""
from collections import defaultdict

import numpy as np

from . import _lib
from . import proto_wire
from . import weights as W
from .MCTS import _seed_from_numpy
from .DataManager import Connection
from .DynamicMCTS import DynamicMCTS as MCTS
from .FixedMCTS import FixedMCTS
from .Network import Network
from .NetworkFactory import NetworkFactory
from .RandomMCTS import RandomMCTS

MAX_CONCURRENT_GAMES = 4096


class ExampleState(object):
    """Blackbird.py:21-81"""

    def __init__(self, evaluation, policy, board, player=None):
        self.MctsPolicy = policy
        self.MctsEval = evaluation
        self.Board = board
        self.Player = player

    @classmethod
    def FromSerialized(cls, serialState):
        f = proto_wire.decode_state(serialState)
        boardDims = np.frombuffer(f['boardDims'], dtype=np.int8)
        policyDims = np.frombuffer(f['policyDims'], dtype=np.int8)
        mctsEval = f['mctsEval'],  # the reference returns a 1-tuple here (trailing comma, Blackbird.py:60)
        mctsPolicy = np.frombuffer(f['mctsPolicy'], dtype=np.float64)
        if policyDims.size and int(np.prod(policyDims.astype(np.int64))) == mctsPolicy.size:
            mctsPolicy = mctsPolicy.reshape(policyDims)
        board = np.frombuffer(f['boardEncoding'], dtype=np.int8).reshape(boardDims)
        return cls(mctsEval, mctsPolicy, board)

    def SerializeState(self):
        pol = np.ascontiguousarray(self.MctsPolicy, dtype=np.float64)
        # the reference stores dims as int8 (Blackbird.py:76-79): 4032 wraps exactly like old numpy did
        pdims = np.array(pol.shape, dtype=np.int64).astype(np.int8)
        return proto_wire.encode_state(0.0 if self.MctsEval is None else self.MctsEval, pol.tobytes(),
                                       np.ascontiguousarray(self.Board, dtype=np.int8).tobytes(),
                                       np.array(self.Board.shape, dtype=np.int8).tobytes(), pdims.tobytes())


def _records_to_examples(game_cls, rec):
    """Engine example records -> ExampleState list (s = AsInputArray planes, pi = visits / total, z)."""
    gi = _lib.game_info(game_cls.GAME_ID)
    st = np.ascontiguousarray(rec['state']).view(_lib.STATE_DTYPE[game_cls.GAME_ID]).reshape(len(rec), -1)
    planes = _lib.game_encode(game_cls.GAME_ID, st)
    out = []
    for i in range(len(rec)):
        tot = float(rec['total'][i])
        visits = rec['visits'][i, :gi.A].astype(np.float64)
        pi = visits / tot if tot > 0 else np.zeros(gi.A)
        out.append(ExampleState(float(rec['z'][i]), pi, planes[i:i + 1], player=int(rec['player'][i])))
    return out


def _timed_selfplay(model, nGames, temp):
    """GenerateTrainingSamples when the searcher has a wall-clock limit per move (mcts.timeLimit, MCTS.py:173-182, :298):
    the reference's loop (FindMove, example, MoveRoot, Winner) for all games of a wave at once through the lock-step entry
    points -- every move, all live games are searched together until `TimeLimit` seconds have passed (and no further than
    PlayLimit simulations when that is set too), then each samples its move with numpy's next uniform."""
    from time import time
    game_cls = model.Game
    game = game_cls.GAME_ID
    gi = _lib.game_info(game)
    max_plies = {_lib.GAME_CONNECT4: 42, _lib.GAME_TICTACTOE: 9}.get(game, 512)
    cap = model._MAX_NODES
    n_slots, _ = _lib.fit_slots(game, min(nGames, MAX_CONCURRENT_GAMES), 64, node_capacity=cap)
    eng = model._make_engine(game, n_slots, 64, node_capacity=cap)
    model._after_engine_created(eng)
    eng.set_rng_stream(int(np.random.randint(0, 2 ** 62, dtype=np.int64)), model._games_played)
    try:
        for first in range(0, nGames, n_slots):
            n = min(n_slots, nGames - first)
            idx = np.arange(n)
            states = np.repeat(_lib.game_initial(game), n, axis=0)
            eng.set_roots(states, slots=idx, game_ids=model._games_played + first + idx)
            alive = np.ones(n_slots, dtype=bool)
            alive[n:] = False
            history = [[] for _ in range(n)]       # per game: (planes, pi, player)
            winner = np.full(n, -1)
            for _ply in range(max_plies):
                live = np.nonzero(alive)[0]
                if not len(live):
                    break
                end, done = time() + model.TimeLimit, 0
                while True:                        # _runMCTS: at least one chunk, so that every root is expanded
                    eng.run_sims(16, mask=alive)
                    eng.synchronize()
                    done += 16
                    if time() >= end or (model.PlayLimit is not None and done >= model.PlayLimit):
                        break
                if eng.counters()['overflow']:
                    raise _lib.BlackbirdHipError('search tree outgrew the node pool')
                u = np.zeros(n_slots)
                u[live] = np.random.random_sample(len(live))
                out = eng.sample_moves(temp, u if temp != 0 else None)
                planes = _lib.game_encode(game, states[live])
                acts = np.full(n_slots, -1, dtype=np.int32)
                for k, gidx in enumerate(live):
                    if out['action'][gidx] < 0:
                        raise ValueError('probabilities contain NaN')
                    plays = out['child_plays'][gidx].astype(np.float64)
                    pi = np.zeros(gi.A)
                    if gi.dense:
                        pi[:] = plays[:gi.A]
                    else:
                        nch = int((out['child_action'][gidx] >= 0).sum())
                        pi[out['child_action'][gidx][:nch]] = plays[:nch]
                    pi /= pi.sum()
                    history[gidx].append((planes[k:k + 1], pi, _side_to_move(game, states[gidx])))
                    acts[gidx] = out['action'][gidx]
                states[live], status = _lib.game_apply(game, states[live], acts[live])
                if (status != 0).any():
                    raise ValueError('Tried to make an illegal move.')
                eng.move_roots(acts)
                w = _lib.game_winner(game, states[live])       # state.Winner(): full scan (Blackbird.py:253)
                for k, gidx in enumerate(live):
                    if w[k] >= 0:
                        winner[gidx] = int(w[k])
                        alive[gidx] = False
            final_planes = _lib.game_encode(game, states[:n])
            for gidx in range(n):                  # terminal example, z, one PutGames per game (Blackbird.py:256-268)
                ex = history[gidx] + [(final_planes[gidx:gidx + 1], np.zeros(gi.A), _side_to_move(game, states[gidx]))]
                wn = winner[gidx]
                blobs = [ExampleState(0 if wn <= 0 else (1 if pl == wn else -1), pi, pln, player=pl).SerializeState()
                         for pln, pi, pl in ex]
                model.Conn.PutGames(model.Name, model.Version, game_cls.GameType, blobs)
    finally:
        eng.close()
    model._games_played = (model._games_played + nGames) % (1 << 31)


def _side_to_move(game, packed):
    """Player of one packed state (include/blackbird_hip.h layouts)."""
    if game == _lib.GAME_DRAGONCHESS:
        return int(np.asarray(packed).view(np.uint8).reshape(-1)[64])
    return int((int(np.asarray(packed).view(np.uint64).reshape(-1)[0]) >> 56) & 3)


def GenerateTrainingSamples(model, nGames, temp):
    """Blackbird.py:219-268.  Raises ValueError if nGames <= 0."""
    if nGames <= 0:
        raise ValueError('Use a positive integer for number of games.')
    game_cls = model.Game
    if model.TimeLimit is not None:
        return _timed_selfplay(model, nGames, temp)
    eng = model._selfplay_engine(nGames)
    # every call is a fresh draw, as in the reference (new numpy choices and new graph noise each time): a new Philox
    # key from numpy's generator (advancing it) and game ids that continue where the model's last run stopped
    eng.set_rng_stream(int(np.random.randint(0, 2 ** 62, dtype=np.int64)), model._games_played)
    model._games_played = (model._games_played + nGames) % (1 << 31)
    eng.reset_counters()
    eng.selfplay_begin(nGames, temp)
    # (s, pi, z) of every example in bulk: AsInputArray planes by one bb_game_encode call per chunk, pi = visits / total in
    # float64 exactly as Node.ChildProbability forms it, wire blobs assembled as one byte matrix -- then one PutGames per
    # game, as the reference issues them (Blackbird.py:267), inside a single sqlite transaction
    gi = _lib.game_info(game_cls.GAME_ID)

    def planes_of(r):
        """AsInputArray planes of a batch of records: one bb_game_encode call.  Made while the GPU is idle (between two launches):
        a kernel launched beside a running persistent launch waits for it -- the launch fills every CU -- and the host work
        queued behind that call would wait with it."""
        st = np.ascontiguousarray(r['state']).view(_lib.STATE_DTYPE[game_cls.GAME_ID]).reshape(len(r), -1)
        return _lib.game_encode(game_cls.GAME_ID, st)

    def blobs_of(r, planes):
        tot = r['total'].astype(np.float64)
        if gi.dense:
            visits = r['visits'][:, :gi.A].astype(np.float64)
        else:  # compact child lists (action id, plays): scatter to the dense A-wide vector the reference stores
            visits = np.zeros((len(r), gi.A), dtype=np.float64)
            live = np.arange(gi.S)[None, :] < r['n_children'][:, None]
            rows = np.nonzero(live)[0]
            visits[rows, r['action'][live]] = r['visits'][live]
        pi = np.where(tot[:, None] > 0, visits / np.maximum(tot, 1.0)[:, None], 0.0)
        return proto_wire.encode_states_batch(r['z'].astype(np.float32), pi, planes[:, None])

    per = max(1, (1 << 22) // gi.A)  # chunks of whole games, about 32 MB of pi at a time (DragonChess: 4032 float64 per example)

    def sink(rec, offs, planes):
        """Blobs + one PutGames per game for a batch of finished games (records compacted in game order): host work only."""
        n, g = len(offs) - 1, 0
        while g < n:
            h = g + 1
            while h < n and offs[h + 1] - offs[g] <= per:
                h += 1
            blobs = blobs_of(rec[offs[g]:offs[h]], planes[offs[g]:offs[h]])
            for k in range(g, h):
                model.Conn.PutGames(model.Name, model.Version, game_cls.GameType,
                                    blobs[offs[k] - offs[g]:offs[k + 1] - offs[g]])
            g = h

    # The host side runs BESIDE the GPU: between two launches (GPU idle) the games that finished during the last one are fetched
    # (compacted on the device, one copy) and their planes encoded; while the next launch plays on they are serialised and
    # stored.  The engine refills a finished game's slot with the next game id by itself, so nGames may exceed the slot count.
    # (Round 2 serialised everything after the last game had ended: the caller saw 60 % of the engine's games/s.)
    deferred = getattr(model.Conn, 'Deferred', None)
    ctx = deferred() if deferred is not None else None
    if ctx is not None:
        ctx.__enter__()
    try:
        seen = np.zeros(nGames, dtype=bool)
        pending = None
        steps = 4  # plies' worth of search per launch: short enough that the last launches' games do not pile up at the end
        while True:
            eng.selfplay_step(steps)            # asynchronous: returns as soon as the launch is queued
            if pending is not None:
                sink(*pending)                  # ... and this runs while it plays
                pending = None
            hdr = eng.selfplay_headers(0, nGames)   # (waits for the launch)
            if eng.counters()['overflow']:  # pool exhausted, a parked slot or an aborted launch: the games would never finish
                raise _lib.BlackbirdHipError('self-play stopped: a search tree outgrew its node pool or a launch was aborted')
            new = np.nonzero((hdr[:, 3] != 0) & ~seen)[0]
            if len(new):
                seen[new] = True
                rec, offs, _win = eng.fetch_games(new, int(hdr[new, 0].sum()))
                pending = (rec, offs, planes_of(rec))
            if seen.all():
                break
        if pending is not None:
            sink(*pending)
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)


def TrainWithExamples(model, batchSize, learningRate, epochs=1, teacher=None, model_override=None,
                      version_override=None):
    """Blackbird.py:271-312"""
    states = model.Conn.GetGames(model_override if model_override is not None else model.Name,
                                 version_override if version_override is not None else model.Version)
    examples = [ExampleState.FromSerialized(state) for state in states]
    order = np.random.choice(len(examples), len(examples) - (len(examples) % batchSize), replace=False)
    examples = [examples[i] for i in order]
    for i in range(len(examples) // batchSize):
        batch = examples[i * batchSize:(i + 1) * batchSize]
        model.train(np.vstack([b.Board for b in batch]), np.hstack([b.MctsEval for b in batch]),
                    np.vstack([b.MctsPolicy for b in batch]), learningRate, teacher)
    model.Version += 1
    model.Conn.PutModel(model.Game.GameType, model.Name, model.Version)


def TestModelsBatched(model1, model2, temp, numTests, **kw):
    """All `numTests` games of TestModels at once on the GPU (blackbird_amd/arena.py); returns an array of +1/0/-1."""
    from .arena import TestModelsBatched as run
    return run(model1, model2, temp, numTests, **kw)


def TestModels(model1, model2, temp, numTests):
    """Blackbird.py:177-216: one head-to-head game, +1 / 0 / -1 for model1's win / draw / loss.  (The reference's
    loop over numTests returns at the end of its first game, so one game is what a call plays.)  The game itself --
    coin toss for the first move, FindMove on the mover's turns, MoveRoot for both after every move -- runs in the
    batched arena with a batch of one."""
    if numTests <= 0:
        return None
    return int(TestModelsBatched(model1, model2, temp, 1)[0])


def _tally(model, opponent, temp, numTests, opName, opVersion=0):
    """The Test* wrappers (Blackbird.py:84-174): numTests games, every result logged, counts returned.  All games are
    played concurrently by the batched arena; results are logged in game order."""
    stats = defaultdict(int)
    if numTests <= 0:
        return stats
    names = {1: 'wins', 0: 'draws', -1: 'losses'}
    for result in TestModelsBatched(model, opponent, temp, numTests):
        stats[names.get(int(result), 'indeterminant')] += 1
        model.Conn.PutTrainingStatistic(int(result), model.Name, model.Version, opName, opVersion)
    return stats


def TestRandom(model, temp, numTests):
    """Blackbird.py:84-111"""
    return _tally(model, RandomMCTS(), temp, numTests, 'RANDOM')


def TestPrevious(model, temp, numTests):
    """Blackbird.py:114-143"""
    oldModel = model.LastVersion()
    return _tally(model, oldModel, temp, numTests, oldModel.Name, oldModel.Version)


def TestGood(model, temp, numTests):
    """Blackbird.py:146-174"""
    good = FixedMCTS(maxDepth=10, explorationRate=0.85, timeLimit=1)
    good.Game = model.Game
    return _tally(model, good, temp, numTests, 'MCTS')


class Model(MCTS, Network):
    """Blackbird.py:315-389: a searcher whose evaluator is its own network; both halves live in the HIP engine."""
    _EVALUATOR = _lib.EVAL_NET

    def __init__(self, game, name, mctsConfig, networkConfig={}, tensorflowConfig={}):
        self.Game, self.Name = game, name
        self.MCTSConfig, self.NetworkConfig, self.TensorflowConfig = mctsConfig, networkConfig, tensorflowConfig
        self.Conn = Connection()
        self.Version = self.Conn.GetLastVersion(game.GameType, name)
        self._saveName = '%s_%s' % (name, self.Version)
        self._batch_engine = None
        self._games_played = 0  # self-play games this model has started: the next run's first global game id
        MCTS.__init__(self, **mctsConfig)
        factory = None
        if networkConfig != {}:
            gi = _lib.game_info(game.GAME_ID)
            factory = NetworkFactory(networkConfig, game.LegalMoves, inputShape=(gi.H, gi.W, gi.C))
        Network.__init__(self, self._saveName, factory, tensorflowConfig)

    def LastVersion(self):
        """Blackbird.py:347-348: a second Model built from the same arguments (it loads the last saved version)."""
        return type(self)(self.Game, self.Name, self.MCTSConfig, self.NetworkConfig, self.TensorflowConfig)

    # ---- engine plumbing -----------------------------------------------------------------------------------
    def _make_engine(self, game_id, n_slots, sims, **kw):
        return _lib.Engine(game_id, n_slots=n_slots, sims_per_move=max(int(sims), 1), mcts_kind=self._KIND,
                           evaluator=_lib.EVAL_NET, c_puct=float(self.ExplorationRate), noise_on=True,
                           alpha=float(self.alpha), epsilon=float(self.epsilon),
                           seed=_seed_from_numpy(), **kw)

    def _after_engine_created(self, engine):
        gi = engine.info
        engine.load_weights(W.flatten(self._ensure_weights(gi.C)))

    def _weights_changed(self):
        Network._weights_changed(self)
        flat = W.flatten(self._weights)
        for eng in (self._engine, self._batch_engine):
            if eng is not None:
                eng.load_weights(flat)
        self.SampleValue.cache_clear()
        self.GetPriors.cache_clear()

    def _selfplay_engine(self, n_games):
        """The batch engine for a GenerateTrainingSamples run: as many concurrent game slots as the run has games, capped
        at MAX_CONCURRENT_GAMES and at what the device's free memory holds (DragonChess pools are sized for 512 plies x
        PlayLimit nodes with 24 edges each: ~0.16 GB per slot at 400 simulations); further games queue on the slots."""
        if self.PlayLimit is None:
            raise ValueError('Not enough information to decide a stop time.')  # (MCTS.py:181-182; a time limit takes _timed_selfplay)
        want = min(n_games, MAX_CONCURRENT_GAMES)
        eng = self._batch_engine
        if eng is not None and (eng.cfg.max_games < n_games or getattr(eng, '_want', 0) < want or eng.cfg.sims_per_move != int(self.PlayLimit)):
            eng.close()
            eng = self._batch_engine = None
        if eng is None:
            n_slots, _per_slot = _lib.fit_slots(self.Game.GAME_ID, want, int(self.PlayLimit), max_games=n_games)
            eng = self._make_engine(self.Game.GAME_ID, n_slots, self.PlayLimit, max_games=n_games)
            eng.load_weights(W.flatten(self._ensure_weights(eng.info.C)))
            eng._want = want
            self._batch_engine = eng
        return eng

    # ---- the Model overrides of the reference (Blackbird.py:350-389), for callers that use them directly; the search
    # itself evaluates leaves inside the engine.  `cache_clear` exists because TrainWithExamples calls it (:288-289);
    # nothing is memoised on the host.
    class _Uncached(object):
        def __init__(self, fn):
            self._fn = fn

        def __get__(self, obj, objtype=None):
            def bound(*args):
                return self._fn(obj, *args)
            bound.cache_clear = lambda: None
            return bound

    def _value_for(self, state, player):
        """SampleValue: the network's tanh value of `state` mapped to [0, 1], from `player`'s side."""
        mine = (self.getEvaluation(state.AsInputArray()) + 1) * 0.5   # float32 arithmetic, as under numpy >= 2
        out = mine if state.Player == player else 1 - mine
        assert out >= 0, 'Value: {}'.format(out)
        return out

    def _priors_for(self, state):
        """GetPriors: the network policy restricted to the legal moves, renormalised."""
        masked = state.LegalActions() * self.getPolicy(state.AsInputArray())
        return masked / np.sum(masked)

    SampleValue = _Uncached(_value_for)
    GetPriors = _Uncached(_priors_for)
