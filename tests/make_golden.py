#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference's own Python modules.

Runs ONLY in the build container (needs /root/reference).  The fixtures are data: inputs and
the reference's outputs for them.  Nothing here, and none of the reference, travels to the GPU
box -- tests read the .npz files only.

Harness-side shims (no reference file is modified):
  * numpy.float = float              (alias removed in numpy >= 1.24; MCTS.py:41,67)
  * PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python so the 2018 state_pb2.py loads
  * sys.modules['tensorflow'] = empty module (only Network/NetworkFactory touch TF; never called)
The network is replaced by a deterministic integer-hash evaluator returning numpy.float32
(the dtype Network.getEvaluation/getPolicy return), so tree-search outputs are exactly
reproducible by the oracle and by the HIP engine.

usage: PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python python tests/make_golden.py [part ...]
"""
import os
import sys
import types

os.environ.setdefault("PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION", "python")
import numpy as np

np.float = float
sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))
sys.path.insert(0, "/root/reference/src")

import Blackbird  # noqa: E402
import Connect4  # noqa: E402
import DragonChess  # noqa: E402
import TicTacToe  # noqa: E402
from DynamicMCTS import DynamicMCTS  # noqa: E402
from FixedMCTS import FixedMCTS  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GAMES = {"c4": Connect4.BoardState, "ttt": TicTacToe.BoardState, "dc": DragonChess.BoardState}
M64 = (1 << 64) - 1


# ---- deterministic evaluator (spec shared with oracle/orc_mcts.c:orc_hash_eval) -----------
def _splitmix(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def _hash_z(x, salt):
    h = 0xCBF29CE484222325 ^ salt
    for b in np.ascontiguousarray(x, dtype=np.int8).view(np.uint8).ravel().tolist():
        h ^= b
        h = (h * 0x100000001B3) & M64
    return _splitmix(h)


def hash_value(x, salt):
    z = _hash_z(x, salt)
    return np.float32(np.float32(z >> 40) * np.float32(1.0 / 8388608.0) - np.float32(1.0))


def hash_policy(x, salt, A):
    z = _hash_z(x, salt)
    out = np.empty(A, dtype=np.float32)
    for a in range(A):
        za = _splitmix((z + (a + 1) * 0x9E3779B97F4A7C15) & M64)
        out[a] = np.float32(1 + (za >> 44))
    return out


class _Conn:
    def __init__(self):
        self.games = []

    def PutGames(self, name, version, gameType, serialized):
        self.games.append((name, version, gameType, list(serialized)))


class RefModel(DynamicMCTS):
    """Duck-typed Blackbird.Model: reference tree search + reference Model overrides, hash net."""
    SampleValue = Blackbird.Model.SampleValue.__wrapped__  # Blackbird.py:350-370 body
    GetPriors = Blackbird.Model.GetPriors.__wrapped__  # Blackbird.py:372-389 body

    def __init__(self, game, salt, **mcts):
        super().__init__(**mcts)
        self.Game, self.Name, self.Version, self.Conn = game, "golden", 0, _Conn()
        self.salt = salt
        self.us = []

    def getEvaluation(self, x):
        return hash_value(x, self.salt)

    def getPolicy(self, x):
        return hash_policy(x, self.salt, self.Game.LegalMoves)

    def FindMove(self, state, temp=0.1, moveTime=None, playLimit=None):
        st = np.random.get_state()
        out = super().FindMove(state, temp, moveTime, playLimit)
        post = np.random.get_state()
        np.random.set_state(st)
        self.us.append(np.random.random_sample())  # the uniform np.random.choice consumed
        np.random.set_state(post)
        return out


class RefFixed(FixedMCTS):
    """FixedMCTS (chain expansion, base-class priors = ones) with the hash value in place of rollouts."""

    def __init__(self, game, salt, **kw):
        super().__init__(**kw)
        self.Game, self.salt = game, salt

    def SampleValue(self, state, player):  # same arithmetic as Model.SampleValue, float32
        value = hash_value(state.AsInputArray(), self.salt)
        value = (value + 1) * 0.5
        if state.Player != player:
            value = 1 - value
        return value


def state_arrays(key, s):
    if key == "dc":
        return (np.array(s.board, dtype=np.int8), s.Player, s.PreviousPlayer or 0,
                [s._white_castle_kingside, s._white_castle_queenside, s._black_castle_kingside,
                 s._black_castle_queenside])
    return np.array(s.Board, dtype=np.int8), s.Player, s.PreviousPlayer or 0, [0, 0, 0, 0]


def wcode(w):
    return -1 if w is None else int(w)


# ---- part 1: random playouts -------------------------------------------------------------
def gen_playouts(key, n_games, seed, max_plies=10 ** 9):
    cls = GAMES[key]
    rng = np.random.RandomState(seed)
    rec = dict(game_start=[0], action=[], player=[], prev=[], win_prev=[], win_none=[], enc=[],
               legal_off=[0], legal_idx=[], board=[], castle=[])
    for _ in range(n_games):
        s = cls()
        ply = 0
        while True:
            la = s.LegalActions()
            idx = np.where(la == 1)[0]
            assert ((la == 0) | (la == 1)).all()
            b, pl, pv, cs = state_arrays(key, s)
            rec["board"].append(b.ravel())
            rec["castle"].append(cs)
            rec["player"].append(pl)
            rec["prev"].append(pv)
            rec["enc"].append(s.AsInputArray().ravel())
            rec["legal_idx"].extend(idx.tolist())
            rec["legal_off"].append(len(rec["legal_idx"]))
            rec["win_none"].append(wcode(s.Winner()))
            if s.Winner() is not None or len(idx) == 0 or ply >= max_plies:
                rec["action"].append(-1)
                rec["win_prev"].append(-1)
                break
            a = int(rng.choice(idx))
            rec["action"].append(a)
            s = s.Copy()
            s.ApplyAction(a)
            rec["win_prev"].append(wcode(s.Winner(a)))
            ply += 1
        rec["game_start"].append(len(rec["action"]))
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["enc"] = out["enc"].astype(np.int8)
    out["board"] = out["board"].astype(np.int8)
    np.savez_compressed(os.path.join(OUT, f"playouts_{key}.npz"), **out)
    print(key, "playouts:", n_games, "games", len(rec["action"]), "positions")


# ---- part 2: arbitrary (not necessarily reachable) boards ----------------------------------
def gen_random_boards(key, n, seed):
    cls = GAMES[key]
    rng = np.random.RandomState(seed)
    rec = dict(board=[], player=[], prev=[], castle=[], legal_off=[0], legal_idx=[], win_none=[],
               win_prev=[], enc=[], apply_ok=[])
    for i in range(n):
        s = cls()
        if key == "dc":
            dens = rng.uniform(0.05, 0.6)
            b = np.zeros((8, 8))
            for r in range(8):
                for c in range(8):
                    if rng.rand() < dens:
                        b[r, c] = rng.choice([1, 2, 2, 2, 3, 4, 5, 6]) * rng.choice([-1, 1])
            s.board = b
            s.Player = int(rng.choice([1, 2]))
            s.PreviousPlayer = [None, 1, 2][rng.randint(3)]
            (s._white_castle_kingside, s._white_castle_queenside, s._black_castle_kingside,
             s._black_castle_queenside) = [bool(x) for x in rng.randint(0, 2, 4)]
            A = 4032
        else:
            H, W = s.Board.shape[:2]
            fill = rng.uniform(0.1, 1.0)
            cells = rng.choice([0, 1, 2], size=(H, W), p=[1 - fill, fill / 2, fill / 2])
            if key == "c4" and i % 2 == 0:  # gravity-consistent half of the time
                for j in range(W):
                    col = [v for v in cells[:, j] if v != 0]
                    cells[:, j] = col + [0] * (H - len(col))
            s.Board = np.zeros((H, W, 2), dtype=np.int8)
            s.Board[:, :, 0] = cells == 1
            s.Board[:, :, 1] = cells == 2
            s.Player = int(rng.choice([1, 2]))
            A = s.LegalMoves
        la = s.LegalActions()
        b, pl, pv, cs = state_arrays(key, s)
        rec["board"].append(b.ravel())
        rec["player"].append(pl)
        rec["prev"].append(pv)
        rec["castle"].append(cs)
        rec["legal_idx"].extend(np.where(la == 1)[0].tolist())
        rec["legal_off"].append(len(rec["legal_idx"]))
        rec["win_none"].append(wcode(s.Winner()))
        rec["enc"].append(s.AsInputArray().ravel())
        if key == "dc":
            rec["win_prev"].append([wcode(s.Winner(0))])
            # ApplyAction on a sample of actions: which raise ValueError?
            acts = rng.randint(0, 4032, 24).tolist() + np.where(la == 1)[0][:8].tolist()
            ok = []
            for a in acts:
                t = s.Copy()
                try:
                    t.ApplyAction(int(a))
                    tb, tp, tv, tc = state_arrays(key, t)
                    ok.append([a, 1, tp, tv] + [int(x) for x in tc] + tb.ravel().tolist())
                except ValueError:
                    ok.append([a, 0, 0, 0, 0, 0, 0, 0] + [0] * 64)
            while len(ok) < 32:
                ok.append([-1, 0, 0, 0, 0, 0, 0, 0] + [0] * 64)
            rec["apply_ok"].append(ok)
        else:
            rec["win_prev"].append([wcode(s.Winner(a)) for a in range(A)])
            ok = []
            for a in range(A):
                t = s.Copy()
                try:
                    t.ApplyAction(a)
                    ok.append(1)
                except ValueError:
                    ok.append(0)
            rec["apply_ok"].append(ok)
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["enc"] = out["enc"].astype(np.int8)
    out["board"] = out["board"].astype(np.int8)
    np.savez_compressed(os.path.join(OUT, f"boards_{key}.npz"), **out)
    print(key, "random boards:", n)


# ---- part 3: tree search under the hash evaluator ----------------------------------------
def gen_mcts(key, tag, n_games, sims, seed, salt, temp=1.0, max_plies=10 ** 9, kind="dynamic",
             max_depth=3, c=0.85, reuse=True):
    cls = GAMES[key]
    np.random.seed(seed)
    rec = dict(game_start=[0], action=[], u=[], plays=[], winrates=[], root_plays=[], v=[], prob=[],
               board=[], player=[], prev=[], castle=[], nodes=[])
    for g in range(n_games):
        if kind == "dynamic":
            m = RefModel(cls, salt + g, explorationRate=c, playLimit=sims)
        else:
            m = RefFixed(cls, salt + g, maxDepth=max_depth, explorationRate=c, playLimit=sims)
        s = cls()
        m.DropRoot()
        ply = 0
        while s.Winner() is None and ply < max_plies:
            st = np.random.get_state()
            nxt, v, prob = m.FindMove(s, temp)
            post = np.random.get_state()
            np.random.set_state(st)
            u = np.random.random_sample()
            np.random.set_state(post)
            b, pl, pv, cs = state_arrays(key, s)
            rec["board"].append(b.ravel())
            rec["player"].append(pl)
            rec["prev"].append(pv)
            rec["castle"].append(cs)
            rec["u"].append(u)
            rec["v"].append(float(v))
            plays = np.array(m.Root.ChildPlays(), dtype=np.float64)
            wr = np.array(m.Root.ChildWinRates(), dtype=np.float64)
            if key == "dc":  # sparse
                nz = np.where(m.Root.LegalActions == 1)[0]
                rec["plays"].append(np.pad(np.stack([nz, plays[nz]], 1), ((0, 140 - len(nz)), (0, 0)), constant_values=-1))
                rec["winrates"].append(np.pad(wr[nz], (0, 140 - len(nz)), constant_values=-1))
                rec["prob"].append(np.pad(np.array(prob)[nz], (0, 140 - len(nz)), constant_values=-1))
            else:
                rec["plays"].append(plays)
                rec["winrates"].append(wr)
                rec["prob"].append(np.array(prob, dtype=np.float64))
            rec["root_plays"].append(m.Root.Plays)
            # which action did the reference pick?
            legal = np.where(s.LegalActions() == 1)[0]
            act = -1
            for a in legal:
                t = s.Copy()
                t.ApplyAction(int(a))
                if t == nxt:
                    act = int(a)
                    break
            assert act >= 0
            rec["action"].append(act)
            s = nxt
            if reuse:
                m.MoveRoot(s)
            else:
                m.DropRoot()
            ply += 1
        rec["game_start"].append(len(rec["action"]))
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["board"] = out["board"].astype(np.int8)
    out["meta"] = np.array([sims, seed, salt, max_depth, int(kind == "fixed"), int(reuse)], dtype=np.int64)
    out["cfg"] = np.array([c, temp], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, f"mcts_{key}_{tag}.npz"), **out)
    print(key, tag, "mcts:", n_games, "games", len(rec["action"]), "moves")


# ---- part 3b: ResetRoot and the Node graph one level down -------------------------------------
def gen_resetroot(key, sims=40, moves=3, sims_after=25, salt=4100):
    """MCTS.ResetRoot (MCTS.py:214-225) after `moves` rounds of FindMove + MoveRoot, all at temp 0 (the PUCT argmax: no random
    draw): the statistics of the top-most ancestor and, through Children, of the nodes on the played line -- they include the
    simulations run from the positions BELOW them, because _backProp (:238-258) recurses through every ancestor -- and then a
    further FindMove from the first position on the tree that ResetRoot left."""
    cls = GAMES[key]
    m = RefModel(cls, salt, explorationRate=0.85, playLimit=sims)
    s = cls()
    acts = []
    for _ in range(moves):
        nxt, _v, _p = m.FindMove(s, 0)
        legal = np.where(s.LegalActions() == 1)[0]
        act = [int(a) for a in legal if m._applyAction(s, int(a)) == nxt][0]
        acts.append(act)
        s = nxt
        m.MoveRoot(s)
    m.ResetRoot()
    out = {"actions": np.array(acts), "meta": np.array([sims, moves, sims_after, salt], dtype=np.int64)}
    node = m.Root
    for depth in range(moves + 1):   # the top-most ancestor, then down the played line
        out[f"plays_{depth}"] = np.float64(node.Plays)
        out[f"value_{depth}"] = np.float64(node.Value)
        out[f"child_plays_{depth}"] = np.array(node.ChildPlays(), dtype=np.float64)
        out[f"child_winrates_{depth}"] = np.array(node.ChildWinRates(), dtype=np.float64)
        out[f"legal_{depth}"] = np.array(node.LegalActions, dtype=np.float64)
        out[f"children_none_{depth}"] = np.array([c is None for c in node.Children]) if node.Children is not None else np.zeros(0, dtype=bool)
        if depth < moves:
            node = node.Children[acts[depth]]
    s0 = cls()
    nxt, v, prob = m.FindMove(s0, 0, playLimit=sims_after)
    out["after_plays"] = np.float64(m.Root.Plays)
    out["after_child_plays"] = np.array(m.Root.ChildPlays(), dtype=np.float64)
    out["after_child_winrates"] = np.array(m.Root.ChildWinRates(), dtype=np.float64)
    out["after_prob"] = np.array(prob, dtype=np.float64)
    out["after_v"] = np.float64(v)
    legal = np.where(s0.LegalActions() == 1)[0]
    out["after_action"] = np.int64([int(a) for a in legal if m._applyAction(s0, int(a)) == nxt][0])
    np.savez_compressed(os.path.join(OUT, f"resetroot_{key}.npz"), **out)
    print(key, "resetroot:", acts, "top plays", out["plays_0"], "after", out["after_plays"])


# ---- part 4: GenerateTrainingSamples end to end -------------------------------------------
def gen_selfplay(key, n_games, sims, seed, salt, temp=1.0):
    cls = GAMES[key]
    np.random.seed(seed)
    m = RefModel(cls, salt, explorationRate=0.85, playLimit=sims)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        Blackbird.GenerateTrainingSamples(m, n_games, temp)  # Blackbird.py:219-268, unmodified
    blobs, off = [], [0]
    for (_n, _v, gtype, ser) in m.Conn.games:
        blobs.extend(ser)
        off.append(len(blobs))
    lens = np.array([len(b) for b in blobs])
    flat = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    # decode with the reference's own protobuf class for the structured view
    from proto.state_pb2 import State
    evals, pis, boards = [], [], []
    for b in blobs:
        st = State()
        st.ParseFromString(b)
        evals.append(st.mctsEval)
        pis.append(np.frombuffer(st.mctsPolicy, dtype=np.float64))
        boards.append(np.frombuffer(st.boardEncoding, dtype=np.int8))
    np.savez_compressed(os.path.join(OUT, f"selfplay_{key}.npz"), blob=flat, blob_len=lens,
                        game_off=np.array(off), u=np.array(m.us), z=np.array(evals, dtype=np.float32),
                        pi=np.array(pis), enc=np.array(boards),
                        meta=np.array([sims, seed, salt, n_games], dtype=np.int64),
                        game_type=np.array([m.Conn.games[0][2]]))
    print(key, "selfplay:", n_games, "games", len(blobs), "examples; blob sizes", sorted(set(lens.tolist())))


PARTS = {
    "resetroot": lambda: (gen_resetroot("c4"), gen_resetroot("ttt", sims=30, moves=2, sims_after=20, salt=4200)),
    "playouts": lambda: (gen_playouts("c4", 200, 11), gen_playouts("ttt", 200, 12),
                         gen_playouts("dc", 16, 13, max_plies=200)),
    "boards": lambda: (gen_random_boards("c4", 600, 21), gen_random_boards("ttt", 400, 22),
                       gen_random_boards("dc", 80, 23)),
    "mcts": lambda: (
        gen_mcts("c4", "s2", 3, 2, 31, 100),
        gen_mcts("c4", "s50", 6, 50, 32, 200),
        gen_mcts("c4", "s800", 2, 800, 33, 300),
        gen_mcts("c4", "s50_noreuse", 2, 50, 34, 400, reuse=False),
        gen_mcts("c4", "s64_t0", 2, 64, 35, 500, temp=0),
        gen_mcts("ttt", "s50", 8, 50, 36, 600),
        gen_mcts("ttt", "s400", 2, 400, 37, 700),
        gen_mcts("c4", "fixed_d3", 3, 20, 38, 800, kind="fixed", max_depth=3),
        gen_mcts("ttt", "fixed_d10", 4, 50, 39, 900, kind="fixed", max_depth=10),
        gen_mcts("dc", "s24", 2, 24, 40, 1000, max_plies=16),
    ),
    "selfplay": lambda: (gen_selfplay("c4", 3, 40, 51, 1100), gen_selfplay("ttt", 4, 30, 52, 1200)),
}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for p in (sys.argv[1:] or list(PARTS)):
        PARTS[p]()
