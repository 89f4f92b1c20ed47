"""ctypes front for the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under blackbird_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liborc.so")

C4, TTT, DC = 0, 1, 2
DYNAMIC, FIXED = 0, 1
EVAL_HASH, EVAL_NET, EVAL_ROLLOUT, EVAL_CALLBACK, EVAL_CALLBACK_KEYED = 0, 1, 2, 3, 4
GAME_IDS = {"Connect4": C4, "TicTacToe": TTT, "DragonChess": DC}


class State(C.Structure):
    _fields_ = [("b", C.c_int8 * 128), ("player", C.c_int8), ("prev", C.c_int8),
                ("castle", C.c_int8 * 4), ("pad", C.c_int8 * 2)]

    def copy(self):
        s = State()
        C.memmove(C.byref(s), C.byref(self), C.sizeof(State))
        return s


class Dims(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("A", C.c_int)]


_FP = C.POINTER(C.c_float)


class Net(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("H", "W", "C", "F", "R", "D", "A")] + [
        (n, _FP) for n in ("conv0_k", "conv0_b", "conv0_bn", "blk_k", "blk_b", "blk_bn",
                           "v_conv_k", "v_conv_b", "v_bn", "v_d1_k", "v_d1_b", "v_d2_k", "v_d2_b",
                           "p_conv_k", "p_conv_b", "p_bn", "p_d_k", "p_d_b")]


EVAL_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(State), _FP, _FP)
EVAL_CB2 = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(State), C.c_uint32, C.c_uint32, _FP, _FP)  # + (game id, node serial)


class Cfg(C.Structure):
    _fields_ = [("game", C.c_int), ("kind", C.c_int), ("max_depth", C.c_int), ("evaluator", C.c_int),
                ("c_puct", C.c_double), ("salt", C.c_uint64), ("seed", C.c_uint64),
                ("net", C.POINTER(Net)), ("noise_on", C.c_int), ("alpha", C.c_float), ("eps", C.c_float),
                ("cb", EVAL_CB), ("cb_ctx", C.c_void_p), ("priors_ones", C.c_int), ("cb2", EVAL_CB2)]


class Stats(C.Structure):
    _fields_ = [("sims", C.c_uint64), ("evals", C.c_uint64), ("sum_depth", C.c_uint64),
                ("nodes", C.c_uint64), ("terminal_leaves", C.c_uint64), ("max_depth_seen", C.c_int)]


def build(force=False):
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB)
            for f in ("orc.h", "orc_games.c", "orc_mcts.c", "orc_net.c", "Makefile")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"] + (["-B"] if force else []))
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        build()
    L = C.CDLL(_LIB)
    SP, DP = C.POINTER(State), C.POINTER(C.c_double)
    L.orc_game_dims.argtypes = [C.c_int, C.POINTER(Dims)]
    L.orc_game_init.argtypes = [C.c_int, SP]
    L.orc_game_legal.argtypes = [C.c_int, SP, DP]
    L.orc_game_apply.argtypes = [C.c_int, SP, C.c_int]
    L.orc_game_apply.restype = C.c_int
    L.orc_game_winner.argtypes = [C.c_int, SP, C.c_int]
    L.orc_game_winner.restype = C.c_int
    L.orc_game_encode.argtypes = [C.c_int, SP, C.c_void_p]
    L.orc_game_equal.argtypes = [C.c_int, SP, SP]
    L.orc_game_equal.restype = C.c_int
    L.orc_hash_eval.argtypes = [C.c_int, C.c_uint64, SP, _FP, C.c_void_p]
    L.orc_net_forward.argtypes = [C.POINTER(Net), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_net_forward_perpixel.argtypes = [C.POINTER(Net), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_np_sum.argtypes = [C.c_void_p, C.c_int]
    L.orc_np_sum.restype = C.c_double
    L.orc_philox.argtypes = [C.c_uint64] + [C.c_uint32] * 4 + [C.POINTER(C.c_uint32 * 4)]
    L.orc_u53.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    L.orc_u53.restype = C.c_double
    L.orc_beta_noise.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float]
    L.orc_beta_noise.restype = C.c_float
    L.orc_search_new.argtypes = [C.POINTER(Cfg), C.c_uint32]
    L.orc_search_new.restype = C.c_void_p
    L.orc_search_free.argtypes = [C.c_void_p]
    L.orc_drop_root.argtypes = [C.c_void_p]
    L.orc_has_root.argtypes = [C.c_void_p]
    L.orc_has_root.restype = C.c_int
    L.orc_move_root.argtypes = [C.c_void_p, SP]
    L.orc_move_root.restype = C.c_int
    L.orc_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.orc_find_move.argtypes = [C.c_void_p, SP, C.c_double, C.c_int, C.c_double, C.c_uint32,
                                C.POINTER(C.c_int), SP, DP, C.c_void_p, C.c_void_p, C.c_void_p, DP]
    L.orc_find_move.restype = C.c_int
    L.orc_select_puct.argtypes = [C.c_void_p]
    L.orc_select_puct.restype = C.c_int
    L.orc_sample_action.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
    L.orc_sample_action.restype = C.c_int
    L.orc_selfplay_game.argtypes = [C.POINTER(Cfg), C.c_uint32, C.c_double, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_int), C.POINTER(Stats)]
    L.orc_selfplay_game.restype = C.c_int
    _lib = L
    return L


def dims(game):
    d = Dims()
    lib().orc_game_dims(game, C.byref(d))
    return d.H, d.W, d.C, d.A


# ---- game helpers -----------------------------------------------------------------------
def new_state(game):
    s = State()
    lib().orc_game_init(game, C.byref(s))
    return s


def state_from_arrays(game, board, player, prev=None, castle=None):
    """board: reference-shaped array (C4 [6,7,2] int8, TTT [3,3,2], DC [8,8] piece codes)."""
    s = State()
    flat = np.ascontiguousarray(np.asarray(board).astype(np.int8)).ravel()
    for i, v in enumerate(flat):
        s.b[i] = int(v)
    s.player = int(player)
    s.prev = 0 if prev is None else int(prev)
    if castle is not None:
        for i in range(4):
            s.castle[i] = int(bool(castle[i]))
    return s


def legal(game, st):
    A = dims(game)[3]
    out = np.zeros(A, dtype=np.float64)
    lib().orc_game_legal(game, C.byref(st), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def apply(game, st, action):
    return lib().orc_game_apply(game, C.byref(st), int(action))


def winner(game, st, prev=None):
    w = lib().orc_game_winner(game, C.byref(st), -1 if prev is None else int(prev))
    return None if w < 0 else w


def encode(game, st):
    H, W, Cc, _ = dims(game)
    out = np.zeros((1, H, W, Cc), dtype=np.int8)
    lib().orc_game_encode(game, C.byref(st), out.ctypes.data)
    return out


def hash_eval(game, salt, st):
    A = dims(game)[3]
    v = C.c_float()
    p = np.zeros(A, dtype=np.float32)
    lib().orc_hash_eval(game, salt, C.byref(st), C.byref(v), p.ctypes.data)
    return np.float32(v.value), p


# ---- network ----------------------------------------------------------------------------
class NetWeights:
    """Holds float32 arrays (TF variable layout) and the C view of them."""

    FIELDS = ("conv0_k", "conv0_b", "conv0_bn", "blk_k", "blk_b", "blk_bn", "v_conv_k", "v_conv_b",
              "v_bn", "v_d1_k", "v_d1_b", "v_d2_k", "v_d2_b", "p_conv_k", "p_conv_b", "p_bn", "p_d_k",
              "p_d_b")

    def __init__(self, H, W, Cc, F, R, D, A, arrays):
        self.shape = (H, W, Cc, F, R, D, A)
        self.arrays = {k: np.ascontiguousarray(arrays[k], dtype=np.float32) for k in self.FIELDS}
        self.c = Net()
        for n, v in zip(("H", "W", "C", "F", "R", "D", "A"), self.shape):
            setattr(self.c, n, v)
        for k in self.FIELDS:
            setattr(self.c, k, self.arrays[k].ctypes.data_as(_FP))


def net_forward(w, boards, perpixel=False):
    """Network forward.  Default: the pooled statement of the heads (what the HIP kernels compute); perpixel=True: the
    reference's literal op order (dense per pixel, then reduce_sum) -- see orc_net.c."""
    boards = np.ascontiguousarray(boards, dtype=np.int8)
    n = boards.shape[0]
    A = w.shape[6]
    value = np.zeros(n, dtype=np.float32)
    logits = np.zeros((n, A), dtype=np.float32)
    policy = np.zeros((n, A), dtype=np.float32)
    fn = lib().orc_net_forward_perpixel if perpixel else lib().orc_net_forward
    fn(C.byref(w.c), boards.ctypes.data, n, value.ctypes.data, logits.ctypes.data, policy.ctypes.data)
    return value, logits, policy


# ---- search -----------------------------------------------------------------------------
def make_cfg(game, kind=DYNAMIC, evaluator=EVAL_HASH, c_puct=0.85, max_depth=10, salt=0, seed=1234,
             net=None, noise_on=False, alpha=0.2, eps=0.3, cb=None, priors_ones=False, cb2=None):
    c = Cfg()
    c.game, c.kind, c.max_depth, c.evaluator = game, kind, max_depth, evaluator
    c.c_puct, c.salt, c.seed = c_puct, salt, seed
    c.noise_on, c.alpha, c.eps = int(noise_on), alpha, eps
    c.priors_ones = int(priors_ones)
    if net is not None:
        c.net = C.pointer(net.c)
        c._net_keep = net
    if cb is not None:
        c.cb = cb
        c._cb_keep = cb
    if cb2 is not None:
        c.cb2 = cb2
        c._cb2_keep = cb2
    return c


class Search:
    def __init__(self, cfg, game_id=0):
        self.cfg = cfg
        self.game = cfg.game
        self.A = dims(cfg.game)[3]
        self.h = lib().orc_search_new(C.byref(cfg), game_id)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_search_free(self.h)
            self.h = None

    def drop_root(self):
        lib().orc_drop_root(self.h)

    def has_root(self):
        return bool(lib().orc_has_root(self.h))

    def move_root(self, st):
        return lib().orc_move_root(self.h, C.byref(st))

    def stats(self):
        s = Stats()
        lib().orc_get_stats(self.h, C.byref(s))
        return s

    def find_move(self, st, temp, play_limit, u=-1.0, ply=0):
        act = C.c_int()
        nxt = State()
        wr = C.c_double()
        rp = C.c_double()
        prob = np.zeros(self.A)
        plays = np.zeros(self.A)
        wrs = np.zeros(self.A)
        rc = lib().orc_find_move(self.h, C.byref(st), temp, play_limit, u, ply, C.byref(act),
                                 C.byref(nxt), C.byref(wr), prob.ctypes.data, plays.ctypes.data,
                                 wrs.ctypes.data, C.byref(rp))
        if rc == -2:
            raise AssertionError("Primed for the correct input state.")
        if rc == -3:
            raise ValueError("probabilities contain NaN")
        if rc < 0:
            raise ValueError("Not enough information to decide a stop time.")
        return dict(action=act.value, next=nxt, winrate=wr.value, prob=prob, plays=plays,
                    winrates=wrs, root_plays=rp.value)


def sample_action(plays, temp, u):
    plays = np.ascontiguousarray(plays, dtype=np.float64)
    return lib().orc_sample_action(plays.ctypes.data, len(plays), temp, u)


def selfplay_game(cfg, game_id, temp, play_limit, max_plies):
    H, W, Cc, A = dims(cfg.game)
    n = max_plies + 1
    boards = np.zeros((n, H, W, Cc), dtype=np.int8)
    pi = np.zeros((n, A), dtype=np.float64)
    player = np.zeros(n, dtype=np.int8)
    z = np.zeros(n, dtype=np.float32)
    actions = np.zeros(n, dtype=np.int32)
    win = C.c_int()
    st = Stats()
    k = lib().orc_selfplay_game(C.byref(cfg), game_id, temp, play_limit, max_plies, boards.ctypes.data,
                                pi.ctypes.data, player.ctypes.data, z.ctypes.data, actions.ctypes.data,
                                C.byref(win), C.byref(st))
    if k < 0:
        raise RuntimeError(f"oracle selfplay failed rc={k}")
    return dict(n=k, boards=boards[:k], pi=pi[:k], player=player[:k], z=z[:k], actions=actions[:k - 1],
                winner=win.value, stats=st)
