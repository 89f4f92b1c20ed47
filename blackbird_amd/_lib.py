"""ctypes binding of libblackbird_hip.so (C ABI: include/blackbird_hip.h).

There is no CPU fallback: if the shared library is missing or no MI355X is visible, the first
operation raises.  The library is built in-tree (blackbird_amd/libblackbird_hip.so) by
`__graft_entry__.build()` / `make -C blackbird_amd/csrc`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libblackbird_hip.so")

GAME_CONNECT4, GAME_TICTACTOE, GAME_DRAGONCHESS = 0, 1, 2
MCTS_DYNAMIC, MCTS_FIXED = 0, 1
EVAL_HASH, EVAL_NET, EVAL_ROLLOUT = 0, 1, 2
NET_FORM_AUTO, NET_FORM_F32, NET_FORM_SPLIT = 0, 1, 2   # bb_config.net_form
LAUNCH_AUTO, LAUNCH_LOCKSTEP, LAUNCH_ROUNDS = 0, 1, 2    # bb_config.launch
OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_NAN, ERR_CAPACITY, ERR_WEIGHTS = 0, -1, -2, -3, -4, -5, -6

EXPORTS = (
    "bb_game_info_get", "bb_last_error", "bb_device_count", "bb_game_legal", "bb_game_apply", "bb_game_winner",
    "bb_game_encode", "bb_game_initial", "bb_create", "bb_destroy", "bb_load_weights", "bb_get_counters",
    "bb_reset_counters", "bb_synchronize", "bb_set_sims_per_move", "bb_timing_enable", "bb_timing_read", "bb_timing_net", "bb_selfplay_mode", "bb_net_form", "bb_net_eval", "bb_hash_eval", "bb_set_roots", "bb_run_sims", "bb_run_sims_masked",
    "bb_sample_moves", "bb_move_roots", "bb_get_root_states", "bb_selfplay_begin", "bb_selfplay_step",
    "bb_selfplay_done", "bb_examples_fetch", "bb_examples_device", "bb_selfplay_headers", "bb_examples_fetch_games", "bb_reset_roots", "bb_node_view", "bb_net_eval_keyed", "bb_set_rng_stream", "bb_fit_slots",
)


class GameInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("H", "W", "C", "A", "S", "state_bytes", "dense", "example_bytes")]


_FP = C.POINTER(C.c_float)


class NetWeights(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("H", "W", "C", "F", "R", "D", "A")] + [
        (n, _FP) for n in ("conv0_k", "conv0_b", "conv0_bn", "blk_k", "blk_b", "blk_bn", "v_conv_k", "v_conv_b",
                           "v_bn", "v_d1_k", "v_d1_b", "v_d2_k", "v_d2_b", "p_conv_k", "p_conv_b", "p_bn", "p_d_k",
                           "p_d_b")]


class Config(C.Structure):
    _fields_ = [("game", C.c_int32), ("n_slots", C.c_int32), ("mcts_kind", C.c_int32), ("max_depth", C.c_int32),
                ("evaluator", C.c_int32), ("sims_per_move", C.c_int32), ("max_plies", C.c_int32),
                ("max_games", C.c_int32), ("c_puct", C.c_double), ("seed", C.c_uint64), ("hash_salt", C.c_uint64),
                ("first_game_id", C.c_uint32), ("noise_on", C.c_int32), ("alpha", C.c_float), ("epsilon", C.c_float),
                ("device", C.c_int32), ("salt_per_game", C.c_int32), ("node_capacity", C.c_int32),
                ("net_form", C.c_int32), ("launch", C.c_int32), ("general_net", C.c_int32), ("track_ancestors", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("sims", "sum_depth", "nodes", "terminal_leaves", "games_finished",
                                           "plies", "overflow", "examples", "evals")]


class BlackbirdHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libblackbird_hip.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BlackbirdHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C blackbird_amd/csrc`).  blackbird_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, ip = C.c_void_p, C.c_int
    L.bb_last_error.restype = C.c_char_p
    L.bb_game_info_get.argtypes = [ip, C.POINTER(GameInfo)]
    L.bb_game_legal.argtypes = [ip, ip, vp, vp]
    L.bb_game_apply.argtypes = [ip, ip, vp, vp, vp]
    L.bb_game_winner.argtypes = [ip, ip, vp, vp, vp]
    L.bb_game_encode.argtypes = [ip, ip, vp, vp]
    L.bb_game_initial.argtypes = [ip, vp]
    L.bb_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.bb_destroy.argtypes = [vp]
    L.bb_load_weights.argtypes = [vp, C.POINTER(NetWeights)]
    L.bb_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.bb_reset_counters.argtypes = [vp]
    L.bb_synchronize.argtypes = [vp]
    L.bb_set_sims_per_move.argtypes = [vp, ip]
    L.bb_timing_enable.argtypes = [vp, ip]
    L.bb_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(ip)]
    L.bb_timing_net.argtypes = [vp, ip, ip, ip, C.POINTER(C.c_double)]
    L.bb_selfplay_mode.argtypes = [vp]
    L.bb_net_form.argtypes = [vp]
    L.bb_net_eval.argtypes = [vp, ip, vp, vp, vp, vp, vp, ip]
    L.bb_net_eval_keyed.argtypes = [vp, ip, vp, vp, vp, vp, vp, vp, vp]
    L.bb_set_rng_stream.argtypes = [vp, C.c_uint64, C.c_uint32]
    L.bb_fit_slots.argtypes = [C.POINTER(Config), C.POINTER(ip), C.POINTER(C.c_uint64)]
    L.bb_hash_eval.argtypes = [vp, ip, vp, vp, vp]
    L.bb_set_roots.argtypes = [vp, ip, vp, vp, vp]
    L.bb_run_sims.argtypes = [vp, ip]
    L.bb_run_sims_masked.argtypes = [vp, ip, vp]
    L.bb_sample_moves.argtypes = [vp, C.c_double, vp, vp, vp, vp, vp, vp, vp]
    L.bb_move_roots.argtypes = [vp, vp]
    L.bb_get_root_states.argtypes = [vp, vp]
    L.bb_selfplay_begin.argtypes = [vp, ip, C.c_double]
    L.bb_selfplay_step.argtypes = [vp, ip]
    L.bb_selfplay_done.argtypes = [vp, C.POINTER(ip), C.POINTER(ip)]
    L.bb_examples_fetch.argtypes = [vp, ip, ip, vp, ip, vp, vp]
    L.bb_selfplay_headers.argtypes = [vp, ip, ip, vp]
    L.bb_reset_roots.argtypes = [vp]
    L.bb_node_view.argtypes = [vp, ip, ip, vp, vp, vp, vp, vp]
    L.bb_examples_fetch_games.argtypes = [vp, ip, vp, vp, ip, vp, vp]
    L.bb_examples_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(vp)]
    for name in EXPORTS:
        if name != "bb_last_error":
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def last_error():
    return (lib().bb_last_error() or b"").decode()


def check(rc):
    """Map bb_status codes to the exception types the reference raises."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_STATE:
        raise AssertionError(msg or "Primed for the correct input state.")
    if rc == ERR_NAN:
        raise ValueError(msg or "probabilities contain NaN")
    raise BlackbirdHipError(f"libblackbird_hip error {rc}: {msg}")


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_info_cache = {}


def game_info(game):
    if game not in _info_cache:
        gi = GameInfo()
        check(lib().bb_game_info_get(game, C.byref(gi)))
        _info_cache[game] = gi
    return _info_cache[game]


# ---- packed states (layout: include/blackbird_hip.h) -------------------------------------------------
GRID = {GAME_CONNECT4: (6, 7, 8), GAME_TICTACTOE: (3, 3, 4)}  # H, W, bit stride
STATE_DTYPE = {GAME_CONNECT4: np.dtype("<u8"), GAME_TICTACTOE: np.dtype("<u8"), GAME_DRAGONCHESS: np.dtype("u1")}


def pack_grid(game, boards, players, prevs=None):
    """boards [n,H,W,2] int8 (reference Board layout), players [n], prevs [n] (0 == None) -> uint64 [n,2]."""
    H, W, stride = GRID[game]
    boards = np.asarray(boards).reshape(-1, H, W, 2)
    n = boards.shape[0]
    out = np.zeros((n, 2), dtype=np.uint64)
    for r in range(H):
        for c in range(W):
            bit = np.uint64(r * stride + c)
            out[:, 0] |= (boards[:, r, c, 0] != 0).astype(np.uint64) << bit
            out[:, 1] |= (boards[:, r, c, 1] != 0).astype(np.uint64) << bit
    pl = np.asarray(players, dtype=np.uint64).reshape(n)
    pv = np.zeros(n, dtype=np.uint64) if prevs is None else np.asarray(prevs, dtype=np.uint64).reshape(n)
    out[:, 0] |= (pl & np.uint64(3)) << np.uint64(56)
    out[:, 0] |= (pv & np.uint64(3)) << np.uint64(58)
    return out


def unpack_grid(game, packed):
    """uint64 [n,2] -> (boards [n,H,W,2] int8, players [n], prevs [n])."""
    H, W, stride = GRID[game]
    packed = np.asarray(packed, dtype=np.uint64).reshape(-1, 2)
    n = packed.shape[0]
    boards = np.zeros((n, H, W, 2), dtype=np.int8)
    for r in range(H):
        for c in range(W):
            bit = np.uint64(r * stride + c)
            boards[:, r, c, 0] = (packed[:, 0] >> bit) & np.uint64(1)
            boards[:, r, c, 1] = (packed[:, 1] >> bit) & np.uint64(1)
    players = ((packed[:, 0] >> np.uint64(56)) & np.uint64(3)).astype(np.int8)
    prevs = ((packed[:, 0] >> np.uint64(58)) & np.uint64(3)).astype(np.int8)
    return boards, players, prevs


def pack_dc(boards, players, prevs=None, castles=None):
    """boards [n,8,8] piece codes, players [n], prevs [n] (0 == None), castles [n,4] -> uint8 [n,80]."""
    boards = np.asarray(boards).reshape(-1, 64)
    n = boards.shape[0]
    out = np.zeros((n, 80), dtype=np.uint8)
    out[:, :64] = boards.astype(np.int8).view(np.uint8)
    out[:, 64] = np.asarray(players, dtype=np.uint8).reshape(n)
    out[:, 65] = 0 if prevs is None else np.asarray(prevs, dtype=np.uint8).reshape(n)
    if castles is not None:
        out[:, 66:70] = (np.asarray(castles).reshape(n, 4) != 0).astype(np.uint8)
    return out


def unpack_dc(packed):
    packed = np.asarray(packed, dtype=np.uint8).reshape(-1, 80)
    boards = packed[:, :64].view(np.int8).reshape(-1, 8, 8).copy()
    return boards, packed[:, 64].astype(np.int8), packed[:, 65].astype(np.int8), packed[:, 66:70].astype(np.int8)


# ---- stateless batched game ops --------------------------------------------------------------------
def game_legal(game, states):
    gi = game_info(game)
    states = np.ascontiguousarray(states)
    n = states.shape[0]
    out = np.zeros((n, gi.A), dtype=np.uint8)
    check(lib().bb_game_legal(game, n, ptr(states), ptr(out)))
    return out


def game_apply(game, states, actions):
    states = np.ascontiguousarray(states).copy()
    n = states.shape[0]
    actions = np.ascontiguousarray(actions, dtype=np.int32)
    status = np.zeros(n, dtype=np.int32)
    check(lib().bb_game_apply(game, n, ptr(states), ptr(actions), ptr(status)))
    return states, status


def game_winner(game, states, prev_actions=None):
    states = np.ascontiguousarray(states)
    n = states.shape[0]
    prev = None if prev_actions is None else np.ascontiguousarray(prev_actions, dtype=np.int32)
    out = np.zeros(n, dtype=np.int8)
    check(lib().bb_game_winner(game, n, ptr(states), ptr(prev), ptr(out)))
    return out


def game_encode(game, states):
    gi = game_info(game)
    states = np.ascontiguousarray(states)
    n = states.shape[0]
    out = np.zeros((n, gi.H, gi.W, gi.C), dtype=np.int8)
    check(lib().bb_game_encode(game, n, ptr(states), ptr(out)))
    return out


def game_initial(game):
    gi = game_info(game)
    buf = np.zeros(gi.state_bytes, dtype=np.uint8)
    check(lib().bb_game_initial(game, ptr(buf)))
    return buf.view(STATE_DTYPE[game]).reshape(1, -1)


def example_dtype(game):
    gi = game_info(game)
    fields = [("game_id", "<u4"), ("ply", "<u2"), ("player", "u1"), ("z", "i1"), ("total", "<u4"),
              ("n_children", "<u4"), ("state", "u1", (gi.state_bytes,)), ("visits", "<u4", (gi.S,))]
    if not gi.dense:
        fields.append(("action", "<u2", (gi.S,)))
    dt = np.dtype(fields)
    assert dt.itemsize == gi.example_bytes, (dt.itemsize, gi.example_bytes)
    return dt


def fit_slots(game, n_slots, sims_per_move, *, mcts_kind=MCTS_DYNAMIC, max_depth=10, max_plies=None, max_games=None,
              node_capacity=0, device=0):
    """bb_fit_slots: (largest slot count <= n_slots whose pools fit the device's free memory, bytes per slot)."""
    if max_plies is None:
        max_plies = {GAME_CONNECT4: 42, GAME_TICTACTOE: 9}.get(game, 512)
    cfg = Config(game=game, n_slots=n_slots, mcts_kind=mcts_kind, max_depth=max_depth, evaluator=EVAL_NET,
                 sims_per_move=sims_per_move, max_plies=max_plies, max_games=max_games or n_slots, c_puct=1.0,
                 device=device, node_capacity=node_capacity)
    fit, per = C.c_int(), C.c_uint64()
    check(lib().bb_fit_slots(C.byref(cfg), C.byref(fit), C.byref(per)))
    return fit.value, per.value


# ---- engine ----------------------------------------------------------------------------------------------
class Engine:
    """One GPU-resident batch of search trees (bb_engine)."""

    def __init__(self, game, n_slots, sims_per_move, *, mcts_kind=MCTS_DYNAMIC, max_depth=10, evaluator=EVAL_NET,
                 c_puct=0.85, max_plies=None, max_games=None, seed=1234, hash_salt=0, first_game_id=0,
                 noise_on=False, alpha=0.2, epsilon=0.3, device=0, salt_per_game=False, node_capacity=0,
                 net_form=0, launch=0, general_net=False, track_ancestors=False):
        """net_form: NET_FORM_AUTO / NET_FORM_F32 / NET_FORM_SPLIT (bb_config.net_form); launch: LAUNCH_AUTO / LAUNCH_LOCKSTEP /
        LAUNCH_ROUNDS (bb_config.launch); general_net: a 16-filter network through the launch-per-layer kernels."""
        self.game = game
        self.info = game_info(game)
        if max_plies is None:
            max_plies = {GAME_CONNECT4: 42, GAME_TICTACTOE: 9}.get(game, 512)
        cfg = Config(game=game, n_slots=n_slots, mcts_kind=mcts_kind, max_depth=max_depth, evaluator=evaluator,
                     sims_per_move=sims_per_move, max_plies=max_plies, max_games=max_games or n_slots, c_puct=c_puct,
                     seed=seed, hash_salt=hash_salt, first_game_id=first_game_id, noise_on=int(noise_on), alpha=alpha,
                     epsilon=epsilon, device=device, salt_per_game=int(salt_per_game), node_capacity=node_capacity,
                     net_form=int(net_form), launch=int(launch), general_net=int(bool(general_net)),
                     track_ancestors=int(bool(track_ancestors)))
        self.cfg = cfg
        self.h = C.c_void_p()
        self.n_slots = n_slots
        self.max_plies = max_plies
        check(lib().bb_create(C.byref(cfg), C.byref(self.h)))
        self._weights_keep = None

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            lib().bb_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def load_weights(self, flat):
        """flat: dict from blackbird_amd.weights.flatten()."""
        gi = self.info
        w = NetWeights()
        k0 = flat["conv0_k"]
        w.H, w.W, w.C, w.F = gi.H, gi.W, k0.shape[2], k0.shape[3]
        w.R, w.D, w.A = flat["blk_k"].shape[0], flat["v_d1_k"].shape[0], flat["p_d_b"].shape[0]
        keep = {}
        for name, _t in NetWeights._fields_[7:]:
            keep[name] = np.ascontiguousarray(flat[name], dtype=np.float32)
            setattr(w, name, keep[name].ctypes.data_as(_FP))
        self._weights_keep = keep
        check(lib().bb_load_weights(self.h, C.byref(w)))

    def counters(self):
        c = Counters()
        check(lib().bb_get_counters(self.h, C.byref(c)))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    def reset_counters(self):
        check(lib().bb_reset_counters(self.h))

    def synchronize(self):
        check(lib().bb_synchronize(self.h))

    def set_sims_per_move(self, sims):
        check(lib().bb_set_sims_per_move(self.h, int(sims)))

    def timing_enable(self, every_n):
        check(lib().bb_timing_enable(self.h, int(every_n)))

    def timing_read(self):
        m, mn, c = C.c_double(), C.c_double(), C.c_int()
        check(lib().bb_timing_read(self.h, C.byref(m), C.byref(mn), C.byref(c)))
        return m.value, mn.value, c.value

    def selfplay_mode(self):
        return check(lib().bb_selfplay_mode(self.h))

    def net_form(self):
        """0 float32-MFMA fused tower, 1 general-filter launches, 2 fused tower on the bf16 matrix pipe (float32 by operand
        splitting), 3 general-filter launches with such tower layers (bb_net_form)."""
        return check(lib().bb_net_form(self.h))

    def timing_net(self, iters=50, noise=True, ablate=0):
        ms = C.c_double()
        check(lib().bb_timing_net(self.h, int(iters), int(noise), int(ablate), C.byref(ms)))
        return ms.value

    def net_eval(self, states=None, planes=None, noise=False):
        A = self.info.A
        if states is not None:
            states = np.ascontiguousarray(states)
            n = states.shape[0]
        else:
            planes = np.ascontiguousarray(planes, dtype=np.int8)
            n = planes.shape[0]
        value = np.zeros(n, dtype=np.float32)
        logits = np.zeros((n, A), dtype=np.float32)
        policy = np.zeros((n, A), dtype=np.float32)
        check(lib().bb_net_eval(self.h, n, ptr(states), ptr(planes), ptr(value), ptr(logits), ptr(policy), int(noise)))
        return value, logits, policy

    def net_eval_keyed(self, game_ids, node_serials, states=None, planes=None):
        """bb_net_eval_keyed: the forward pass with the prior noise of (global game id, node serial) per position --
        exactly the priors a self-play kernel uses for that node."""
        A = self.info.A
        if states is not None:
            states = np.ascontiguousarray(states)
            n = states.shape[0]
        else:
            planes = np.ascontiguousarray(planes, dtype=np.int8)
            n = planes.shape[0]
        gids = np.ascontiguousarray(game_ids, dtype=np.uint32).reshape(n)
        sers = np.ascontiguousarray(node_serials, dtype=np.int32).reshape(n)
        value = np.zeros(n, dtype=np.float32)
        logits = np.zeros((n, A), dtype=np.float32)
        policy = np.zeros((n, A), dtype=np.float32)
        check(lib().bb_net_eval_keyed(self.h, n, ptr(states), ptr(planes), ptr(gids), ptr(sers), ptr(value), ptr(logits),
                                      ptr(policy)))
        return value, logits, policy

    def set_rng_stream(self, seed, first_game_id=0):
        check(lib().bb_set_rng_stream(self.h, C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint32(int(first_game_id) & 0xFFFFFFFF)))
        self.cfg.seed = int(seed) & (2 ** 64 - 1)
        self.cfg.first_game_id = int(first_game_id) & 0xFFFFFFFF

    def hash_eval(self, states):
        states = np.ascontiguousarray(states)
        n = states.shape[0]
        value = np.zeros(n, dtype=np.float32)
        policy = np.zeros((n, self.info.A), dtype=np.float32)
        check(lib().bb_hash_eval(self.h, n, ptr(states), ptr(value), ptr(policy)))
        return value, policy

    def set_roots(self, states, slots=None, game_ids=None):
        states = np.ascontiguousarray(states)
        n = states.shape[0]
        slots = None if slots is None else np.ascontiguousarray(slots, dtype=np.int32)
        gids = None if game_ids is None else np.ascontiguousarray(game_ids, dtype=np.uint32)
        check(lib().bb_set_roots(self.h, n, ptr(slots), ptr(states), ptr(gids)))

    def run_sims(self, sims, mask=None):
        if mask is None:
            check(lib().bb_run_sims(self.h, int(sims)))
        else:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            assert mask.shape[0] == self.n_slots
            check(lib().bb_run_sims_masked(self.h, int(sims), ptr(mask)))

    def sample_moves(self, temp, u=None):
        n, S = self.n_slots, self.info.S
        u = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        out = dict(action=np.zeros(n, np.int32), root_winrate=np.zeros(n, np.float32),
                   root_plays=np.zeros(n, np.int32), child_action=np.zeros((n, S), np.int32),
                   child_plays=np.zeros((n, S), np.int32), child_value=np.zeros((n, S), np.float32))
        check(lib().bb_sample_moves(self.h, float(temp), ptr(u), ptr(out["action"]), ptr(out["root_winrate"]),
                                    ptr(out["root_plays"]), ptr(out["child_action"]), ptr(out["child_plays"]),
                                    ptr(out["child_value"])))
        return out

    def move_roots(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        assert actions.shape[0] == self.n_slots
        check(lib().bb_move_roots(self.h, ptr(actions)))

    def reset_roots(self):
        """bb_reset_roots: MCTS.ResetRoot for every slot."""
        check(lib().bb_reset_roots(self.h))

    def node_view(self, slot=0, node=-1):
        """bb_node_view: dict(child, plays, value [S], state (packed), flags, legal_mask, node)."""
        S = self.info.S
        child, plays, value = np.zeros(S, np.int32), np.zeros(S, np.int32), np.zeros(S, np.float32)
        state = np.zeros((1, self.info.state_bytes), dtype=np.uint8)
        info = np.zeros(3, np.int32)
        check(lib().bb_node_view(self.h, int(slot), int(node), ptr(child), ptr(plays), ptr(value), ptr(state), ptr(info)))
        return dict(child=child, plays=plays, value=value, state=state.view(STATE_DTYPE[self.game]), flags=int(info[0]),
                    legal_mask=int(np.uint32(info[1])), node=int(info[2]))

    def root_states(self):
        buf = np.zeros((self.n_slots, self.info.state_bytes), dtype=np.uint8)
        check(lib().bb_get_root_states(self.h, ptr(buf)))
        return buf.view(STATE_DTYPE[self.game])

    def selfplay_begin(self, n_games, temp):
        check(lib().bb_selfplay_begin(self.h, int(n_games), float(temp)))
        self._n_games = int(n_games)

    def selfplay_step(self, plies=1):
        check(lib().bb_selfplay_step(self.h, int(plies)))

    def selfplay_done(self):
        d, f = C.c_int(), C.c_int()
        check(lib().bb_selfplay_done(self.h, C.byref(d), C.byref(f)))
        return bool(d.value), f.value

    def fetch_examples(self, first_game=0, n_games=None):
        n_games = self._n_games if n_games is None else n_games
        dt = example_dtype(self.game)
        cap = n_games * (self.max_plies + 1)
        rec = np.zeros(cap, dtype=dt)
        offs = np.zeros(n_games + 1, dtype=np.int32)
        win = np.zeros(n_games, dtype=np.int8)
        k = check(lib().bb_examples_fetch(self.h, first_game, n_games, ptr(rec), cap, ptr(offs), ptr(win)))
        return rec[:k], offs, win

    def selfplay_headers(self, first_game=0, n_games=None):
        """bb_selfplay_headers: int32 [n_games][4] = (n_examples, winner, plies, done)."""
        n_games = self._n_games if n_games is None else n_games
        hdr = np.zeros((n_games, 4), dtype=np.int32)
        check(lib().bb_selfplay_headers(self.h, int(first_game), int(n_games), ptr(hdr)))
        return hdr

    def fetch_games(self, game_ids, n_records=None):
        """bb_examples_fetch_games: the records of the listed (finished) games, compacted in list order."""
        ids = np.ascontiguousarray(game_ids, dtype=np.int32)
        cap = int(n_records) if n_records is not None else len(ids) * (self.max_plies + 1)
        rec = np.zeros(max(cap, 1), dtype=example_dtype(self.game))
        offs = np.zeros(len(ids) + 1, dtype=np.int32)
        win = np.zeros(len(ids), dtype=np.int8)
        k = check(lib().bb_examples_fetch_games(self.h, len(ids), ptr(ids), ptr(rec), cap, ptr(offs), ptr(win)))
        return rec[:k], offs, win

    def examples_device(self):
        """bb_examples_device: (records device pointer, bytes, record bytes, game header device pointer)."""
        p, hdr = C.c_void_p(), C.c_void_p()
        nb, rb = C.c_uint64(), C.c_uint64()
        check(lib().bb_examples_device(self.h, C.byref(p), C.byref(nb), C.byref(rb), C.byref(hdr)))
        return p.value, nb.value, rb.value, hdr.value
