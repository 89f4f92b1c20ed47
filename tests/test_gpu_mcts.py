"""-m gpu: HIP tree search (through the C ABI) vs reference golden vectors and vs the oracle.
Evaluator = the deterministic integer hash (same spec in reference harness, oracle and HIP), so
visit counts, Values, priors and sampled moves must match EXACTLY."""
import glob
import os

import numpy as np
import pytest

from blackbird_amd import _lib

pytestmark = pytest.mark.gpu
GAMES = {"c4": (_lib.GAME_CONNECT4, 0), "ttt": (_lib.GAME_TICTACTOE, 1), "dc": (_lib.GAME_DRAGONCHESS, 2)}


def pack_states(game, g):
    if game == _lib.GAME_DRAGONCHESS:
        return _lib.pack_dc(g["board"].reshape(-1, 8, 8), g["player"], g["prev"], g["castle"])
    H, W, _s = _lib.GRID[game]
    return _lib.pack_grid(game, g["board"].reshape(-1, H, W, 2), g["player"], g["prev"])


def same_position(game, a, b):
    """GameState.__eq__: PreviousPlayer is ignored."""
    if game == _lib.GAME_DRAGONCHESS:
        a, b = a.copy(), b.copy()
        a[:, 65] = 0
        b[:, 65] = 0
        return np.array_equal(a[:, :70], b[:, :70])
    m = np.uint64((1 << 58) - 1)
    return np.array_equal(a & m, b & m)
FILES = sorted(os.path.basename(p) for p in glob.glob(
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mcts_*.npz"))
    if os.path.basename(p).split("_")[1] in GAMES)


@pytest.mark.parametrize("fname", FILES)
def test_find_move_golden(golden_dir, fname):
    g = np.load(os.path.join(golden_dir, fname), allow_pickle=False)
    key = fname.split("_")[1]
    game, _ = GAMES[key]
    A = _lib.game_info(game).A
    dc = game == _lib.GAME_DRAGONCHESS
    cells = 64 if dc else _lib.GRID[game][0] * _lib.GRID[game][1]
    sims, seed, salt, max_depth, fixed, reuse = [int(x) for x in g["meta"]]
    c, temp = [float(x) for x in g["cfg"]]
    gs = g["game_start"]
    ng = len(gs) - 1
    lens = np.diff(gs)
    st_all = pack_states(game, g)
    eng = _lib.Engine(game, n_slots=ng, sims_per_move=sims, mcts_kind=_lib.MCTS_FIXED if fixed else _lib.MCTS_DYNAMIC,
                      max_depth=max_depth, evaluator=_lib.EVAL_HASH, c_puct=c, hash_salt=salt, salt_per_game=True,
                      node_capacity=sims * (cells + 1) * (max_depth if fixed else 1) + 8)
    eng.set_roots(st_all[gs[:-1]], game_ids=np.arange(ng))
    for ply in range(int(lens.max())):
        live = lens > ply
        idx = gs[:-1] + np.minimum(ply, lens - 1)
        # FindMove asserts Root.State == state: the engine's root must be the fixture's position
        roots = eng.root_states()
        assert same_position(game, roots[live], st_all[idx][live])
        eng.run_sims(sims)
        out = eng.sample_moves(temp, u=g["u"][idx])
        for s in np.where(live)[0]:
            i = idx[s]
            if dc:  # compact child lists: (action, plays) pairs in ascending action order
                k = int((g["plays"][i][:, 0] >= 0).sum())
                assert np.array_equal(out["child_action"][s, :k], g["plays"][i][:k, 0].astype(np.int32)), (fname, s, ply)
                assert (out["child_action"][s, k:] == -1).all()
                plays = out["child_plays"][s, :k].astype(np.float64)
                assert np.array_equal(plays, g["plays"][i][:k, 1]), (fname, s, ply)
                n32 = out["child_plays"][s, :k].astype(np.float32)
                wr = np.where(n32 > 0, out["child_value"][s, :k] / np.maximum(n32, 1), 0).astype(np.float64)
                assert np.array_equal(wr, g["winrates"][i][:k])
                assert out["root_plays"][s] == g["root_plays"][i] and float(out["root_winrate"][s]) == g["v"][i]
                assert out["action"][s] == g["action"][i], (fname, s, ply)
                assert np.array_equal(plays / plays.sum(), g["prob"][i][:k])
                continue
            plays = out["child_plays"][s, :A].astype(np.float64)
            assert np.array_equal(plays, g["plays"][i]), (fname, s, ply, plays, g["plays"][i])
            n32 = out["child_plays"][s, :A].astype(np.float32)
            wr = np.where(n32 > 0, out["child_value"][s, :A] / np.maximum(n32, 1), 0).astype(np.float64)
            assert np.array_equal(wr, g["winrates"][i]), (fname, s, ply)
            assert out["root_plays"][s] == g["root_plays"][i]
            assert float(out["root_winrate"][s]) == g["v"][i]
            assert out["action"][s] == g["action"][i], (fname, s, ply)
            tot = plays.sum()
            assert np.array_equal(plays / tot if tot > 0 else plays, g["prob"][i])
        acts = np.where(live, g["action"][idx], -1).astype(np.int32)
        if reuse:
            eng.move_roots(acts)
        else:  # the fixture called DropRoot() after every move
            nxt_live = lens > ply + 1
            if nxt_live.any():
                sl = np.where(nxt_live)[0]
                eng.set_roots(st_all[gs[:-1][sl] + ply + 1], slots=sl, game_ids=sl)
    assert eng.counters()["overflow"] == 0
    eng.close()


@pytest.mark.parametrize("key,sims,n_games,n_slots", [("c4", 48, 96, 64), ("ttt", 30, 200, 64), ("c4", 200, 12, 16)])
def test_selfplay_vs_oracle(orc, key, sims, n_games, n_slots):
    """Batched GenerateTrainingSamples on the GPU == the oracle's serial games, example by example."""
    game, og = GAMES[key]
    A = _lib.game_info(game).A
    eng = _lib.Engine(game, n_slots=n_slots, sims_per_move=sims, evaluator=_lib.EVAL_HASH, hash_salt=4242, seed=99,
                      max_games=n_games, first_game_id=1000)
    eng.selfplay_begin(n_games, 1.0)
    guard = 0
    while not eng.selfplay_done()[0]:
        eng.selfplay_step(3)
        guard += 1
        assert guard < 200
    rec, offs, win = eng.fetch_examples()
    cnt = eng.counters()
    assert cnt["overflow"] == 0 and cnt["games_finished"] == n_games
    cfg = orc.make_cfg(og, evaluator=orc.EVAL_HASH, salt=4242, seed=99)
    sims_total = depth_total = 0
    for gidx in range(n_games):
        o = orc.selfplay_game(cfg, 1000 + gidx, 1.0, sims, eng.max_plies)
        r = rec[offs[gidx]:offs[gidx + 1]]
        assert len(r) == o["n"], (gidx, len(r), o["n"])
        assert win[gidx] == o["winner"]
        assert (r["game_id"] == 1000 + gidx).all() and np.array_equal(r["ply"], np.arange(len(r)))
        tot = np.maximum(r["total"].astype(np.float64), 1.0)[:, None]
        assert np.array_equal(r["visits"][:, :A] / tot, o["pi"]), gidx
        assert np.array_equal(r["player"], o["player"])
        assert np.array_equal(r["z"].astype(np.float32), o["z"])
        st = r["state"].copy().view(np.uint64).reshape(-1, 2)
        assert np.array_equal(_lib.game_encode(game, st), o["boards"])
        sims_total += o["stats"].sims
        depth_total += o["stats"].sum_depth
    assert cnt["sims"] == sims_total and cnt["sum_depth"] == depth_total
    assert cnt["examples"] == len(rec)
    eng.close()


def test_rollout_fixed_vs_oracle(orc):
    """BASELINE config 1 shape: TicTacToe, FixedMCTS(maxDepth=10, c=0.85, playLimit=50), rollout evaluator."""
    game, og = GAMES["ttt"]
    n_games = 40
    eng = _lib.Engine(game, n_slots=16, sims_per_move=50, mcts_kind=_lib.MCTS_FIXED, max_depth=10,
                      evaluator=_lib.EVAL_ROLLOUT, seed=31, max_games=n_games)
    eng.selfplay_begin(n_games, 1.0)
    while not eng.selfplay_done()[0]:
        eng.selfplay_step(3)
    rec, offs, win = eng.fetch_examples()
    assert eng.counters()["overflow"] == 0
    cfg = orc.make_cfg(og, kind=orc.FIXED, max_depth=10, evaluator=orc.EVAL_ROLLOUT, seed=31)
    for gidx in range(n_games):
        o = orc.selfplay_game(cfg, gidx, 1.0, 50, 9)
        r = rec[offs[gidx]:offs[gidx + 1]]
        assert len(r) == o["n"] and win[gidx] == o["winner"], gidx
        tot = np.maximum(r["total"].astype(np.float64), 1.0)[:, None]
        assert np.array_equal(r["visits"][:, :9] / tot, o["pi"]), gidx
        assert np.array_equal(r["z"].astype(np.float32), o["z"])
    eng.close()


def test_error_mapping():
    game = _lib.GAME_CONNECT4
    eng = _lib.Engine(game, n_slots=2, sims_per_move=1, evaluator=_lib.EVAL_HASH)
    with pytest.raises(ValueError):
        eng.selfplay_begin(0, 1.0)  # Blackbird.py:235-236
    with pytest.raises(ValueError):
        eng.selfplay_begin(2, 1.0)  # 1 sim on a fresh root -> NaN probabilities (MCTS.py:336-338)
    with pytest.raises(ValueError):
        eng.run_sims(0)  # no stop rule (MCTS.py:181-182)
    st = np.repeat(_lib.game_initial(game), 2, axis=0)
    eng.set_roots(st)
    eng.run_sims(1)
    out = eng.sample_moves(1.0, u=np.array([0.5, 0.5]))
    assert (out["action"] == _lib.ERR_NAN).all() and (out["root_plays"] == 1).all()
    eng.run_sims(1)  # 2 sims: child visits [1,0,...] (SURVEY 8a edge case)
    out = eng.sample_moves(1.0, u=np.array([0.0, 0.99]))
    assert out["child_plays"][0].sum() == 1 and (out["root_plays"] == 2).all()
    eng.run_sims(3)  # tree reuse: playLimit adds to Root.Plays
    out = eng.sample_moves(1.0, u=np.array([0.0, 0.99]))
    assert (out["root_plays"] == 5).all() and out["child_plays"][0].sum() == 4
    eng.close()
    with pytest.raises(ValueError):
        _lib.Engine(game, n_slots=2, sims_per_move=8, mcts_kind=_lib.MCTS_FIXED, max_depth=0)  # FixedMCTS.py:15


def test_dc_selfplay_vs_oracle(orc):
    """DragonChess (compact child lists, double-move turn order): batched self-play == oracle, example by example."""
    game = _lib.GAME_DRAGONCHESS
    n_games, sims, cap = 12, 20, 40
    eng = _lib.Engine(game, n_slots=8, sims_per_move=sims, evaluator=_lib.EVAL_HASH, hash_salt=555, seed=7,
                      max_games=n_games, max_plies=cap)
    eng.selfplay_begin(n_games, 1.0)
    guard = 0
    while not eng.selfplay_done()[0]:
        eng.selfplay_step(4)
        guard += 1
        assert guard < 100
    rec, offs, win = eng.fetch_examples()
    assert eng.counters()["overflow"] == 0
    cfg = orc.make_cfg(orc.DC, evaluator=orc.EVAL_HASH, salt=555, seed=7)
    for gidx in range(n_games):
        o = orc.selfplay_game(cfg, gidx, 1.0, sims, cap)
        r = rec[offs[gidx]:offs[gidx + 1]]
        assert len(r) == o["n"], (gidx, len(r), o["n"])
        assert win[gidx] == o["winner"]
        for k in range(len(r)):
            pi = np.zeros(4032)
            nch = int(r["n_children"][k])
            if r["total"][k] > 0:
                pi[r["action"][k][:nch]] = r["visits"][k][:nch] / float(r["total"][k])
            assert np.array_equal(pi, o["pi"][k]), (gidx, k)
        assert np.array_equal(r["player"], o["player"]) and np.array_equal(r["z"].astype(np.float32), o["z"])
        assert np.array_equal(_lib.game_encode(game, np.ascontiguousarray(r["state"])), o["boards"])
    eng.close()


def test_dc_selfplay_one_wave_per_game(orc):
    """DragonChess with the network evaluator: the default launch structure (one wave keeps its game for a whole launch,
    tree step and network in the same wave: selfplay_mode 5) must give the examples of the launch-per-simulation
    structure byte for byte, with and without prior noise, and -- noise off -- the oracle's search fed the GPU network's
    outputs.  Ragged slot count (5 slots, 4 waves per workgroup) and slot reuse (7 games)."""
    from blackbird_amd import weights as W
    game = _lib.GAME_DRAGONCHESS
    n_games, sims, cap = 7, 12, 20
    flat = W.flatten(W.init_weights(17, 16, 2, 16, 4032, seed=5))

    def run(launch, noise):
        eng = _lib.Engine(game, n_slots=5, sims_per_move=sims, evaluator=_lib.EVAL_NET, seed=7, max_games=n_games,
                          max_plies=cap, noise_on=noise, launch=launch)
        eng.load_weights(flat)
        mode = eng.selfplay_mode()
        eng.selfplay_begin(n_games, 1.0)
        guard = 0
        while not eng.selfplay_done()[0]:
            eng.selfplay_step(3)
            guard += 1
            assert guard < 100
        rec, offs, win = eng.fetch_examples()
        assert eng.counters()["overflow"] == 0
        eng.close()
        return rec, offs, win, mode

    for noise in (True, False):
        a, b = run(_lib.LAUNCH_AUTO, noise), run(_lib.LAUNCH_LOCKSTEP, noise)
        assert (a[3], b[3]) == (5, 0)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[0].tobytes() == b[0].tobytes(), noise
    rec, offs, win, _m = a  # noise off
    ev = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
    ev.load_weights(flat)

    def cb(_ctx, stp, vp, pp):
        v, _l, p = ev.net_eval(planes=orc.encode(orc.DC, stp.contents))
        vp[0] = float(v[0])
        if pp:
            for i in range(4032):
                pp[i] = float(p[0, i])

    cfg = orc.make_cfg(orc.DC, evaluator=orc.EVAL_CALLBACK, seed=7, cb=orc.EVAL_CB(cb))
    for gidx in range(n_games):
        o = orc.selfplay_game(cfg, gidx, 1.0, sims, cap)
        r = rec[offs[gidx]:offs[gidx + 1]]
        assert len(r) == o["n"] and win[gidx] == o["winner"], gidx
        for k in range(len(r)):
            pi = np.zeros(4032)
            nch = int(r["n_children"][k])
            if r["total"][k] > 0:
                pi[r["action"][k][:nch]] = r["visits"][k][:nch] / float(r["total"][k])
            assert np.array_equal(pi, o["pi"][k]), (gidx, k)
    ev.close()


@pytest.mark.parametrize("game,og,n_slots,n_games,sims,cap", [
    (_lib.GAME_CONNECT4, 0, 19, 24, 40, 43),      # persistent work-queue kernel (mode 3)
    (_lib.GAME_DRAGONCHESS, 2, 5, 6, 16, 14),     # one wave per game (mode 5)
])
def test_exact_ties_follow_the_float64_first_maximum(orc, game, og, n_slots, n_games, sims, cap):
    """The float32 pre-filter of the PUCT argmax (tree.hip.h grp_argmax_puct, tree_dc.hip.h select) must hand every tie to
    the float64 path.  An all-zero network makes ties the rule: every prior is 1 / legal moves, every value 0.5, so
    unvisited children tie EXACTLY and np.argmax's first-maximum decides -- the engine's visit counts, moves and winners
    must be the oracle's, example by example (noise off; the oracle's callback gets the engine's own network outputs)."""
    from blackbird_amd import weights as W
    gi = _lib.game_info(game)
    w = W.init_weights(gi.C, 16, 2, 16, gi.A, seed=3)
    # (batch-norm variances stay at one; kernels, biases, gammas, betas and means are zero)
    wz = {k: (np.ones_like(v) if "variance" in k else np.zeros_like(v)) for k, v in w.items()}
    flat = W.flatten(wz)
    eng = _lib.Engine(game, n_slots=n_slots, sims_per_move=sims, evaluator=_lib.EVAL_NET, seed=11, max_games=n_games,
                      max_plies=cap, noise_on=False)
    eng.load_weights(flat)
    assert eng.selfplay_mode() in (3, 5)
    eng.selfplay_begin(n_games, 1.0)
    guard = 0
    while not eng.selfplay_done()[0]:
        eng.selfplay_step(3)
        guard += 1
        assert guard < 200 and eng.counters()["overflow"] == 0
    rec, offs, win = eng.fetch_examples()
    eng.close()
    ev = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET)
    ev.load_weights(flat)
    st0 = _lib.game_initial(game)
    v0, _l0, p0 = ev.net_eval(states=st0)
    assert float(v0[0]) == 0.0 and np.ptp(p0[0]) == 0.0  # the ties are real: uniform policy, zero value

    def cb(_ctx, stp, vp, pp):
        v, _l, p = ev.net_eval(planes=orc.encode(og, stp.contents))
        vp[0] = float(v[0])
        if pp:
            for i in range(gi.A):
                pp[i] = float(p[0, i])

    cfg = orc.make_cfg(og, evaluator=orc.EVAL_CALLBACK, seed=11, cb=orc.EVAL_CB(cb))
    for gidx in range(n_games):
        o = orc.selfplay_game(cfg, gidx, 1.0, sims, cap - 1 if gi.dense else cap)
        r = rec[offs[gidx]:offs[gidx + 1]]
        assert len(r) == o["n"] and win[gidx] == o["winner"], gidx
        for k in range(len(r)):
            pi = np.zeros(gi.A)
            if r["total"][k] > 0:
                if gi.dense:
                    pi[:] = r["visits"][k][:gi.A] / float(r["total"][k])
                else:
                    nch = int(r["n_children"][k])
                    pi[r["action"][k][:nch]] = r["visits"][k][:nch] / float(r["total"][k])
            assert np.array_equal(pi, o["pi"][k]), (gidx, k)
        assert np.array_equal(r["player"], o["player"]) and np.array_equal(r["z"].astype(np.float32), o["z"])
    ev.close()
