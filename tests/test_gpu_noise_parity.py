"""-m gpu: the default self-play kernel in the EXACT configuration bench.py measures -- network evaluator with the
graph's Beta(alpha 0.2, 1 - alpha) prior noise mixed in at eps 0.3 (NetworkFactory.py:176-182), the persistent
work-queue kernel k_selfplay_queue -- against the oracle's search, example by example.

How the noise is made comparable: the noise of a node is a pure function of (Philox key, global game id, node serial,
action) in the RNG spec both sides share (the serial is the order in which the search first reaches nodes).
The oracle's keyed callback evaluator hands (game id, node serial) of every node it evaluates to `bb_net_eval_keyed`,
which returns the very priors the engine's network waves computed for that node (same head code, same Philox stream), and
the oracle -- Model.GetPriors' masking and renormalisation, PUCT, backup, move sampling, all restated from the reference
-- has to arrive at the same visit counts, moves, winners and z, bit for bit.

Second part: the launch structures (persistent work queue / asynchronous rounds / lock-step launches) must produce
byte-identical example records with the noise on (the former tools/queue_cmp.py, now part of the suite)."""
import numpy as np
import pytest

from blackbird_amd import _lib, weights as W

pytestmark = pytest.mark.gpu
ALPHA, EPS = 0.2, 0.3


def _selfplay(game, n_slots, n_games, sims, blocks, seed, first_id, noise=True, launch=_lib.LAUNCH_AUTO):
    gi = _lib.game_info(game)
    flat = W.flatten(W.init_weights(gi.C, 16, blocks, 16, gi.A, seed=21, perturb=True))
    eng = _lib.Engine(game, n_slots=n_slots, sims_per_move=sims, evaluator=_lib.EVAL_NET, seed=seed, max_games=n_games,
                      first_game_id=first_id, noise_on=noise, alpha=ALPHA, epsilon=EPS, launch=launch)
    eng.load_weights(flat)
    mode = eng.selfplay_mode()
    eng.selfplay_begin(n_games, 1.0)
    guard = 0
    while not eng.selfplay_done()[0]:
        eng.selfplay_step(2)
        guard += 1
        assert guard < 400 and eng.counters()["overflow"] == 0
    rec, offs, win = eng.fetch_examples()
    cnt = eng.counters()
    eng.close()
    return flat, rec, offs, win, cnt, mode


@pytest.mark.parametrize("game,og,n_slots,n_games,sims,max_plies,blocks", [
    (_lib.GAME_CONNECT4, 0, 19, 30, 48, 43, 4),    # ragged: 19 slots = one full 16-game workgroup + 3, every slot reused
    (_lib.GAME_TICTACTOE, 1, 21, 50, 24, 10, 2),
])
def test_queue_kernel_with_prior_noise_matches_oracle(orc, game, og, n_slots, n_games, sims, max_plies, blocks):
    gi = _lib.game_info(game)
    seed, first_id = 17, 500
    flat, rec, offs, win, cnt, mode = _selfplay(game, n_slots, n_games, sims, blocks, seed, first_id)
    assert mode == 3, "the default launch structure for a 16-filter network is the persistent work-queue kernel"
    ev = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET, seed=seed, alpha=ALPHA, epsilon=EPS)
    ev.load_weights(flat)
    calls = {"n": 0, "with_policy": 0}

    def getpolicy(_ctx, stp, gid, serial, vp, pp):
        planes = orc.encode(og, stp.contents)
        v, _l, p = ev.net_eval_keyed([gid], [serial], planes=planes)
        vp[0] = float(v[0])
        calls["n"] += 1
        if pp:
            calls["with_policy"] += 1
            for a in range(gi.A):
                pp[a] = float(p[0, a])

    cfg = orc.make_cfg(og, evaluator=orc.EVAL_CALLBACK_KEYED, seed=seed, cb2=orc.EVAL_CB2(getpolicy))
    sims_total = 0
    for gidx in range(n_games):
        o = orc.selfplay_game(cfg, first_id + gidx, 1.0, sims, max_plies - 1)
        r = rec[offs[gidx]:offs[gidx + 1]]
        assert len(r) == o["n"] and win[gidx] == o["winner"], gidx
        assert (r["game_id"] == first_id + gidx).all()
        tot = np.maximum(r["total"].astype(np.float64), 1.0)[:, None]
        assert np.array_equal(r["visits"][:, :gi.A] / tot, o["pi"]), gidx
        assert np.array_equal(r["player"], o["player"]) and np.array_equal(r["z"].astype(np.float32), o["z"])
        sims_total += o["stats"].sims
    assert cnt["sims"] == sims_total and calls["with_policy"] > 0
    # the noise really is in those priors: the keyed policy differs from the clean softmax and from another node's draw
    st = _lib.game_initial(game)
    clean = ev.net_eval(states=st)[2]
    n0 = ev.net_eval_keyed([first_id], [0], states=st)[2]
    n1 = ev.net_eval_keyed([first_id], [1], states=st)[2]
    again = ev.net_eval_keyed([first_id], [0], states=st)[2]
    assert np.array_equal(n0, again) and not np.array_equal(n0, n1) and not np.array_equal(n0, clean)
    assert abs(float(n0.sum()) - 1.0) < 1e-5
    ev.close()


def test_keyed_noise_is_the_oracles_beta_stream(orc):
    """bb_net_eval_keyed mixes Beta(alpha, 1-alpha) draws of the shared Philox spec: undo the mix and compare with the
    oracle's orc_beta_noise (library powf there, v_log/v_exp here: 1e-5)."""
    game = _lib.GAME_CONNECT4
    gi = _lib.game_info(game)
    ev = _lib.Engine(game, n_slots=4, sims_per_move=2, evaluator=_lib.EVAL_NET, seed=99, alpha=ALPHA, epsilon=EPS)
    ev.load_weights(W.flatten(W.init_weights(gi.C, 16, 4, 16, gi.A, seed=2)))
    st = np.repeat(_lib.game_initial(game), 64, axis=0)
    gids = np.arange(64, dtype=np.uint32) * 7919 + 3
    sers = (np.arange(64, dtype=np.int32) * 31) % 1000
    clean = ev.net_eval(states=st[:1])[2][0].astype(np.float64)
    noisy = ev.net_eval_keyed(gids, sers, states=st)[2].astype(np.float64)
    for i in range(64):
        nz = np.array([orc.lib().orc_beta_noise(99, int(gids[i]), int(sers[i]), a, ALPHA) for a in range(gi.A)])
        q = (1 - EPS) * clean + EPS * nz
        assert np.max(np.abs(noisy[i] - q / q.sum())) <= 1e-5, i
    ev.close()


@pytest.mark.parametrize("game,n_slots,n_games,sims", [(_lib.GAME_CONNECT4, 37, 60, 40), (_lib.GAME_TICTACTOE, 16, 40, 24)])
def test_launch_structures_are_byte_identical_with_noise(game, n_slots, n_games, sims):
    """Per game the sequence of simulations is the sequential one whatever the launch structure, and the random streams
    are keyed by game / ply / node: work queue (3) == asynchronous rounds (1) == lock-step launches (0), noise on."""
    runs = {}
    for name, launch in (("queue", _lib.LAUNCH_AUTO), ("rounds", _lib.LAUNCH_ROUNDS), ("lockstep", _lib.LAUNCH_LOCKSTEP)):
        runs[name] = _selfplay(game, n_slots, n_games, sims, 4, 5, 0, launch=launch)
    assert [runs[k][5] for k in ("queue", "rounds", "lockstep")] == [3, 1, 0]
    a = runs["queue"]
    for other in ("rounds", "lockstep"):
        b = runs[other]
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]), other
        assert a[1].tobytes() == b[1].tobytes(), other
        assert a[4]["sims"] == b[4]["sims"] and a[4]["sum_depth"] == b[4]["sum_depth"], other
