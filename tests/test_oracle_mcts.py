"""Oracle tree search (oracle/orc_mcts.c) vs golden vectors produced by the reference's
MCTS.py / DynamicMCTS.py / FixedMCTS.py + Model.SampleValue/GetPriors bodies under the
deterministic hash evaluator (tests/make_golden.py part 'mcts')."""
import glob
import os

import numpy as np
import pytest

KEYS = {"c4": 0, "ttt": 1, "dc": 2}
FILES = sorted(os.path.basename(p) for p in glob.glob(
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mcts_*.npz")))


def dense(game_A, sparse_rows, vals):
    out = np.zeros(game_A)
    for (idx, _), v in zip(sparse_rows, vals):
        if idx >= 0:
            out[int(idx)] = v
    return out


@pytest.mark.parametrize("fname", FILES)
def test_mcts_golden(orc, golden_dir, fname):
    g = np.load(os.path.join(golden_dir, fname), allow_pickle=False)
    key = fname.split("_")[1]
    game = KEYS[key]
    A = orc.dims(game)[3]
    sims, seed, salt, max_depth, fixed, reuse = [int(x) for x in g["meta"]]
    c, temp = [float(x) for x in g["cfg"]]
    gs = g["game_start"]
    for gi in range(len(gs) - 1):
        cfg = orc.make_cfg(game, kind=orc.FIXED if fixed else orc.DYNAMIC, evaluator=orc.EVAL_HASH,
                           c_puct=c, max_depth=max_depth, salt=salt + gi, priors_ones=bool(fixed))
        s = orc.Search(cfg)
        for i in range(gs[gi], gs[gi + 1]):
            st = orc.state_from_arrays(game, g["board"][i], g["player"][i], g["prev"][i] or None,
                                       g["castle"][i])
            r = s.find_move(st, temp, sims, u=float(g["u"][i]))
            if key == "dc":
                plays = dense(A, g["plays"][i], g["plays"][i][:, 1])
                wr = dense(A, g["plays"][i], g["winrates"][i])
                prob = dense(A, g["plays"][i], g["prob"][i])
            else:
                plays, wr, prob = g["plays"][i], g["winrates"][i], g["prob"][i]
            assert np.array_equal(r["plays"], plays), (fname, gi, i, r["plays"], plays)
            assert np.array_equal(r["winrates"], wr), (fname, gi, i)
            assert np.array_equal(r["prob"], prob)
            assert r["root_plays"] == g["root_plays"][i]
            assert r["winrate"] == g["v"][i]
            assert r["action"] == g["action"][i], (fname, gi, i)
            if reuse:
                s.move_root(r["next"])
            else:
                s.drop_root()


def test_sampling_law(orc):
    # np.random.choice(len(p), p=p) == searchsorted(cumsum(p)/cumsum(p)[-1], u, 'right')
    rng = np.random.RandomState(3)
    for _ in range(300):
        A = int(rng.choice([7, 9]))
        plays = rng.randint(0, 50, A).astype(np.float64)
        if plays.sum() == 0:
            continue
        temp = float(rng.choice([1.0, 0.1, 0.5, 2.0]))
        st = rng.get_state()
        u = rng.random_sample()
        rng.set_state(st)
        allp = sum([p ** (1 / temp) for p in plays])
        p = [c ** (1 / temp) / allp for c in plays]
        want = rng.choice(len(plays), p=p)
        assert orc.sample_action(plays, temp, u) == want


def test_edge_cases(orc):
    # playLimit=1 on a fresh root -> 0/0 -> ValueError; 2 sims -> [1,0,...]; reuse adds playLimit
    cfg = orc.make_cfg(0, salt=7)
    s = orc.Search(cfg)
    st = orc.new_state(0)
    with pytest.raises(ValueError):
        s.find_move(st, 1.0, 1, u=0.5)
    s = orc.Search(cfg)
    r = s.find_move(st, 1.0, 2, u=0.0)
    assert r["root_plays"] == 2 and r["plays"].sum() == 1 and r["winrate"] == 0.0
    r = s.find_move(st, 1.0, 3, u=0.0)
    assert r["root_plays"] == 5 and r["plays"].sum() == 4
    # root mismatch -> AssertionError
    other = st.copy()
    orc.apply(0, other, 0)
    with pytest.raises(AssertionError):
        s.find_move(other, 1.0, 2, u=0.0)
    # FixedMCTS(maxDepth=3), 1 sim: root + 3 levels x 7 children = 22 nodes
    cfg = orc.make_cfg(0, kind=orc.FIXED, max_depth=3, evaluator=orc.EVAL_ROLLOUT)
    s = orc.Search(cfg)
    s.find_move(st, 1.0, 1, u=0.5)
    assert s.stats().nodes == 22
