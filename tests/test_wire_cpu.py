"""Example wire format (state.proto) and the sqlite sink: host logic, checked against the blobs the
reference's own GenerateTrainingSamples + state_pb2 produced (tests/golden/selfplay_*.npz)."""
import os

import numpy as np
import pytest

from blackbird_amd import proto_wire
from blackbird_amd.Blackbird import ExampleState


@pytest.mark.parametrize("key,shape", [("c4", (1, 6, 7, 3)), ("ttt", (1, 3, 3, 3))])
def test_serialize_matches_reference_bytes(golden_dir, key, shape):
    g = np.load(os.path.join(golden_dir, f"selfplay_{key}.npz"), allow_pickle=False)
    pos = 0
    for i, ln in enumerate(g["blob_len"]):
        blob = g["blob"][pos:pos + ln].tobytes()
        pos += ln
        ex = ExampleState(float(g["z"][i]), g["pi"][i].astype(np.float64), g["enc"][i].reshape(shape))
        assert ex.SerializeState() == blob, i
        back = ExampleState.FromSerialized(blob)
        assert back.MctsEval == (float(g["z"][i]),)  # 1-tuple, as in the reference (Blackbird.py:60)
        assert np.array_equal(back.MctsPolicy, g["pi"][i]) and np.array_equal(back.Board, g["enc"][i].reshape(shape))
    assert pos == len(g["blob"])
    assert set(g["blob_len"].tolist()) <= ({200, 195} if key == "c4" else {117, 112})  # SURVEY 8a known answers


def test_zero_eval_is_omitted():
    # proto3: a 0.0 float is not written -> 5 bytes shorter (draws)
    ex = ExampleState(0.0, np.zeros(7), np.zeros((1, 6, 7, 3), dtype=np.int8))
    assert len(ex.SerializeState()) == 195
    f = proto_wire.decode_state(ex.SerializeState())
    assert f["mctsEval"] == 0.0 and len(f["mctsPolicy"]) == 56 and f["boardDims"] == bytes([1, 6, 7, 3])


def test_datamanager_roundtrip(tmp_path):
    from blackbird_amd.DataManager import Connection
    c = Connection(directory=str(tmp_path))
    assert c.GetLastVersion("Connect4", "m") == 1
    c.PutGames("m", 1, "Connect4", [b"a", b"bc"])
    assert c.GetGames("m", 1) == [b"a", b"bc"]
    c.PutTrainingStatistic(1, "m", 1, "RANDOM")
    c.PutModel("Connect4", "m", 2)
    assert c.GetLastVersion("Connect4", "m") == 2
    c.Close()
    c2 = Connection(directory=str(tmp_path))  # existing schema is reused
    assert c2.GetGames("m", 1) == [b"a", b"bc"]
    c2.Close()


def test_training_loss_matches_reference_formulas():
    """NetworkFactory.py:185-201 restated in numpy for a tiny batch (noise fixed)."""
    import torch
    from blackbird_amd import weights as W
    from blackbird_amd.training import Trainer
    w = W.init_weights(3, 16, 1, 16, 7, seed=4, perturb=True)
    tr = Trainer(w, alpha=0.2, epsilon=0.3, device="cpu")
    rng = np.random.RandomState(0)
    B = 5
    boards = rng.randint(-1, 2, size=(B, 6, 7, 3)).astype(np.float32)
    ev = rng.uniform(-1, 1, B).astype(np.float32)
    pl = rng.dirichlet(np.ones(7), B).astype(np.float32)
    noise = torch.tensor(rng.beta(0.2, 0.8, 7).astype(np.float32))
    total, (le, lp, l2) = tr.loss(torch.tensor(boards), torch.tensor(ev), torch.tensor(pl), noise=noise)
    value, logits = tr.forward(torch.tensor(boards))
    value, logits = value.detach().numpy().astype(np.float64), logits.detach().numpy().astype(np.float64)
    sm = np.exp(logits - logits.max(1, keepdims=True))
    sm /= sm.sum(1, keepdims=True)
    pol = 0.7 * sm + 0.3 * noise.numpy()[None, :]
    pol /= pol.sum()
    assert np.isclose(float(le), np.mean((value - ev) ** 2), rtol=1e-5)
    assert np.isclose(float(lp), -np.mean(np.log(pol) @ pl.T), rtol=1e-4)
    l2_ref = np.mean([0.5 * np.sum(np.asarray(v, np.float64) ** 2) for k, v in w.items()
                      if "bias" not in k and "moving" not in k])
    assert np.isclose(float(l2), l2_ref, rtol=1e-5)
    before = tr.export()
    tr.step(boards, ev, pl, 1e-3)
    after = tr.export()
    assert any(not np.array_equal(before[k], after[k]) for k in before if "moving" not in k)
    assert all(np.array_equal(before[k], after[k]) for k in before if "moving" in k)


def test_batch_encoder_equals_per_example_encoder():
    """proto_wire.encode_states_batch (used by GenerateTrainingSamples for all examples at once) is byte for byte the
    per-example encoder, including the omitted +0.0 eval, the kept -0.0, and the int8-wrapped 4032 policy dim."""
    from blackbird_amd import proto_wire
    rng = np.random.RandomState(0)
    for A, shape in ((7, (6, 7, 3)), (9, (3, 3, 3)), (4032, (8, 8, 17))):
        K = 40
        z = rng.choice([-1.0, 0.0, 1.0, -0.0], K).astype(np.float32)
        pi = rng.rand(K, A)
        pi[::5] = 0
        boards = rng.randint(-1, 2, (K, 1) + shape).astype(np.int8)
        blobs = proto_wire.encode_states_batch(z, pi, boards)
        for i in range(K):
            pdims = np.array(pi[i].shape, dtype=np.int64).astype(np.int8)
            ref = proto_wire.encode_state(z[i], pi[i].tobytes(), boards[i].tobytes(),
                                          np.array(boards[i].shape, dtype=np.int8).tobytes(), pdims.tobytes())
            assert ref == blobs[i]
