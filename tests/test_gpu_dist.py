"""-m gpu: the device-resident half of the epoch-end exchange (blackbird_amd/dist.py) on one GPU: the engine's
example store viewed in place as torch tensors (bb_examples_device -> __cuda_array_interface__), compacted on the device,
must be the records bb_examples_fetch returns; and the collective itself on a one-rank process group (RCCL when the
backend is available, gloo otherwise) must hand the same bytes back.  The multi-rank behaviour of the same code is
covered on CPU by tests/test_dist_cpu.py (two gloo ranks)."""
import os
import socket

import numpy as np
import pytest

from blackbird_amd import _lib, dist as bdist

pytestmark = pytest.mark.gpu


def _played_engine():
    eng = _lib.Engine(_lib.GAME_CONNECT4, n_slots=24, sims_per_move=16, evaluator=_lib.EVAL_HASH, hash_salt=5, seed=3,
                      max_games=60, first_game_id=bdist.shard(3)[0])
    eng.selfplay_begin(60, 1.0)
    for _ in range(30):          # not all 60 games finish: unfinished rows must be left out
        eng.selfplay_step(1)
    return eng


def test_device_view_of_the_example_store_equals_the_host_fetch():
    import torch
    eng = _played_engine()
    rec, offs, win = eng.fetch_examples()
    dev = bdist.engine_records_device(eng, "cuda:0")
    assert dev.is_cuda and dev.dtype == torch.uint8 and dev.shape == (len(rec), rec.dtype.itemsize)
    assert dev.cpu().numpy().tobytes() == rec.tobytes()
    assert 0 < len(rec) and eng.selfplay_done()[1] < 60
    eng.close()


def test_allgather_of_device_records_on_a_one_rank_group():
    import torch
    import torch.distributed as dist
    eng = _played_engine()
    rec, _offs, _win = eng.fetch_examples()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        out, counts, host = bdist.allgather_engine_examples(eng, "cuda:0", dtype=rec.dtype)
        assert out.is_cuda and counts == [len(rec)]
        assert host.tobytes() == rec.tobytes()
        sums, maxima = bdist.reduce_totals([3.0, 4.0], [1.25], device="cuda:0")
        assert sums == [3.0, 4.0] and maxima == [1.25]
    finally:
        dist.destroy_process_group()
    eng.close()
