"""N > 1 path on CPU: two gloo ranks exchange variable-length example shards (the epoch-end
all-gather of SURVEY.md 8e) and get identical, rank-ordered results; shards own disjoint game ids."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from blackbird_amd import dist as bdist

DT = np.dtype([("game_id", "<u4"), ("ply", "<u2"), ("player", "u1"), ("z", "i1"), ("total", "<u4"),
               ("n_children", "<u4"), ("state", "u1", (16,)), ("visits", "<u4", (8,))])


def make_records(rank):
    first, seed = bdist.shard(rank)
    rng = np.random.RandomState(seed)
    n = 37 + 50 * rank  # different length per rank
    rec = np.zeros(n, dtype=DT)
    rec["game_id"] = first + np.arange(n) // 7
    rec["ply"] = np.arange(n) % 7
    rec["visits"] = rng.randint(0, 800, size=(n, 8))
    rec["z"] = rng.choice([-1, 0, 1], n)
    return rec


def make_store(rank, empty=False):
    """A fake engine example store + game headers (what bb_examples_device exposes): 9 games of up to 6 records of 64 bytes;
    ragged lengths, unfinished games in between (their records must not travel), rank 1 optionally with nothing finished."""
    rng = np.random.RandomState(100 + rank)
    ng, per, rb = 9, 6, 64
    store = rng.randint(0, 256, size=(ng, per, rb)).astype(np.uint8)
    hdr = np.zeros((ng, 4), dtype=np.int32)
    hdr[:, 0] = rng.randint(1, per + 1, ng)          # n_examples
    hdr[:, 3] = (np.arange(ng) + rank) % 3 != 0      # done (every third game still running)
    if empty:
        hdr[:, 3] = 0
    want = np.concatenate([store[g, :hdr[g, 0]] for g in range(ng) if hdr[g, 3]] or [np.zeros((0, rb), np.uint8)])
    return store, hdr, want


def worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    allrec = bdist.allgather_records(make_records(rank))
    np.save(os.path.join(out, f"r{rank}.npy"), allrec.view(np.uint8))
    # a rank without any finished game still takes part (zero-length shard)
    some = bdist.allgather_records(make_records(rank)[:0] if rank == 0 else make_records(rank))
    np.save(os.path.join(out, f"e{rank}.npy"), some.view(np.uint8))
    # bench.py's reduction: whole-job totals are sums over ranks, the wall time is the slowest rank's
    sums, maxima = bdist.reduce_totals([100.0 + rank, 7.0 * (rank + 1), 3.0], [1.5 + 0.25 * rank])
    np.save(os.path.join(out, f"t{rank}.npy"), np.array(sums + maxima))
    # the device path's compaction + exchange (dist.allgather_store: what allgather_engine_examples does with the engine's
    # store) on fake stores: ragged ranks, then rank 1 without a single finished game
    import torch
    for tag, empty in (("s", False), ("z", True)):
        store, hdr, _want = make_store(rank, empty=empty and rank == 1)
        allrec, counts = bdist.allgather_store(torch.from_numpy(store), torch.from_numpy(hdr))
        np.save(os.path.join(out, f"{tag}{rank}.npy"), allrec.numpy())
        np.save(os.path.join(out, f"{tag}c{rank}.npy"), np.array(counts))
    dist.destroy_process_group()


def test_allgather_two_ranks(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [np.load(os.path.join(tmp_path, f"r{r}.npy")).view(DT) for r in range(2)]
    want = np.concatenate([make_records(0), make_records(1)])
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want)
    ids0, ids1 = set(make_records(0)["game_id"]), set(make_records(1)["game_id"])
    assert not (ids0 & ids1)
    for r in range(2):
        assert np.array_equal(np.load(os.path.join(tmp_path, f"e{r}.npy")).view(DT), make_records(1))
        assert np.array_equal(np.load(os.path.join(tmp_path, f"t{r}.npy")), [201.0, 21.0, 6.0, 1.75])
        w0, w1 = make_store(0)[2], make_store(1)[2]
        assert np.array_equal(np.load(os.path.join(tmp_path, f"s{r}.npy")), np.concatenate([w0, w1]))
        assert list(np.load(os.path.join(tmp_path, f"sc{r}.npy"))) == [len(w0), len(w1)]
        assert np.array_equal(np.load(os.path.join(tmp_path, f"z{r}.npy")), w0)          # rank 1 contributed nothing
        assert list(np.load(os.path.join(tmp_path, f"zc{r}.npy"))) == [len(w0), 0]


def test_shards_are_disjoint():
    firsts = [bdist.shard(r)[0] for r in range(8)]
    seeds = [bdist.shard(r)[1] for r in range(8)]
    assert len(set(firsts)) == 8 and len(set(seeds)) == 8
    assert all(b - a >= 40_000_000 for a, b in zip(firsts, firsts[1:]))
