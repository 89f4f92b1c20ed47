"""Multi-GPU plumbing.  Self-play shards by game: rank r owns a disjoint range of
global game ids (=> disjoint Philox streams) and runs its own engine; nothing crosses GPUs while
games are played.  The one exchange step is the epoch-end all-gather of the (s, pi, z) example
records -- variable length per rank, so sizes are gathered first and the payload is padded to the
maximum (RCCL has no all-gather-v).  torch.distributed is used for the collective only
(backend "nccl" == RCCL over xGMI on MI355X; "gloo" in the CPU tests).

On the GPU the records never visit the host on their way out: `engine_records_device` wraps the engine's example
store (bb_examples_device) as a torch tensor without copying, compacts the finished games' records on the device,
and `allgather_bytes` hands that tensor to RCCL; only the gathered result is brought to the host, by whoever consumes it
(the sqlite sink, the training set)."""
import numpy as np

GAME_ID_STRIDE = 50_000_000  # global game ids of rank r start at r * GAME_ID_STRIDE


def shard(rank, base_seed=1234):
    """Per-rank engine settings: (first_game_id, seed)."""
    return rank * GAME_ID_STRIDE, base_seed + rank


class _DeviceMemory(object):
    """A span of device memory owned by someone else (the engine), exposed through __cuda_array_interface__ so that
    torch.as_tensor() views it in place."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def compact_records(store, hdr):
    """store: uint8 tensor [max_games, max_plies + 1, record_bytes] (the engine's example store); hdr: int32 tensor
    [max_games, 4] = (n_examples, winner, plies, done).  Returns the records of the FINISHED games in (game, ply) order as a
    fresh uint8 tensor [n_records, record_bytes] on the same device -- a boolean-mask compaction, no host copy."""
    import torch
    per = store.shape[1]
    n_ex = torch.where(hdr[:, 3] != 0, hdr[:, 0], torch.zeros_like(hdr[:, 0]))
    keep = torch.arange(per, device=store.device, dtype=torch.int32)[None, :] < n_ex[:, None]
    return store[keep]


def engine_records_device(eng, device):
    """The finished games' example records of `eng` as a uint8 tensor [n_records, record_bytes] ON `device`, in
    (game, ply) order -- the device-side twin of Engine.fetch_examples.  No host copy: the example store and the game
    headers are viewed in place (bb_examples_device) and compacted on the device."""
    import torch
    eng.synchronize()
    rec_ptr, _nbytes, rb, hdr_ptr = eng.examples_device()
    ng, per = int(eng.cfg.max_games), int(eng.max_plies) + 1
    dev = torch.device(device)
    store = torch.as_tensor(_DeviceMemory(rec_ptr, (ng, per, rb), "|u1"), device=dev)
    hdr = torch.as_tensor(_DeviceMemory(hdr_ptr, (ng, 4), "<i4"), device=dev)
    return compact_records(store, hdr)


def allgather_bytes(payload):
    """All-gather a 1-D uint8 tensor of rank-dependent length across the default process group, on the device the tensor
    lives on: sizes first, then the payload padded to the longest.  Returns (concatenation in rank order, sizes)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    payload = payload.reshape(-1)
    n_local = torch.tensor([payload.numel()], dtype=torch.int64, device=payload.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.uint8, device=payload.device)
    pad[:payload.numel()] = payload
    gathered = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(gathered, pad)
    return torch.cat([g[:n] for g, n in zip(gathered, sizes)]), sizes


def allgather_store(store, hdr):
    """compact_records + allgather_bytes for one example store per rank: (tensor [N, record_bytes] of every rank's finished
    games' records in rank order, per-rank record counts).  The part of the epoch-end exchange that does not need an engine
    (tests/test_dist_cpu.py runs it on fake stores: ragged and empty ranks)."""
    rec = compact_records(store, hdr)
    rb = int(store.shape[2])
    flat, sizes = allgather_bytes(rec)
    return flat.reshape(-1, rb), [s // rb for s in sizes]


def allgather_engine_examples(eng, device, dtype=None):
    """Epoch-end exchange for one engine per rank: device records -> RCCL all-gather -> (device tensor [N, record_bytes]
    of every rank's records in rank order, per-rank record counts).  With `dtype` (Engine example dtype) the result is
    also returned as a host structured array for host-side consumers."""
    rec = engine_records_device(eng, device)
    rb = rec.shape[1] if rec.dim() == 2 else int(eng.examples_device()[2])
    flat, sizes = allgather_bytes(rec)
    out = flat.reshape(-1, rb)
    counts = [s // rb for s in sizes]
    if dtype is None:
        return out, counts
    return out, counts, out.cpu().numpy().reshape(-1).view(dtype)


def allgather_records(records, device=None):
    """All-gather a 1-D numpy structured array (engine example records already on the host) across the default
    process group.  Returns the concatenation in rank order (identical on every rank).  Same collective as the device
    path (`allgather_bytes`); this entry is for host-resident records and the CPU (gloo) tests."""
    import torch
    raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    flat, _sizes = allgather_bytes(torch.from_numpy(raw.copy()).to(dev))
    return flat.cpu().numpy().view(records.dtype)


def reduce_totals(sums, maxima, device=None):
    """bench.py's cross-rank reduction: element-wise SUM of `sums` (games, simulations, plies, ...) and MAX of `maxima`
    (wall times) over all ranks.  Returns (list of summed floats, list of maxima)."""
    import torch
    import torch.distributed as dist
    dev = torch.device(device) if device is not None else torch.device("cpu")
    s = torch.tensor(list(sums), dtype=torch.float64, device=dev)
    m = torch.tensor(list(maxima), dtype=torch.float64, device=dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return [float(x) for x in s.tolist()], [float(x) for x in m.tolist()]
