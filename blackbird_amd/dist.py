"""Multi-GPU plumbing.  Self-play shards by game: rank r owns a disjoint range of
global game ids (=> disjoint Philox streams) and runs its own engine; nothing crosses GPUs while
games are played.  The one exchange step is the epoch-end all-gather of the (s, pi, z) example
records -- variable length per rank, so sizes are gathered first and the payload is padded to the
maximum (RCCL has no all-gather-v).  torch.distributed is used for the collective only
(backend "nccl" == RCCL over xGMI on MI355X; "gloo" in the CPU tests)."""
import numpy as np

GAME_ID_STRIDE = 50_000_000  # global game ids of rank r start at r * GAME_ID_STRIDE


def shard(rank, base_seed=1234):
    """Per-rank engine settings: (first_game_id, seed)."""
    return rank * GAME_ID_STRIDE, base_seed + rank


def allgather_records(records, device=None):
    """All-gather a 1-D numpy structured array (engine example records) across the default process
    group.  Returns the concatenation in rank order (identical on every rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    raw = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    n_local = torch.tensor([raw.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.uint8, device=dev)
    if raw.size:
        pad[:raw.size] = torch.from_numpy(raw.copy()).to(dev)
    gathered = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(gathered, pad)
    parts = [g[:n].cpu().numpy() for g, n in zip(gathered, sizes)]
    return np.concatenate(parts).view(records.dtype)
