"""Multi-GPU = independent image chains: rank r of W owns images[r::W]; UNet
weights are replicated; there is NO collective on the data path.  The only cross-rank traffic is
the end-of-run bookkeeping below (a MAX over per-rank wall times and a gather of per-image
results), which works on any torch.distributed backend (RCCL on GPUs, gloo in the CPU tests)."""
from typing import List, Sequence

import torch


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    return list(range(n_items))[rank::world]


def max_over_ranks(value: float, device=None) -> float:
    """MAX-reduce a scalar over all ranks (identity when torch.distributed is not initialised)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if dist.get_backend() == "gloo":
        device = None                    # gloo reduces host tensors
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_per_image(values: Sequence[float], n_items: int, device=None) -> List[float]:
    """Every rank passes the values of ITS images (in shard order); returns the full list in image order."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(values)
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() == "gloo":
        device = None
    per = (n_items + world - 1) // world
    mine = torch.full((per,), float("nan"), dtype=torch.float64, device=device)
    mine[:len(values)] = torch.tensor(list(values), dtype=torch.float64, device=device)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    out = [float("nan")] * n_items
    for r in range(world):
        for k, idx in enumerate(shard_indices(n_items, r, world)):
            out[idx] = float(allv[r][k])
    return out
