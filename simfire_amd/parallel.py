"""Multi-GPU layout of the path: environments are independent, so they are sharded across the
ranks of one node (one process per GPU) with NO data-path collective; the only exchange is one
all-gather of the per-environment result block (int32 [n_envs_local, 8]: running, elapsed_steps,
cell counts per BurnStatus) per rollout - a few KB over RCCL/xGMI (backend ``nccl`` on ROCm) or
gloo on CPU.  SURVEY.md section 8e."""
from typing import Tuple

import numpy as np


def shard_envs(n_envs_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, stop) of the environment axis owned by ``rank``."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    base, extra = divmod(int(n_envs_total), world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_results(local_block, n_envs_total: int = None):
    """All-gather the per-environment result blocks of all ranks, in rank (= environment) order.

    ``local_block``: torch int32 tensor [n_local, 8] on the device of the process group's backend
    (cuda for nccl, cpu for gloo).  Ranks may own different numbers of environments (ragged
    shards are padded to the largest shard for the collective and trimmed afterwards)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return local_block
    world = dist.get_world_size()
    n_local = torch.tensor([local_block.shape[0]], dtype=torch.int64, device=local_block.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    padded = torch.zeros((n_max, local_block.shape[1]), dtype=local_block.dtype, device=local_block.device)
    padded[: local_block.shape[0]] = local_block
    out = torch.empty((world * n_max, local_block.shape[1]), dtype=local_block.dtype, device=local_block.device)
    dist.all_gather_into_tensor(out, padded)
    parts = [out[r * n_max: r * n_max + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0)
