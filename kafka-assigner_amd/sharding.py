"""Multi-GPU layer: scenarios are independent, so a batch is cut into contiguous scenario ranges,
one per rank (one process per GPU), solved with no communication at all, and the ONE data-path
collective is an all-gather of the fixed 32-byte per-scenario result records
(kas_scenario_result: status, failing topic / partition, movement counts, digest).

backend "nccl" is RCCL over xGMI on the GPU box; the same code runs over "gloo" on CPU tensors
(tests/test_sharding_gloo.py, world_size 2).  Gathering full assignments is deliberately not part
of the path: 64k scenarios x 1.2 MB would be 78.6 GB, the records are 2 MiB.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from . import abi

RECORD_BYTES = abi.SCENARIO_RESULT_DTYPE.itemsize      # 32


def shard_range(n_scenarios: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of scenario indices owned by `rank` (sizes differ by <= 1)."""
    assert 0 <= rank < world
    base, extra = divmod(n_scenarios, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_scenarios: int, world: int):
    return [shard_range(n_scenarios, r, world)[1] - shard_range(n_scenarios, r, world)[0] for r in range(world)]


def gather_records(local_records, n_scenarios: int, group=None, out=None):
    """All-gather the per-scenario records of every rank's shard.

    local_records: uint8 torch tensor [shard * 32] on this rank's device (HBM for nccl, CPU for
    gloo) holding kas_scenario_result records of shard_range(n_scenarios, rank, world).
    Returns a uint8 tensor [n_scenarios * 32] in global scenario order on the same device
    (`out` when given and the shards are even, so a step allocates nothing).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_scenarios, world)
    assert local_records.numel() == sizes[rank] * RECORD_BYTES, "record buffer does not match the shard"
    if len(set(sizes)) == 1:
        if out is None:
            out = torch.empty(n_scenarios * RECORD_BYTES, dtype=torch.uint8, device=local_records.device)
        dist.all_gather_into_tensor(out, local_records, group=group)
        return out
    # ragged shards: pad to the largest, gather, drop the padding
    m = max(sizes) * RECORD_BYTES
    padded = torch.zeros(m, dtype=torch.uint8, device=local_records.device)
    padded[:local_records.numel()] = local_records
    out = torch.empty(world * m, dtype=torch.uint8, device=local_records.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * m:r * m + sizes[r] * RECORD_BYTES] for r in range(world)])


def records_view(buf) -> np.ndarray:
    """uint8 torch tensor of records -> numpy structured array (host copy)."""
    return buf.cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
