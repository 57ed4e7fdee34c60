"""link_amd/parallel.py -- multi-GPU host logic for the LinK hot path.

The path shards by independent frames (batch items never interact: the batch index is part of
every block key, /root/reference/segmentation/core/models/utils.py:45), so there is NO data-path
collective: one process per GPU, each rank runs whole frames.  The only communication is the
"trivial result gather" of per-frame summaries after the compute -- the same two-phase pattern the
reference uses for its prediction gather (sizes first, then payload padded to the max;
/root/reference/detection/det3d/torchie/trainer/utils.py:114-155) minus the pickling.
Backend: torch.distributed "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist

__all__ = ["shard_frames", "gather_frame_rows"]


def shard_frames(n_frames: int, world_size: int, rank: int) -> List[int]:
    """Frame ids processed by `rank`: round-robin, so every round keeps all GPUs busy and rank r owns
    frames r, r+W, r+2W, ... (BASELINE.json configs[3]: 8 frames over 1/2/4/8 GPUs = 8/4/2/1 rounds)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, n_frames, world_size))


def gather_frame_rows(rows: torch.Tensor) -> torch.Tensor:
    """All-gather variable-length per-frame summary rows [k_r, F] (float64) -> [sum_r k_r, F] on every
    rank, ordered by rank.  Two collectives: sizes, then padded payload."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return rows
    world = dist.get_world_size()
    k = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    sizes = [torch.zeros_like(k) for _ in range(world)]
    dist.all_gather(sizes, k)
    sizes = [int(s.item()) for s in sizes]
    kmax = max(sizes)
    pad = torch.zeros((kmax, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)
