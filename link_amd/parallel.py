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

import glob
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

__all__ = ["shard_frames", "gather_frame_rows", "gpu_numa_node", "pin_rank_to_gpu_numa"]


def _parse_cpulist(txt: str) -> List[int]:
    """'0-63,128-191' -> [0, ..., 63, 128, ..., 191] (the kernel's cpulist format)."""
    cpus: List[int] = []
    for part in txt.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(device_index: int, sysfs: str = "/sys") -> Optional[int]:
    """NUMA node of the GPU torch calls `cuda:device_index`, from sysfs (`<pci device>/numa_node`); None when unknown (-1, a
    container without sysfs, a PCI address that is not there).  The PCI address comes from torch's device properties, so the
    answer follows HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES remapping (a bare /sys/class/drm/card<i> index would not)."""
    try:
        props = torch.cuda.get_device_properties(device_index)
        addr = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        path = os.path.join(sysfs, "bus", "pci", "devices", addr, "numa_node")
        node = int(open(path).read().strip())
        return node if node >= 0 else None
    except Exception:  # noqa: BLE001
        return None


def pin_rank_to_gpu_numa(device_index: int, local_rank: int = 0, local_world: int = 1, sysfs: str = "/sys") -> Dict:
    """Pin THIS process (the thread that issues the rank's launches included) to CPUs of its GPU's NUMA node.

    A rank's issuing thread spends 12-20 us of host time per 35 us frame; on a two-socket host an unpinned rank may run -- and
    allocate its launch queues -- on the far socket from its GPU.  The reference leaves placement to its launcher
    (detection/det3d/torchie/trainer/utils.py:99-112 only picks the device); here every rank narrows its affinity mask to the
    CPUs of `numa_node` of its GPU's PCI device that its current mask allows, and when several ranks share a node each takes
    its own slice of those CPUs (by local rank) so that eight issuing threads never contend for one core.  Nothing is pinned
    when the node is unknown or the slice would be empty -- the returned record says which:
    {"pinned": bool, "numa_node": int | None, "cpus": n, "first_cpu": c, "reason": str}."""
    rec: Dict = {"pinned": False, "numa_node": None, "cpus": 0, "first_cpu": None, "reason": ""}
    if not hasattr(os, "sched_setaffinity"):
        rec["reason"] = "no sched_setaffinity on this platform"
        return rec
    allowed = sorted(os.sched_getaffinity(0))
    rec["cpus"], rec["first_cpu"] = len(allowed), allowed[0] if allowed else None
    node = gpu_numa_node(device_index, sysfs)
    rec["numa_node"] = node
    if node is None:
        rec["reason"] = "NUMA node of the GPU unknown (sysfs numa_node missing or -1)"
        return rec
    try:
        node_cpus = set(_parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read()))
    except Exception as e:  # noqa: BLE001
        rec["reason"] = f"node{node}/cpulist unreadable: {e!r}"[:120]
        return rec
    mine = [c for c in allowed if c in node_cpus]
    if not mine:
        rec["reason"] = f"no allowed CPU on node {node}"
        return rec
    # ranks that share the node share its CPUs evenly: local ranks whose GPU sits on the same node, in local-rank order
    peers = [r for r in range(max(local_world, 1)) if gpu_numa_node(r, sysfs) == node] if local_world > 1 else [local_rank]
    if local_rank in peers and len(peers) > 1 and len(mine) >= len(peers):
        per = len(mine) // len(peers)
        k = peers.index(local_rank)
        mine = mine[k * per:(k + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError as e:
        rec["reason"] = f"sched_setaffinity refused: {e!r}"[:120]
        return rec
    rec.update(pinned=True, cpus=len(mine), first_cpu=mine[0], reason=f"{len(mine)} CPUs of NUMA node {node}")
    return rec


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


def streams_on_own_queues(n: int, device, max_extra: int = 9):
    """`n` torch streams that sit on pairwise different hardware queues (include/link_amd.h: link_streams_share_queue -- the runtime
    multiplexes streams onto a few queues, and frames kept in flight on two streams of one queue run one after the other: one of six
    triples of consecutive torch streams did, tools/stream_placement.py).  Streams are drawn from torch's pool until `n` pass the test
    pair by pair (at most `max_extra` rejects; then the best effort so far).  Returns (streams, record) -- record: the delays measured
    between the kept streams (us; ~5 = own queues) and how many candidates were passed over.  Synchronises the streams it tests."""
    import ctypes
    from . import _lib as L
    lib = L.lib()
    kept, rejected, delays = [], 0, []
    while len(kept) < n:
        s = torch.cuda.Stream(device=device)
        worst = 0.0
        for k in kept:
            d = ctypes.c_double(0.0)
            L.check(lib.link_streams_share_queue(k.cuda_stream, s.cuda_stream, ctypes.byref(d)), "link_streams_share_queue")
            worst = max(worst, float(d.value))
        if worst > 75.0 and rejected < max_extra:
            rejected += 1
            continue
        kept.append(s)
        delays.append(round(worst, 1))
    return kept, {"max_delay_us_to_earlier_kept_stream": delays, "candidates_passed_over": rejected,
                  "note": "a kernel on the stream behind a 150 us kernel + event record on an earlier kept stream: ~5 us = hardware queues of their own"}

