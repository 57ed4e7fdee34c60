"""world_size-2 gloo test of the N>1 path's host logic (frame sharding + the two-phase result gather).
The data path itself has no collective (independent frames), so this is everything that differs
between 1 and N GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from link_amd.parallel import gather_frame_rows, shard_frames
    mine = shard_frames(n_frames, world, rank)
    # per-frame summary: (frame id, voxels, checksum) -- what bench.py gathers
    rows = torch.tensor([[f, 1000 + f, 0.5 * f] for f in mine], dtype=torch.float64).reshape(-1, 3)
    allr = gather_frame_rows(rows)
    dist.barrier()
    if rank == 0:
        q.put(allr.tolist())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [8, 5, 1])
def test_gloo_world2_shard_and_gather(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    rows = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames = sorted(int(r[0]) for r in rows)
    assert frames == list(range(n_frames))                      # every frame exactly once
    for f, vox, chk in rows:
        assert vox == 1000 + f and chk == 0.5 * f


def test_shard_frames_properties():
    from link_amd.parallel import shard_frames
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            part = shard_frames(8, world, r)
            assert len(part) == 8 // world
            seen += part
        assert sorted(seen) == list(range(8))
    with pytest.raises(ValueError):
        shard_frames(8, 2, 2)
