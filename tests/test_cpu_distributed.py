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


def _fake_sysfs(tmp_path, nodes):
    """sysfs tree with one PCI device per GPU: nodes = {gpu index: numa node}, 8 CPUs per node."""
    for gpu, node in nodes.items():
        d = tmp_path / "bus" / "pci" / "devices" / f"0000:{gpu + 1:02x}:00.0"
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    for node in set(nodes.values()):
        if node < 0:
            continue
        d = tmp_path / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(f"{8 * node}-{8 * node + 7}\n")
    return str(tmp_path)


def test_pin_rank_to_gpu_numa(tmp_path, monkeypatch):
    """Round 6: every rank narrows its CPU affinity to the NUMA node of ITS GPU (PCI address from torch's
    device properties -> sysfs numa_node -> node cpulist), ranks that share a node take disjoint slices; unknown node = no pinning."""
    import types
    import torch
    from link_amd import parallel as P
    assert P._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    sysfs = _fake_sysfs(tmp_path, {0: 0, 1: 0, 2: 1, 3: -1})
    monkeypatch.setattr(torch.cuda, "get_device_properties",
                        lambda i: types.SimpleNamespace(pci_domain_id=0, pci_bus_id=i + 1, pci_device_id=0))
    state = {"mask": set(range(16))}
    monkeypatch.setattr(P.os, "sched_getaffinity", lambda pid: set(state["mask"]), raising=False)
    monkeypatch.setattr(P.os, "sched_setaffinity", lambda pid, cpus: state.__setitem__("mask", set(cpus)), raising=False)
    assert P.gpu_numa_node(2, sysfs) == 1 and P.gpu_numa_node(3, sysfs) is None and P.gpu_numa_node(7, sysfs) is None
    rec = P.pin_rank_to_gpu_numa(2, local_rank=2, local_world=4, sysfs=sysfs)
    assert rec["pinned"] and rec["numa_node"] == 1 and state["mask"] == set(range(8, 16)) and rec["cpus"] == 8
    state["mask"] = set(range(16))
    rec0 = P.pin_rank_to_gpu_numa(0, local_rank=0, local_world=4, sysfs=sysfs)
    m0 = set(state["mask"])
    state["mask"] = set(range(16))
    rec1 = P.pin_rank_to_gpu_numa(1, local_rank=1, local_world=4, sysfs=sysfs)
    m1 = set(state["mask"])
    assert rec0["pinned"] and rec1["pinned"] and m0 == {0, 1, 2, 3} and m1 == {4, 5, 6, 7}      # two ranks on node 0: disjoint halves
    state["mask"] = set(range(16))
    rec3 = P.pin_rank_to_gpu_numa(3, local_rank=3, local_world=4, sysfs=sysfs)
    assert not rec3["pinned"] and rec3["numa_node"] is None and state["mask"] == set(range(16)) and "unknown" in rec3["reason"]
    state["mask"] = {12, 13}                                                                     # a cpuset that excludes the GPU's node
    rec = P.pin_rank_to_gpu_numa(0, local_rank=0, local_world=1, sysfs=sysfs)
    assert not rec["pinned"] and state["mask"] == {12, 13} and "no allowed CPU" in rec["reason"]
