"""initial_voxelize / point_to_voxel / voxel_to_point on a 120k-point cloud: the dense-grid forms against the hash-based
reference algorithm on the op kernels (forced through the GridTooLarge path).   python tools/pvbench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import link_amd as la
from link_amd import pointvoxel as PV
from link_amd.index import GridTooLarge

g = torch.Generator().manual_seed(11)
P = 120000
pts = torch.cat([(torch.rand(P, 3, generator=g) - 0.4) * torch.tensor([90.0, 70.0, 12.0]), torch.zeros(P, 1)], 1).cuda()
feats = torch.randn(P, 9, generator=g).cuda()


def step():
    z = la.PointTensor(feats, pts.clone())
    st = la.initial_voxelize(z, 1.0, 0.5)
    v = la.point_to_voxel(st, z)
    return la.voxel_to_point(la.SparseTensor(st.F, st.C, 1), z)


def timeit(k=30):
    for _ in range(5):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k


with torch.no_grad():
    t_native = timeit()
    real = (PV.BlockIndex, PV.foreign_neighbor_map)
    def boom(*a, **k):
        raise GridTooLarge("forced")
    PV.BlockIndex = PV.foreign_neighbor_map = boom
    t_hash = timeit()
    PV.BlockIndex, PV.foreign_neighbor_map = real
print(f"voxelize + point_to_voxel + voxel_to_point, {P} points: dense-grid forms {t_native:.3f} ms, hash-based op chain {t_hash:.3f} ms")
