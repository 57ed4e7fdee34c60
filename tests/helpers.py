"""Shared helpers for the test-suite (fixture loading, seeded generators, error metrics)."""
import glob
import json
import os

import functools

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    d = {k: z[k] for k in z.files if k != "meta"}
    d["meta"] = json.loads(str(z["meta"]))
    return d


def golden_files(pattern):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, pattern)))


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the 'within 1e-4 rel on fp32 features' metric of BASELINE.json."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    denom = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max() / denom)


@functools.lru_cache(maxsize=12)
def _s_uniform_perm(grid, seed):
    """The generator's permutation of the grid's cells (what costs the time: 16.7 M cells for grid 256); cached per (grid, seed) --
    many tests draw the same seeds."""
    import torch
    return torch.randperm(grid ** 3, generator=torch.Generator().manual_seed(seed))


def s_uniform(n, grid=256, seed=0, batch=0):
    """SURVEY.md section 8d S-uniform generator (unique voxels, uniform in grid^3)."""
    import torch
    lin = _s_uniform_perm(grid, seed)[:n]
    x, y, z = lin % grid, (lin // grid) % grid, lin // (grid * grid)
    return torch.stack([x, y, z, torch.full_like(x, batch)], 1).int()


def lidar_like(n_target=60000, seed=0, voxel=0.05, stride=1):
    """Cheap LiDAR-shaped frame: ground plane + a few boxes, ring pattern; unique int32 voxels."""
    rng = np.random.default_rng(seed)
    n_az, n_el = 2048, 64
    az = rng.uniform(0, 2 * np.pi, n_az)
    el = np.deg2rad(np.linspace(-24.8, 2.0, n_el))
    A, E = np.meshgrid(az, el)
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    h = 1.73
    t = np.where(d[:, 2] < -1e-3, -h / np.minimum(d[:, 2], -1e-3), 50.0)
    t = np.minimum(t, 50.0)
    # a few walls
    for _ in range(12):
        c = rng.uniform(-35, 35, 2); w = rng.uniform(2, 8)
        for ax in (0, 1):
            tt = (c[ax] - 0) / np.where(np.abs(d[:, ax]) > 1e-6, d[:, ax], 1e-6)
            p = d * tt[:, None]
            ok = (tt > 1) & (np.abs(p[:, 1 - ax] - c[1 - ax]) < w) & (p[:, 2] + h < 3.5) & (p[:, 2] + h > 0)
            t = np.where(ok & (tt < t), tt, t)
    pts = d * t[:, None] + np.array([0, 0, h]) + rng.normal(0, 0.01, (d.shape[0], 3))
    pc = np.round(pts / voxel).astype(np.int64)
    pc -= pc.min(0)
    pc = (pc // stride) * stride
    pc = np.unique(pc, axis=0)
    rng.shuffle(pc)
    pc = pc[:n_target]
    out = np.zeros((pc.shape[0], 4), np.int32)
    out[:, :3] = pc
    return out
