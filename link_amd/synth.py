"""link_amd/synth.py -- the synthetic frame generators of SURVEY.md section 8d (pure numpy, seeded).

    s_uniform(n, grid, seed)     cfg1 / cfg2: n unique voxels uniform in grid^3 (torch generator, as specified)
    s_kitti(seed)                cfg3 / cfg4: 64-beam scan, 0.05 m voxels   -> ~90-125k voxels
    s_nusc(seed)                 cfg5: 32-beam, 10 sweeps, 0.075/0.075/0.2 m grid 1440x1440x40 -> ~150-175k voxels

The LiDAR frames come from one ray caster: rays (elevation x azimuth) from a sensor at height h, nearest hit
against the ground plane z = 0, 40 seeded axis-aligned boxes (footprint U[1,8] m, height U[1,4] m, centres
U[-40,40] m, boxes containing the sensor dropped) and 4 walls at +-45 m (height 6 m), range limit 50 m,
N(0, 1 cm) jitter on the hit point.  Voxelisation follows the reference's data pipeline:
  S-kitti   pc = round(xyz / 0.05); pc -= pc.min(0)                   segmentation/core/datasets/semantic_kitti.py:219-220
  S-nusc    crop [-54,54]^2 x [-5,3] m, floor((p - min) / (0.075, 0.075, 0.2)), per-voxel mean of <= 10 points
            over (x, y, z, intensity, dt)                              detection/configs/nusc/voxelnet/
            nusc_centerpoint_voxelnet_0075voxel_fix_bn_z_elkv3.py:141-146, det3d/models/readers/voxel_encoder.py:17-24
What matters for the LinK kernels is the block occupancy these scenes produce (N/M of 3-16, surfaces, empty
space): it decides which kernel regime runs.  Nothing here touches the GPU.
"""
from __future__ import annotations

import numpy as np


def s_uniform(n: int, grid: int = 256, seed: int = 0, batch: int = 0):
    """SURVEY 8d S-uniform: torch.randperm(grid^3)[:n] -> (x, y, z, batch) int32 [n, 4]."""
    import torch
    g = torch.Generator().manual_seed(seed)
    lin = torch.randperm(grid ** 3, generator=g)[:n]
    x, y, z = lin % grid, (lin // grid) % grid, lin // (grid * grid)
    return torch.stack([x, y, z, torch.full_like(x, batch)], 1).int()


def _scene(rng):
    """40 boxes (lo, hi corners; the ones containing the sensor dropped) + the 4 walls as thin boxes."""
    c = rng.uniform(-40.0, 40.0, (40, 2))
    fp = rng.uniform(1.0, 8.0, (40, 2))
    hgt = rng.uniform(1.0, 4.0, 40)
    lo = np.concatenate([c - fp / 2, np.zeros((40, 1))], 1)
    hi = np.concatenate([c + fp / 2, hgt[:, None]], 1)
    keep = ~((lo[:, 0] < 0) & (hi[:, 0] > 0) & (lo[:, 1] < 0) & (hi[:, 1] > 0))
    lo, hi = lo[keep], hi[keep]
    w, t, h = 45.0, 0.2, 6.0
    walls_lo = np.array([[w, -w, 0], [-w - t, -w, 0], [-w, w, 0], [-w, -w - t, 0]], float)
    walls_hi = np.array([[w + t, w, h], [-w, w, h], [w, w + t, h], [w, -w, h]], float)
    return np.concatenate([lo, walls_lo]), np.concatenate([hi, walls_hi])


def _raycast(origin, d, lo, hi, max_range):
    """Nearest hit distance of rays origin + t d against the ground z = 0 and the boxes (slab test)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(d[:, 2] < -1e-9, -origin[2] / d[:, 2], np.inf)
        inv = 1.0 / d
        for b in range(lo.shape[0]):
            t0 = (lo[b] - origin) * inv
            t1 = (hi[b] - origin) * inv
            tn = np.nanmax(np.minimum(t0, t1), 1)
            tf = np.nanmin(np.maximum(t0, t1), 1)
            hit = (tn <= tf) & (tf > 0) & (tn > 0)
            t = np.where(hit & (tn < t), tn, t)
    return np.where(t <= max_range, t, np.nan)


def _scan(rng, n_el, el_lo, el_hi, n_az, height, lo, hi, az_phase=0.0, shift=0.0, max_range=50.0):
    el = np.deg2rad(np.linspace(el_lo, el_hi, n_el))
    az = az_phase + 2 * np.pi * np.arange(n_az) / n_az
    A, E = np.meshgrid(az, el)
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    origin = np.array([shift, 0.0, height])
    t = _raycast(origin, d, lo, hi, max_range)
    ok = ~np.isnan(t)
    p = origin + d[ok] * t[ok, None]
    return p + rng.normal(0.0, 0.01, p.shape)


def s_kitti(seed: int = 0, n_az: int = 4608, voxel: float = 0.05, return_points: bool = False):
    """SURVEY 8d S-kitti.  Returns (coords int32 [N,4] with batch 0, feats float32 [N,4] = first point's
    (x, y, z, intensity) per voxel, as sparse_quantize(return_index=True) selects them)."""
    rng = np.random.default_rng(seed)
    lo, hi = _scene(rng)
    pts = _scan(rng, 64, -24.8, 2.0, n_az, 1.73, lo, hi)
    inten = rng.uniform(0.0, 1.0, (pts.shape[0], 1))
    pc = np.round(pts / voxel).astype(np.int32)
    pc -= pc.min(0, keepdims=True)
    _, idx = np.unique(pc, axis=0, return_index=True)
    idx = np.sort(idx)
    coords = np.zeros((idx.shape[0], 4), np.int32)
    coords[:, :3] = pc[idx]
    feats = np.concatenate([pts[idx], inten[idx]], 1).astype(np.float32)
    return (coords, feats, pts) if return_points else (coords, feats)


def s_nusc(seed: int = 0, n_az: int = 2180, sweeps: int = 10, max_points: int = 10):
    """SURVEY 8d S-nusc.  Returns (coords int32 [N,4] in (x, y, z, batch) order on the 1440 x 1440 x 40 grid,
    feats float32 [N,5] = per-voxel mean of at most `max_points` points over (x, y, z, intensity, dt))."""
    rng = np.random.default_rng(seed)
    lo, hi = _scene(rng)
    allp = []
    for k in range(sweeps):
        p = _scan(rng, 32, -30.0, 10.0, n_az, 1.84, lo, hi, az_phase=rng.uniform(0, 2 * np.pi),
                  shift=k * rng.uniform(0.0, 0.5))
        f = np.concatenate([p, rng.uniform(0, 1, (p.shape[0], 1)), np.full((p.shape[0], 1), 0.05 * k)], 1)
        allp.append(f)
    f = np.concatenate(allp)
    pmin = np.array([-54.0, -54.0, -5.0])
    pmax = np.array([54.0, 54.0, 3.0])
    ok = np.all((f[:, :3] >= pmin) & (f[:, :3] < pmax), 1)
    f = f[ok]
    vs = np.array([0.075, 0.075, 0.2])
    vox = np.floor((f[:, :3] - pmin) / vs).astype(np.int64)
    lin = (vox[:, 0] * 1440 + vox[:, 1]) * 40 + vox[:, 2]
    order = np.argsort(lin, kind="stable")
    lin, f, vox = lin[order], f[order], vox[order]
    uniq, start, counts = np.unique(lin, return_index=True, return_counts=True)
    # mean of the first <= max_points points of every voxel (VoxelFeatureExtractorV3 over the voxeliser's cap)
    rank = np.arange(lin.shape[0]) - np.repeat(start, counts)
    use = rank < max_points
    seg = np.repeat(np.arange(uniq.shape[0]), counts)[use]
    sums = np.zeros((uniq.shape[0], 5))
    np.add.at(sums, seg, f[use])
    feats = (sums / np.minimum(counts, max_points)[:, None]).astype(np.float32)
    coords = np.zeros((uniq.shape[0], 4), np.int32)
    coords[:, :3] = vox[start]
    return coords, feats


def block_stats(coords: np.ndarray, s: int):
    """(N, M, N/M, max voxels per block) for block edge s (the quantity that picks kernel regimes)."""
    blk = np.floor_divide(coords[:, :3], s)
    key = np.concatenate([blk, coords[:, 3:4]], 1)
    _, cnt = np.unique(key, axis=0, return_counts=True)
    return coords.shape[0], cnt.shape[0], coords.shape[0] / cnt.shape[0], int(cnt.max())
