"""SURVEY.md section 8d generators (link_amd/synth.py): sizes and block occupancy in the specified ranges,
reproducible per seed, unique voxels.  CPU only."""
import numpy as np

from link_amd import synth


def test_s_kitti_shape_and_occupancy():
    coords, feats = synth.s_kitti(0)
    assert coords.dtype == np.int32 and coords.shape[1] == 4 and feats.shape == (coords.shape[0], 4)
    n = coords.shape[0]
    assert 85_000 <= n <= 135_000                                  # spec: 89k-124k over seeds 0..7
    assert np.unique(coords, axis=0).shape[0] == n and coords.min() == 0 and (coords[:, 3] == 0).all()
    sizes = [np.unique(coords[:, :3] // st, axis=0).shape[0] for st in (2, 4, 8, 16)]
    assert 40_000 <= sizes[0] <= 75_000 and 16_000 <= sizes[1] <= 34_000
    assert 6_000 <= sizes[2] <= 13_000 and 2_000 <= sizes[3] <= 4_500
    n_, m, ratio, mx = synth.block_stats(coords, 7)                # (3x7)^3 on stride-1 voxels: surfaces, N/M >> 2
    assert 5.0 <= ratio <= 12.0 and mx <= 343
    c2, f2 = synth.s_kitti(0)
    assert np.array_equal(coords, c2) and np.array_equal(feats, f2)
    c3, _ = synth.s_kitti(1)
    assert c3.shape != coords.shape or not np.array_equal(c3, coords)


def test_s_nusc_shape_and_occupancy():
    coords, feats = synth.s_nusc(0)
    n = coords.shape[0]
    assert 130_000 <= n <= 185_000 and feats.shape == (n, 5)       # spec: 152k-175k
    assert coords[:, 0].max() < 1440 and coords[:, 1].max() < 1440 and coords[:, 2].max() < 40 and coords.min() >= 0
    assert np.unique(coords, axis=0).shape[0] == n
    _, m, ratio, mx = synth.block_stats(coords, 7)
    assert ratio > 8.0 and mx <= 343
    assert np.isfinite(feats).all() and 0.0 <= feats[:, 4].min() and feats[:, 4].max() <= 0.45 + 1e-6


def test_s_uniform_matches_bench_generator():
    import bench
    a = synth.s_uniform(5000, 64, 3).numpy()
    b = bench.s_uniform(5000, 64, 3).numpy()
    assert np.array_equal(a, b) and np.unique(a, axis=0).shape[0] == 5000
