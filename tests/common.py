"""Shared synthetic cases for the tests (seeded; SURVEY.md §8d recipe at test scale)."""
import functools

import numpy as np

from msf_loam_amd import synth


@functools.lru_cache(maxsize=None)
def small_world(target=50000):
    w = synth.World(ground_half=synth.ground_half_for_target(target))
    mc, ms = synth.make_map(w)
    return w, mc, ms


@functools.lru_cache(maxsize=None)
def scans(n, target=50000, seed=synth.SEED + 2):
    """n raw sensor clouds + true poses + perturbed guesses."""
    w, _, _ = small_world(target)
    poses = synth.random_poses(n, seed)
    rng = np.random.default_rng(seed + 1000)
    out = []
    for i in range(n):
        pts, ring = synth.make_scan(w, poses[i], seed + 10 + i)
        out.append((pts, ring, poses[i], synth.perturb_pose(poses[i], rng)))
    return out


def features_from_oracle(orc, pts, ring):
    """Feature clouds as the mapping thread receives them: oracle extraction + PCL-style voxel grid
    (0.2 m corner / 0.4 m surf, laser_mapping.cc:264-270)."""
    f = orc.extract_features(pts, ring)
    corner = orc.voxel_grid(f["full"][f["less_sharp"]], 0.2)
    surf = orc.voxel_grid(f["full"][f["less_flat"]], 0.4)
    return f, corner, surf
