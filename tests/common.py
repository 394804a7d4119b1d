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


# ---- round 5: the other two synthetic worlds (msf_loam_amd/worlds.py) -------------------------------------------------

@functools.lru_cache(maxsize=None)
def other_world(kind):
    """(world, map_corner, map_surf) of the `outdoor` / `corridor` world (`room`: small_world())."""
    if kind == "room":
        return small_world()
    w = synth.World(kind=kind)
    mc, ms = synth.make_map(w)
    return w, mc, ms


@functools.lru_cache(maxsize=None)
def other_scans(kind, n, seed=synth.SEED + 2):
    """n raw sensor clouds of that world + true poses + perturbed guesses."""
    if kind == "room":
        return scans(n)
    w, _, _ = other_world(kind)
    poses = synth.world_poses(w, n, seed)
    rng = np.random.default_rng(seed + 1000)
    out = []
    for i in range(n):
        pts, ring = synth.make_scan(w, poses[i], seed + 10 + i)
        out.append((pts, ring, poses[i], synth.perturb_pose(poses[i], rng)))
    return out


def world_drive(kind, n):
    """A drive through the world for the sequential replay (~0.4 m and a degree or two per scan): along the x = 0 street of the
    outdoor world (relief followed, gentle weaving), down the corridor's axis — examples/replay_synthetic.py:world_drive."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import replay_synthetic as rp
    w, _, _ = other_world(kind)
    return rp.world_drive(w, kind, n)
