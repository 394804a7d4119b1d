"""Generates tests/golden/msfl_golden_v1.npz.

PROVENANCE: the reference (kekeliu-whu/MSF_LOAM) holds no golden vectors or fixtures for this path
and cannot be built here, so these vectors come from the repo's own CPU oracle
(oracle/msfl_oracle.c, itself pinned against numpy/scipy in tests/test_oracle_*.py) on seeded
synthetic input.  They pin the oracle against regressions and give the GPU tests a fixed target
that does not depend on running the oracle.   Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from msf_loam_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    orc.build()
    world = synth.World(ground_half=25.0)
    mc, ms = synth.make_map(world, corner_spacing=0.2, surf_spacing=0.4)
    keep = (np.abs(ms[:, 0]) < 25) & (np.abs(ms[:, 1]) < 22)
    ms = ms[keep]
    poses = synth.random_poses(2, synth.SEED + 11)
    rng = np.random.default_rng(synth.SEED + 12)
    out = dict(map_corner=mc, map_surf=ms)
    for k in range(2):
        pts, ring = synth.make_scan(world, poses[k], synth.SEED + 20 + k, n_az=600)
        guess = synth.perturb_pose(poses[k], rng)
        f = orc.extract_features(pts, ring)
        corner = orc.voxel_grid(f["full"][f["less_sharp"]], 0.2)
        surf = orc.voxel_grid(f["full"][f["less_flat"]], 0.4)
        corr = orc.associate_scan2map(mc, ms, corner, surf, guess)
        p1, summ = orc.ceres_solve(corr, guess)
        rc, pose, info = orc.match_scan2map(mc, ms, corner, surf, guess)
        out.update({
            f"s{k}_pts": pts, f"s{k}_ring": ring, f"s{k}_truth": poses[k], f"s{k}_guess": guess,
            f"s{k}_full": f["full"], f"s{k}_full_ring": f["ring"], f"s{k}_curvature": f["curvature"], f"s{k}_label": f["label"],
            f"s{k}_sharp": f["sharp"], f"s{k}_less_sharp": f["less_sharp"], f"s{k}_flat": f["flat"], f"s{k}_less_flat": f["less_flat"],
            f"s{k}_corner_ds": corner, f"s{k}_surf_ds": surf,
            f"s{k}_rec_kind": corr["kind"], f"s{k}_rec_C": corr["C"], f"s{k}_rec_N": corr["N"],
            f"s{k}_solve1_pose": p1, f"s{k}_solve1_iters": np.array([summ.iterations, summ.successful_steps, summ.termination]),
            f"s{k}_solve1_costs": np.array([summ.initial_cost, summ.final_cost] + list(summ.trace_cost)[:summ.iterations]),
            f"s{k}_solve1_radius": np.array(list(summ.trace_radius)[:summ.iterations]),
            f"s{k}_pose": pose, f"s{k}_info": np.array(list(info.n_edge) + list(info.n_plane) + list(info.lm_iterations) + list(info.lm_successful)),
        })
    # stage B: scan 1 registered against scan 0's features, from identity
    f0 = {k: out[f"s0_{k}"] for k in ("full", "full_ring", "sharp", "less_sharp", "flat", "less_flat")}
    f1 = {k: out[f"s1_{k}"] for k in ("full", "full_ring", "sharp", "less_sharp", "flat", "less_flat")}
    # use a nearby second view: re-render scan 1 from a pose close to scan 0
    near = synth.perturb_pose(poses[0], rng, 0.2, 1.5)
    pts, ring = synth.make_scan(world, near, synth.SEED + 30, n_az=600)
    fb = orc.extract_features(pts, ring)
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    rc, rel, info = orc.match_scan2scan(f0["full"][f0["less_sharp"]], f0["full_ring"][f0["less_sharp"]],
                                        f0["full"][f0["less_flat"]], f0["full_ring"][f0["less_flat"]],
                                        fb["full"][fb["sharp"]], fb["full"][fb["flat"]], ident)
    out.update(odo_pts=pts, odo_ring=ring, odo_rc=np.array([rc]), odo_pose=rel,
               odo_info=np.array(list(info.n_edge) + list(info.n_plane) + list(info.lm_iterations)))
    path = os.path.join(ROOT, "tests", "golden", "msfl_golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", "map", len(mc), len(ms), "scan pts", len(out["s0_pts"]),
          "F", len(out["s0_corner_ds"]), len(out["s0_surf_ds"]), "odo rc", rc, list(info.n_edge), list(info.n_plane))


if __name__ == "__main__":
    main()
