"""VERDICT r01 'Next round' 5a: the oracle's Ceres restatement against an INDEPENDENT numpy restatement of the
full trust-region trajectory (tests/ceres_numpy.py, written from the Ceres 1.14 documentation without reading
oracle/msfl_oracle.c): candidate cost, radius, step quality, accept / reject / invalid and the stop reason of
every iteration, on the six solver corner cases and on real correspondences of the 50k-map fixtures."""
import numpy as np
import pytest

from msf_loam_amd import synth
from tests import ceres_numpy as cn
from tests import common
from tests.test_gpu_scan2map import _synthetic_corr

TERMINATION = {0: "max_iterations", 1: "gradient", 2: "parameter", 3: "function", 4: "min_radius", 5: "invalid_steps", 6: "empty"}


def _compare(oracle, corr, guess, what, **overrides):
    """overrides: Solver::Options fields (oracle field name -> value) applied to BOTH implementations."""
    oo = oracle.default_solver_options()

    class NOpt(cn.Options):
        pass
    names = dict(max_consecutive_invalid_steps="max_num_consecutive_invalid_steps")
    for k, v in overrides.items():
        setattr(oo, k, v)
        setattr(NOpt, names.get(k, k), v)
    pose_o, s = oracle.ceres_solve(corr, guess, oo)
    pose_n, t = cn.solve(corr, guess, NOpt)
    assert TERMINATION[s.termination] == t.termination, (what, s.termination, t.termination)
    assert s.iterations == t.iterations and s.successful_steps == t.successful_steps, what
    if t.termination == "empty":
        assert np.array_equal(pose_o, guess) and np.array_equal(pose_n, guess)
        return s
    assert abs(s.initial_cost - t.initial_cost) <= 1e-11 * t.initial_cost + 1e-24, what      # 1e-24: costs that are rounding noise (at the optimum)
    for k in range(t.iterations):
        assert s.trace_accepted[k] == t.accepted[k], (what, k)
        assert abs(s.trace_radius[k] - t.radius[k]) <= 1e-9 * t.radius[k], (what, k, "radius")
        assert abs(s.trace_cost[k] - t.cost[k]) <= 1e-9 * t.cost[k] + 1e-24, (what, k, "candidate cost")
        if t.accepted[k] != -1:
            assert abs(s.trace_step_norm[k] - t.step_norm[k]) <= 1e-7 * max(t.step_norm[k], 1e-12), (what, k, "step norm")
        last_is_stop = k == t.iterations - 1 and t.termination in ("parameter", "function")
        if t.accepted[k] != -1 and not last_is_stop:
            assert abs(s.trace_rel_decrease[k] - t.rel_decrease[k]) <= 1e-6 * max(1.0, abs(t.rel_decrease[k])), (what, k, "step quality")
    assert abs(s.final_cost - t.final_cost) <= 1e-9 * t.final_cost + 1e-24, what
    dt, dr = synth.pose_error(pose_o, pose_n)
    assert dt < 1e-9 and dr < 1e-9, (what, dt, dr)
    return s


@pytest.mark.parametrize("case", ["generic", "at_optimum", "edges_only", "two_normals", "outliers", "many_edges", "no_residuals",
                                  "far_guess"])
def test_trust_region_trajectory_matches_the_independent_restatement(oracle, case):
    rng = np.random.default_rng(5)
    kw = dict(generic=dict(n_plane=400, n_edge=40, noise=0.01),
              at_optimum=dict(n_plane=300, n_edge=30, noise=0.0),
              edges_only=dict(n_plane=0, n_edge=120, noise=0.01),
              two_normals=dict(n_plane=300, n_edge=0, noise=0.005, normals=[[0, 0, 1], [1, 0, 0]]),
              outliers=dict(n_plane=400, n_edge=40, noise=0.01),
              many_edges=dict(n_plane=500, n_edge=2600, noise=0.01),
              no_residuals=dict(n_plane=20, n_edge=5, noise=0.0),
              far_guess=dict(n_plane=300, n_edge=30, noise=0.02))[case]
    corr, truth = _synthetic_corr(rng, **kw)
    if case == "outliers":
        corr["C"][::7] += rng.normal(scale=3.0, size=(len(corr[::7]), 3))
    if case == "many_edges":
        idx = np.nonzero(rng.random(2600) < 0.6)[0]
        corr["kind"][idx] = 0; corr["N"][idx] = 0.0; corr["C"][idx] = 0.0
    if case == "no_residuals":
        corr["kind"][:] = 0
    if case == "at_optimum":
        guess = truth.copy()
    elif case == "far_guess":
        guess = synth.perturb_pose(truth, rng, 3.0, 25.0)          # deep in the Huber region: rejected steps, shrinking radius
    else:
        guess = synth.perturb_pose(truth, rng, 0.2, 2.0)
    s = _compare(oracle, corr, guess, case)
    if case == "far_guess":
        assert s.successful_steps >= 3


@pytest.mark.parametrize("variant", ["reject_chain", "mixed", "tiny_radius", "tight_radius_cap"])
def test_rejected_steps_and_radius_updates(oracle, variant):
    """The branches the default options rarely take: rejected steps (radius / 2, / 4, ... with the LM diagonal reused),
    a radius that starts tiny and grows by the (1/3 floor) rule, the max-radius clamp, a longer iteration budget."""
    rng = np.random.default_rng(11)
    corr, truth = _synthetic_corr(rng, n_plane=300, n_edge=30, noise=0.02)
    corr["C"][::9] += rng.normal(scale=2.0, size=(len(corr[::9]), 3))
    guess = synth.perturb_pose(truth, rng, 1.0, 20.0)
    # step quality on this problem is 1.04 ... 1.96 (the Huber cost falls faster than the quadratic model predicts), so a
    # threshold inside that band rejects steps for real
    ov = dict(reject_chain=dict(min_relative_decrease=1.5, max_num_iterations=8),
              mixed=dict(min_relative_decrease=1.0437, function_tolerance=1e-14, max_num_iterations=12),
              tiny_radius=dict(initial_trust_region_radius=1e-3),
              tight_radius_cap=dict(initial_trust_region_radius=1.0, max_trust_region_radius=2.0))[variant]
    s = _compare(oracle, corr, guess, variant, **ov)
    acc = list(s.trace_accepted[:s.iterations])
    if variant == "reject_chain":
        assert acc == [1, 1, 0, 0, 0, 0, 0, 0], acc               # radius / 2, / 4, / 8 ... with the diagonal reused
        assert s.trace_radius[7] < 1e-3 * s.trace_radius[2]
    if variant == "mixed":
        assert acc[:3] == [1, 1, 1] and 0 in acc[:-1], acc        # accepted steps, then rejections inside the budget
    if variant == "tiny_radius":
        assert s.iterations == 6 and s.trace_radius[1] > s.trace_radius[0]


def test_trajectory_on_real_correspondences(oracle):
    """Both outer iterations of a scan-to-map registration on the 50k-map fixture: the first solve starts 0.3 m / 3 deg off
    (many residuals in the Huber region), the second near the optimum (tolerance stops)."""
    _, mc, ms = common.small_world()
    stops = set()
    for pts, ring, truth, guess in common.scans(2):
        _, corner, surf = common.features_from_oracle(oracle, pts, ring)
        pose = np.array(guess, dtype=np.float64)
        for outer in range(2):
            corr = oracle.associate_scan2map(mc, ms, corner, surf, pose)
            keep = np.nonzero(corr["kind"] != 0)[0][::3]                 # every third accepted correspondence: seconds, not minutes
            s = _compare(oracle, corr[keep], pose, "real scan, outer %d" % outer)
            stops.add(s.termination)
            pose, _ = oracle.ceres_solve(corr, pose)
    assert len(stops) >= 1
