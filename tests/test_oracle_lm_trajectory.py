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


# Minimizer progress table Ceres prints for Powell's function in its own tutorial (Ceres Solver documentation,
# "Non-linear Least Squares" tutorial, section "Powell's Function": x0 = (3, -1, 0, 1), default options, i.e.
# LEVENBERG_MARQUARDT with Jacobi scaling, initial radius 1e4).  Columns used here: iter, cost, |step|, tr_ratio, tr_radius.
# Transcribed without network access: a |step| entry that was not certain is None; the run ends with
# "Final x1 = 0.000146222, x2 = -1.46222e-05, x3 = 2.40957e-05, x4 = 2.40957e-05".
POWELL_PUBLISHED = [
    (0, 1.075000e+02, 0.00e+00, 0.00e+00, 1.00e+04),
    (1, 5.036190e+00, 2.16e+00, 9.53e-01, 3.00e+04),
    (2, 3.148168e-01, 6.23e-01, 9.37e-01, 9.00e+04),
    (3, 1.967760e-02, 3.08e-01, 9.37e-01, 2.70e+05),
    (4, 1.229900e-03, None, 9.37e-01, 8.10e+05),
    (5, 7.687123e-05, None, 9.37e-01, 2.43e+06),
    (6, 4.804625e-06, None, 9.37e-01, 7.29e+06),
    (7, 3.003028e-07, None, 9.37e-01, 2.19e+07),
    (8, 1.877006e-08, None, 9.37e-01, 6.56e+07),
    (9, 1.173223e-09, None, 9.37e-01, 1.97e+08),
    (10, 7.333425e-11, None, 9.37e-01, 5.90e+08),
    (11, 4.584044e-12, None, 9.37e-01, 1.77e+09),
]


def test_the_restated_minimizer_reproduces_the_run_ceres_publishes_for_powells_function():
    """What pins the pinning: tests/ceres_numpy.py is itself a restatement, so its trust-region loop (Jacobi scaling fixed at
    iteration 0, the clamped LM diagonal, the model cost change, step quality, the radius update
    r / max(1/3, 1 - (2 rho - 1)^3), the evaluation order) is run on the one problem for which Ceres' documentation prints
    the solver's own per-iteration output.  Every printed cost is reproduced to its seven digits, and so are the step
    quality, the trust-region radius and the step norms of the rows recalled with them.  The chain is then:
    published Ceres run -> numpy restatement (this test) -> oracle trajectory (the tests above) -> HIP kernels (-m gpu).
    Not covered by this vector: the Huber corrector and the pose manifold (tested against independent formulas in
    test_oracle_math.py)."""
    s5, s10 = np.sqrt(5.0), np.sqrt(10.0)

    def powell(_, x, opt, want_jacobian=True):
        x1, x2, x3, x4 = x
        r = np.array([x1 + 10 * x2, s5 * (x3 - x4), (x2 - 2 * x3) ** 2, s10 * (x1 - x4) ** 2])
        J = np.array([[1, 10, 0, 0], [0, 0, s5, -s5], [0, 2 * (x2 - 2 * x3), -4 * (x2 - 2 * x3), 0],
                      [2 * s10 * (x1 - x4), 0, 0, -2 * s10 * (x1 - x4)]])
        return 0.5 * float(r @ r), r, J

    class Defaults(cn.Options):
        max_num_iterations = 50                        # Solver::Options default (the matcher lowers it to 6)

    x, t = cn.solve(None, np.array([3.0, -1.0, 0.0, 1.0]), Defaults, evaluate_fn=powell, plus_fn=lambda x, d: x + d, n_tangent=4)

    def digits(got, want, rel):
        return abs(got - want) <= rel * abs(want)
    assert digits(t.initial_cost, POWELL_PUBLISHED[0][1], 5e-7)
    radius_after = t.radius[1:] + [None]                  # the table prints the radius AFTER the iteration's update
    for it, cost, step, ratio, radius in POWELL_PUBLISHED[1:]:
        k = it - 1
        assert t.accepted[k] == 1
        assert digits(t.cost[k], cost, 5e-7), (it, t.cost[k], cost)
        assert digits(t.rel_decrease[k], ratio, 1e-3), (it, t.rel_decrease[k])
        assert digits(radius_after[k], radius, 5e-3), (it, radius_after[k])
        if step is not None:
            assert digits(t.step_norm[k], step, 5e-3), (it, t.step_norm[k])
    assert t.termination == "gradient" and t.final_cost < 1e-12        # "Gradient tolerance reached" in the printed report
    for got, want in zip(x, (0.000146222, -1.46222e-05, 2.40957e-05, 2.40957e-05)):       # the printed final parameters
        assert digits(got, want, 5e-6), (got, want)


def test_the_restated_minimizer_reproduces_the_ceres_hello_world_run():
    """The other run the Ceres tutorial prints in full ("Hello World!": one residual f(x) = 10 - x from x = 0.5):
    cost 4.512500e+01 -> 4.511598e-07 -> 5.012552e-16, steps 9.50e+00 and 9.50e-04, step quality 1.00, radius 3e4 -> 9e4.
    With a single column the numbers isolate the damping itself: the first step falls short of 9.5 by exactly the
    factor 1 / (1 + 1 / radius) that D = sqrt(diag(J'J) / radius) on the Jacobi-scaled column gives."""
    def f(_, x, opt, want_jacobian=True):
        r = np.array([10.0 - x[0]])
        return 0.5 * float(r @ r), r, np.array([[-1.0]])

    class Defaults(cn.Options):
        max_num_iterations = 50

    x, t = cn.solve(None, np.array([0.5]), Defaults, evaluate_fn=f, plus_fn=lambda x, d: x + d, n_tangent=1)
    assert t.initial_cost == 45.125
    for k, (cost, step, radius_after) in enumerate([(4.511598e-07, 9.50e+00, 3.00e+04), (5.012552e-16, 9.50e-04, 9.00e+04)]):
        assert t.accepted[k] == 1
        assert abs(t.cost[k] - cost) <= 5e-7 * cost, (k, t.cost[k])
        assert abs(t.step_norm[k] - step) <= 5e-3 * step
        assert abs(t.rel_decrease[k] - 1.0) < 5e-3
        assert abs(t.radius[k + 1] - radius_after) <= 1e-9 * radius_after
    assert abs(x[0] - 10.0) < 1e-6
