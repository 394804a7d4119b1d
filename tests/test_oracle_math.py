"""CPU tests pinning the oracle's restated third-party arithmetic (Eigen / FLANN / Ceres pieces)
against INDEPENDENT numpy/scipy formulations.  The reference has no tests on this path
(SURVEY.md §4), so these known-answer tests are what pins the oracle ("parity unpinned" otherwise).
"""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from msf_loam_amd import synth


def _rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def test_quat_rotate_and_matrix_match_scipy(oracle):
    rng = np.random.default_rng(1)
    for _ in range(50):
        q = _rand_quat(rng)
        v = rng.normal(size=3) * 10
        R = Rotation.from_quat(q).as_matrix()       # scipy uses [x y z w] too
        assert np.allclose(oracle.quat_rotate(q, v), R @ v, atol=1e-13)
        assert np.allclose(oracle.quat_to_matrix(q), R, atol=1e-14)


def test_transform_point_is_f32_of_f64_transform(oracle):
    rng = np.random.default_rng(2)
    for _ in range(50):
        pose = np.r_[rng.normal(size=3) * 5, _rand_quat(rng)]
        p = (rng.normal(size=3) * 20).astype(np.float32)
        want = (Rotation.from_quat(pose[3:]).as_matrix() @ p.astype(np.float64) + pose[:3])
        got = oracle.transform_point(pose, p)
        assert got.dtype == np.float32
        assert np.all(np.abs(got - want) <= np.spacing(np.abs(want).astype(np.float32)))
    ident = np.array([0, 0, 0, 0, 0, 0, 1.0])
    p = np.array([1.2345678, -9.87, 3.3], np.float32)
    assert np.array_equal(oracle.transform_point(ident, p), p)      # identity is bit-exact


def test_pose_plus_matches_rotation_composition(oracle):
    rng = np.random.default_rng(3)
    for scale in (1e-9, 1e-3, 0.3, 2.0):
        x = np.r_[rng.normal(size=3), _rand_quat(rng)]
        d = np.r_[rng.normal(size=3), rng.normal(size=3) * scale]
        out = oracle.pose_plus(x, d)
        assert np.allclose(out[:3], x[:3] + d[:3], atol=0)
        want = (Rotation.from_quat(x[3:]) * Rotation.from_rotvec(d[3:])).as_quat()
        assert min(np.abs(out[3:] - want).max(), np.abs(out[3:] + want).max()) < 1e-12
        assert abs(np.linalg.norm(out[3:]) - 1) < 1e-15
    # zero step keeps a unit quaternion (up to normalisation rounding)
    x = np.r_[1.0, 2.0, 3.0, _rand_quat(rng)]
    assert np.allclose(oracle.pose_plus(x, np.zeros(6)), x, atol=1e-15)


def test_pose_compose_matches_matrices(oracle):
    rng = np.random.default_rng(4)
    a = np.r_[rng.normal(size=3), _rand_quat(rng)]
    b = np.r_[rng.normal(size=3), _rand_quat(rng)]
    c = oracle.pose_compose(a, b)
    Ra, Rb = Rotation.from_quat(a[3:]), Rotation.from_quat(b[3:])
    assert np.allclose(c[:3], Ra.as_matrix() @ b[:3] + a[:3], atol=1e-13)
    assert np.allclose(Rotation.from_quat(c[3:]).as_matrix(), (Ra * Rb).as_matrix(), atol=1e-13)


def test_sym_eigen3_matches_eigh(oracle):
    rng = np.random.default_rng(5)
    for k in range(200):
        M = rng.normal(size=(3, 5)) * (10.0 ** rng.integers(-3, 3))
        if k % 3 == 0:
            M[:, 3:] = M[:, :2]          # rank deficient
        A = M @ M.T
        ev, V = oracle.sym_eigen3(A)
        w, U = np.linalg.eigh(A)
        assert np.allclose(ev, w, rtol=1e-12, atol=1e-12 * max(1.0, w[-1]))
        assert np.allclose(V.T @ V, np.eye(3), atol=1e-13)
        assert np.allclose(A @ V, V * ev, atol=1e-11 * max(1.0, w[-1]))
    ev, V = oracle.sym_eigen3(np.diag([3.0, 1.0, 2.0]))
    assert np.allclose(ev, [1, 2, 3])


def test_lstsq_5x3_matches_numpy(oracle):
    rng = np.random.default_rng(6)
    for _ in range(200):
        # 5 points near a plane far from the origin (the conditioning the matcher really sees)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        base = rng.normal(size=3) * 30
        t1 = np.cross(n, [1, 0, 0.3]); t1 /= np.linalg.norm(t1); t2 = np.cross(n, t1)
        P = base + rng.normal(size=(5, 1)) * 0.4 * t1 + rng.normal(size=(5, 1)) * 0.4 * t2 + rng.normal(size=(5, 1)) * 0.005 * n
        b = -np.ones(5)
        x, rank = oracle.lstsq_5x3(P, b)
        want = np.linalg.lstsq(P, b, rcond=None)[0]
        assert rank == 3
        assert np.allclose(x, want, rtol=1e-7, atol=1e-9)
        assert np.allclose(x / np.linalg.norm(x), want / np.linalg.norm(want), atol=1e-9)


def test_lstsq_reproduces_the_eigen_tutorial_example_and_its_rank_rule(oracle):
    """The example Eigen's own tutorial prints for `A.colPivHouseholderQr().solve(b)` ("Linear algebra and
    decompositions": A = [1 2 3; 4 5 6; 7 8 10], b = [3 3 4], solution -2 1 1), padded to five rows with zero
    equations, which leave a least-squares solution unchanged; and the rank-revealing side of the decomposition the
    matcher relies on for degenerate neighbourhoods: a column of zeros (five map points with z = 0 exactly) is
    rank 2 and gets a zero component, which is what ColPivHouseholderQR::solve returns for the dropped pivot."""
    A = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 10], [0, 0, 0], [0, 0, 0]], np.float64)
    b = np.array([3, 3, 4, 0, 0], np.float64)
    x, rank = oracle.lstsq_5x3(A, b)
    assert rank == 3 and np.allclose(x, [-2.0, 1.0, 1.0], atol=1e-12)
    P = np.array([[1.0, 2.0, 0], [2.0, -1.0, 0], [0.5, 0.25, 0], [3.0, 1.0, 0], [-1.0, 4.0, 0]])
    x, rank = oracle.lstsq_5x3(P, -np.ones(5))
    want = np.linalg.lstsq(P[:, :2], -np.ones(5), rcond=None)[0]
    assert rank == 2 and x[2] == 0.0 and np.allclose(x[:2], want, atol=1e-12)


def test_edge_and_plane_fit_match_numpy(oracle):
    rng = np.random.default_rng(7)
    n_line = n_not = 0
    for k in range(300):
        if k % 2 == 0:
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            P = rng.normal(size=3) * 20 + np.outer(rng.uniform(-0.5, 0.5, 5), d) + rng.normal(size=(5, 3)) * 0.01
        else:
            P = rng.normal(size=3) * 20 + rng.normal(size=(5, 3)) * 0.2
        P = P.astype(np.float32)
        ok, C, N = oracle.edge_fit(P)
        Pd = P.astype(np.float64)
        c = Pd.mean(axis=0)
        w, U = np.linalg.eigh((Pd - c).T @ (Pd - c))
        want_ok = w[2] > 3 * w[1]
        if abs(w[2] - 3 * w[1]) > 1e-9 * w[2]:
            assert ok == want_ok
        if ok and want_ok:
            n_line += 1
            assert abs(abs(N @ U[:, 2]) - 1) < 1e-10
            # C = center + 0.1 * dir (point_a), (a-b).normalized() = dir
            assert np.allclose(C, c + 0.1 * N, atol=1e-12)
        else:
            n_not += 1
    assert n_line > 50 and n_not > 50
    n_ok = 0
    for k in range(300):
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        t1 = np.cross(nrm, [0.2, 1, 0]); t1 /= np.linalg.norm(t1); t2 = np.cross(nrm, t1)
        noise = 0.005 if k % 2 == 0 else 0.3
        P = (rng.normal(size=3) * 30 + np.outer(rng.uniform(-0.6, 0.6, 5), t1) + np.outer(rng.uniform(-0.6, 0.6, 5), t2)
             + np.outer(rng.normal(size=5) * noise, nrm)).astype(np.float32)
        ok, C, N = oracle.plane_fit(P)
        Pd = P.astype(np.float64)
        x = np.linalg.lstsq(Pd, -np.ones(5), rcond=None)[0]
        x /= np.linalg.norm(x)
        c = Pd.mean(axis=0)
        dist = np.abs((Pd - c) @ x)
        if np.all(np.abs(dist - 0.2) > 1e-9):
            assert ok == bool(np.all(dist <= 0.2))
        if ok:
            n_ok += 1
            assert np.allclose(N, x, atol=1e-8) and np.allclose(C, c, atol=1e-12)
    assert n_ok > 100


def _numeric_jac(fn, pose, oracle, eps=1e-6):
    cols = []
    for k in range(6):
        d = np.zeros(6); d[k] = eps
        cols.append((fn(oracle.pose_plus(pose, d)) - fn(oracle.pose_plus(pose, -d))) / (2 * eps))
    return np.stack(cols, axis=1)


def test_factor_jacobians_vs_central_differences(oracle):
    """lidar_factor.cc Jacobians are the derivative w.r.t. the PoseLocalParameterization::Plus
    perturbation (first 6 columns; ComputeJacobian is [I6; 0])."""
    rng = np.random.default_rng(8)
    for _ in range(20):
        pose = np.r_[rng.normal(size=3) * 3, _rand_quat(rng)]
        p = rng.normal(size=3) * 10
        C = rng.normal(size=3) * 10
        N = rng.normal(size=3); N /= np.linalg.norm(N)
        r, J = oracle.edge_factor(pose, p, C, N)
        R = Rotation.from_quat(pose[3:]).as_matrix()
        assert np.allclose(r, np.cross(N, R @ p + pose[:3] - C), atol=1e-12)
        Jn = _numeric_jac(lambda x: oracle.edge_factor(x, p, C, N)[0], pose, oracle)
        assert np.allclose(J[:, :6], Jn, atol=2e-7) and np.all(J[:, 6] == 0)
        r, J = oracle.plane_factor(pose, p, C, N)
        assert np.allclose(r, N @ (R @ p + pose[:3] - C), atol=1e-12)
        Jn = _numeric_jac(lambda x: oracle.plane_factor(x, p, C, N)[0], pose, oracle)
        assert np.allclose(J[:, :6], Jn, atol=2e-7) and J[0, 6] == 0


def _f32_d2(cloud, q):
    """flann::L2_Simple<float>: ((dx*dx) + dy*dy) + dz*dz with every step rounded to f32."""
    c = cloud[:, :3].astype(np.float32)
    q = np.asarray(q, np.float32)
    d = c - q
    r = d[:, 0] * d[:, 0]
    r = (r + d[:, 1] * d[:, 1]).astype(np.float32)
    r = (r + d[:, 2] * d[:, 2]).astype(np.float32)
    return r


def test_knn_brute_and_kdtree_are_exact(oracle):
    rng = np.random.default_rng(9)
    cloud = np.zeros((5000, 4), np.float32)
    cloud[:, :3] = rng.uniform(-20, 20, (5000, 3))
    cloud[100:200, :3] = cloud[0:100, :3]            # exact duplicates -> distance ties
    cloud[300:400, :3] = np.round(cloud[300:400, :3])  # lattice points -> more ties
    tree = oracle.KdTree(cloud)
    for k in range(300):
        q = rng.uniform(-22, 22, 3).astype(np.float32)
        if k % 5 == 0:
            q = cloud[rng.integers(0, 5000), :3].copy()
        if k % 7 == 0:
            q = np.round(q)
        d2 = _f32_d2(cloud, q)
        order = np.lexsort((np.arange(len(cloud)), d2))[:5]     # (distance, index) total order
        ib, db = oracle.knn_brute(cloud, q, 5)
        ik, dk = tree.knn(q, 5)
        assert np.array_equal(ib, order) and np.array_equal(db, d2[order])
        assert np.array_equal(ik, ib) and np.array_equal(dk, db)
    i1, d1 = tree.knn(cloud[7, :3], 1)
    assert d1[0] == 0.0


def _make_problem(rng, n_edge=60, n_plane=400, outlier_frac=0.1):
    """Random point-to-line / point-to-plane correspondences around a ground-truth pose."""
    from oracle import oracle as orc
    truth = np.r_[rng.normal(size=3), _rand_quat(rng)]
    R = Rotation.from_quat(truth[3:]).as_matrix()
    corr = np.zeros(n_edge + n_plane, dtype=orc.CORR)
    for i in range(n_edge + n_plane):
        p = rng.uniform(-20, 20, 3)
        w = R @ p + truth[:3]
        N = rng.normal(size=3); N /= np.linalg.norm(N)
        off = rng.normal(size=3) * (0.5 if rng.uniform() < outlier_frac else 0.01)
        corr[i]["p"] = p
        corr[i]["N"] = N
        corr[i]["C"] = w + off if i >= n_edge else w + off + 0.3 * N
        corr[i]["kind"] = 1 if i < n_edge else 2
    return truth, corr


def _np_cost(corr, pose, a=0.1):
    """Independent numpy formulation of the Ceres objective 1/2 sum rho_huber(|r_i|^2)."""
    R = Rotation.from_quat(pose[3:] / np.linalg.norm(pose[3:])).as_matrix()
    d = corr["p"] @ R.T + pose[:3] - corr["C"]
    s = np.where(corr["kind"] == 1, np.sum(np.cross(corr["N"], d) ** 2, axis=1), np.sum(corr["N"] * d, axis=1) ** 2)
    s = np.where(corr["kind"] == 0, 0.0, s)
    rho = np.where(s <= a * a, s, 2 * a * np.sqrt(s) - a * a)
    return 0.5 * rho.sum()


def test_evaluate_matches_numpy_cost_and_gradient(oracle):
    rng = np.random.default_rng(10)
    truth, corr = _make_problem(rng)
    pose = synth.perturb_pose(truth, rng, 0.2, 2.0)
    cost, H, g = oracle.evaluate(corr, pose)
    assert abs(cost - _np_cost(corr, pose)) < 1e-10 * cost
    # g = gradient of the robust cost in the tangent space (exact for Huber: IRLS gradient)
    eps = 1e-6
    gn = np.array([(_np_cost(corr, oracle.pose_plus(pose, e * eps)) - _np_cost(corr, oracle.pose_plus(pose, -e * eps))) / (2 * eps)
                   for e in np.eye(6)])
    assert np.allclose(g, gn, rtol=1e-5, atol=1e-5)
    assert np.allclose(H, H.T) and np.all(np.linalg.eigvalsh(H) > 0)


def test_first_lm_step_matches_independent_formula(oracle):
    """Ceres LM step with Jacobi scaling: delta = -S (S H S + diag(clamp(diag(S H S)))/radius)^-1 S g."""
    rng = np.random.default_rng(11)
    truth, corr = _make_problem(rng)
    pose = synth.perturb_pose(truth, rng, 0.2, 2.0)
    cost, H, g = oracle.evaluate(corr, pose)
    S = np.diag(1.0 / (1.0 + np.sqrt(np.diag(H))))
    Hs = S @ H @ S
    D = np.diag(np.clip(np.diag(Hs), 1e-6, 1e32) / 1e4)
    delta = -S @ np.linalg.solve(Hs + D, S @ g)
    cand = oracle.pose_plus(pose, delta)
    opt = oracle.default_solver_options()
    opt.max_num_iterations = 1
    out, summ = oracle.ceres_solve(corr, pose, opt)
    assert summ.iterations == 1 and summ.trace_accepted[0] == 1
    assert np.allclose(out, cand, atol=1e-12)
    assert abs(summ.trace_step_norm[0] - np.linalg.norm(pose - cand)) < 1e-12
    model = -(g @ delta) - 0.5 * delta @ H @ delta
    assert abs(summ.trace_rel_decrease[0] - (cost - _np_cost(corr, cand)) / model) < 1e-8
    # accepted step: radius grows by 1/max(1/3, 1-(2 rho-1)^3)
    opt.max_num_iterations = 2
    _, summ2 = oracle.ceres_solve(corr, pose, opt)
    rho = summ2.trace_rel_decrease[0]
    assert abs(summ2.trace_radius[1] - 1e4 / max(1 / 3, 1 - (2 * rho - 1) ** 3)) < 1e-6


def test_converged_solve_is_the_huber_optimum(oracle):
    """Run the restated solver to convergence and check first-order optimality of the INDEPENDENT
    numpy cost, plus agreement with scipy's generic minimiser started at that point."""
    from scipy.optimize import minimize
    rng = np.random.default_rng(12)
    truth, corr = _make_problem(rng)
    pose = synth.perturb_pose(truth, rng, 0.2, 2.0)
    opt = oracle.default_solver_options()
    opt.max_num_iterations = 100
    opt.function_tolerance = 1e-16
    opt.parameter_tolerance = 1e-14
    sol, summ = oracle.ceres_solve(corr, pose, opt)
    assert summ.final_cost < summ.initial_cost
    f = lambda d: _np_cost(corr, oracle.pose_plus(sol, d))
    eps = 1e-6
    grad = np.array([(f(e * eps) - f(-e * eps)) / (2 * eps) for e in np.eye(6)])
    assert np.abs(grad).max() < 1e-6 * summ.final_cost + 1e-7
    res = minimize(f, np.zeros(6), method="BFGS", options=dict(gtol=1e-10))
    assert np.linalg.norm(res.x) < 1e-5 and res.fun >= f(np.zeros(6)) - 1e-12
    assert synth.pose_error(sol, truth)[0] < 0.02          # and it is near the truth


def test_six_iteration_budget_and_termination_bookkeeping(oracle):
    rng = np.random.default_rng(13)
    truth, corr = _make_problem(rng)
    pose = synth.perturb_pose(truth, rng, 0.3, 3.0)
    out, summ = oracle.ceres_solve(corr, pose)                # defaults: max 6 iterations
    assert 1 <= summ.iterations <= 6
    assert summ.final_cost <= summ.initial_cost
    # cost trace is monotone over accepted steps
    acc = [summ.trace_cost[i] for i in range(summ.iterations) if summ.trace_accepted[i] == 1]
    assert all(a >= b for a, b in zip([summ.initial_cost] + acc, acc))
    # empty problem: pose untouched bit for bit, termination "no residuals"
    empty = corr.copy(); empty["kind"] = 0
    out2, s2 = oracle.ceres_solve(empty, pose)
    assert np.array_equal(out2, pose) and s2.termination == 6 and s2.iterations == 0
    # already converged: parameter/function tolerance fires and the candidate is discarded
    out3, s3 = oracle.ceres_solve(corr, out)
    out4, s4 = oracle.ceres_solve(corr, out3)
    assert synth.pose_error(out4, out3)[0] < 1e-6


def test_voxel_grid_matches_numpy_centroids(oracle):
    rng = np.random.default_rng(14)
    pts = np.zeros((4000, 4), np.float32)
    pts[:, :3] = rng.uniform(-5, 5, (4000, 3))
    pts[:, 3] = rng.uniform(0, 0.1, 4000)
    out = oracle.voxel_grid(pts, 0.4)
    want = synth.voxel_downsample_np(pts, 0.4)
    assert out.shape == want.shape
    assert np.allclose(out, want, atol=2e-6)      # f32 vs f64 accumulation
    # every output lies in a distinct voxel, ordered by voxel index
    ijk = np.floor(out[:, :3] / np.float32(0.4)).astype(int)
    assert len(np.unique(ijk, axis=0)) == len(out)


def test_timed_batch_equals_plain_batch_and_reports_stages(oracle):
    """orc_match_scan2map_batch_timed (bench.py's stage table) runs the same arithmetic as the plain batch."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    inp = bench.build_inputs(3, 20000, 0)
    args = (inp["map_corner"], inp["map_surf"], inp["corner"], inp["corner_off"], inp["surf"], inp["surf_off"], inp["guesses"])
    p0, s0 = oracle.match_scan2map_batch(*args, threads=1, rebuild_tree_per_scan=True)
    p1, s1, st = oracle.match_scan2map_batch_timed(*args)
    assert np.array_equal(p0, p1) and np.array_equal(s0, s1)
    assert set(st) == {"build tree", "Data association", "Solver time"} and all(v > 0 for v in st.values())
