"""The reference's line test (mapping_scan_matcher.cc:130-151) reads the eigen-decomposition of a 3 x 3 covariance through
Eigen::SelfAdjointEigenSolver, which tridiagonalises and runs implicit QR steps; the oracle and the HIP kernels restate it
as cyclic Jacobi.  tests/eigen_ql.py restates Eigen's own algorithm; this file shows the choice cannot be seen at the
parity bar: on the covariances the matcher really meets, the two agree on every accept / reject decision and on the line
direction to ~1e-13, and around the `> 3 x` threshold they can only differ inside a band of a few ulps."""
import numpy as np
from scipy.spatial import cKDTree

from msf_loam_amd import synth
from tests import common, eigen_ql


def test_the_ql_restatement_is_an_eigensolver():
    rng = np.random.default_rng(11)
    for k in range(300):
        M = rng.normal(size=(3, 5)) * (10.0 ** rng.integers(-3, 3))
        if k % 4 == 0:
            M[:, 3:] = M[:, :2]                       # rank deficient
        A = M @ M.T
        if k % 7 == 0:
            A = np.diag(np.diag(A))                   # already diagonal: the deflation path
        w, Q = eigen_ql.eigh3(A)
        w, Q = np.array(w), np.array(Q)
        ref = np.linalg.eigh(A)[0]
        assert np.allclose(w, ref, rtol=1e-12, atol=1e-13 * max(1.0, ref[-1]))
        assert np.allclose(Q.T @ Q, np.eye(3), atol=1e-13)
        assert np.allclose(A @ Q, Q * w, atol=1e-11 * max(1.0, ref[-1]))


def _covariances(oracle, n_scans=3):
    """Covariances of the five nearest map corner points of every corner feature, built as the reference builds them:
    f32 points -> f64, centre = sum / 5, sum of outer products of the centred points (:131-138)."""
    _, mc, _ = common.small_world()
    tree = cKDTree(mc[:, :3].astype(np.float64))
    out = []
    for pts, ring, truth, guess in common.scans(n_scans):
        _, corner, _ = common.features_from_oracle(oracle, pts, ring)
        R = synth.quat_to_matrix(guess[3:])
        q = (R @ corner[:, :3].astype(np.float64).T).T + guess[:3]
        _, idx = tree.query(q, k=5)
        for row in idx:
            P = mc[row, :3].astype(np.float64)
            d = P - P.sum(0) / 5.0
            out.append(d.T @ d)
    return out


def test_jacobi_and_ql_agree_on_the_matchers_covariances(oracle):
    covs = _covariances(oracle)
    assert len(covs) > 400
    n_line = 0
    worst_dir = worst_ev = 0.0
    for S in covs:
        ev_j, V_j = oracle.sym_eigen3(S)
        ev_q, V_q = eigen_ql.eigh3(S)
        ev_q, V_q = np.array(ev_q), np.array(V_q)
        scale = max(ev_j[2], 1e-300)
        worst_ev = max(worst_ev, float(np.abs(ev_j - ev_q).max() / scale))
        line_j, line_q = ev_j[2] > 3 * ev_j[1], ev_q[2] > 3 * ev_q[1]
        margin = abs(ev_j[2] - 3 * ev_j[1]) / scale
        assert line_j == line_q or margin < 1e-12
        if line_j and line_q:
            n_line += 1
            worst_dir = max(worst_dir, 1.0 - abs(float(V_j[:, 2] @ V_q[:, 2])))      # the sign is free
    assert n_line > 100
    assert worst_ev < 1e-13 and worst_dir < 1e-12, (worst_ev, worst_dir)


def test_the_line_test_can_only_flip_within_a_few_ulps_of_its_threshold(oracle):
    rng = np.random.default_rng(12)
    flips_far = 0
    for k in range(400):
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        mid = 10.0 ** rng.uniform(-4, 0)
        delta = rng.choice([-1, 1]) * 10.0 ** rng.uniform(-15.5, -9)
        S = Q @ np.diag([mid * rng.uniform(0, 1), mid, 3 * mid * (1 + delta)]) @ Q.T
        S = 0.5 * (S + S.T)
        ev_j, _ = oracle.sym_eigen3(S)
        ev_q, _ = eigen_ql.eigh3(S)
        if (ev_j[2] > 3 * ev_j[1]) != (ev_q[2] > 3 * ev_q[1]) and abs(delta) > 1e-13:
            flips_far += 1
    assert flips_far == 0
