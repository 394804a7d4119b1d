"""Independent restatement of what Eigen::SelfAdjointEigenSolver<Matrix3d>::compute does ([3P-recall] Eigen 3.3 / 3.4,
Eigenvalues/SelfAdjointEigenSolver.h and Tridiagonalization.h): scale by the largest coefficient, the closed-form 3 x 3
tridiagonalisation, implicit symmetric QR steps with Wilkinson shift and deflation, ascending sort.  Plain Python floats
(IEEE doubles), no numpy inside the iteration.  Test infrastructure only.

The oracle (and the HIP kernels) restate the solver as cyclic Jacobi; `mapping_scan_matcher.cc:136-140` consumes the
largest / middle eigenvalues (line test `> 3 x`) and the largest eigenvector.  tests/test_oracle_eigen_ql.py checks that
the two algorithms make the same decisions and agree on the direction far below the parity bar.
"""
import math

EPS = 2.220446049250313e-16
TINY = 2.2250738585072014e-308


def _givens(p, q):
    """JacobiRotation::makeGivens (real case): c, s with [c s; -s c]^T [p; q] = [r; 0]."""
    if q == 0.0:
        return (1.0 if p >= 0 else -1.0), 0.0
    if p == 0.0:
        return 0.0, (-1.0 if q > 0 else 1.0)
    if abs(p) > abs(q):
        t = q / p
        u = math.sqrt(1.0 + t * t)
        if p < 0:
            u = -u
        c = 1.0 / u
        return c, -t * c
    t = p / q
    u = math.sqrt(1.0 + t * t)
    if q < 0:
        u = -u
    s = -1.0 / u
    return -t * s, s


def _qr_step(d, e, start, end, Q):
    td = (d[end - 1] - d[end]) * 0.5
    ee = e[end - 1]
    mu = d[end]
    if td == 0.0:
        mu -= abs(ee)
    elif ee != 0.0:
        e2 = ee * ee
        h = math.hypot(td, ee)
        if e2 == 0.0:
            mu -= ee / ((td + (h if td > 0 else -h)) / ee)
        else:
            mu -= e2 / (td + (h if td > 0 else -h))
    x = d[start] - mu
    z = e[start]
    k = start
    while k < end and z != 0.0:
        c, s = _givens(x, z)
        sdk = s * d[k] + c * e[k]
        dkp1 = s * e[k] + c * d[k + 1]
        d[k] = c * (c * d[k] - s * e[k]) - s * (c * e[k] - s * d[k + 1])
        d[k + 1] = s * sdk + c * dkp1
        e[k] = c * sdk - s * dkp1
        if k > start:
            e[k - 1] = c * e[k - 1] - s * z
        x = e[k]
        if k < end - 1:
            z = -s * e[k + 1]
            e[k + 1] = c * e[k + 1]
        # Q = Q * G  (applyOnTheRight(k, k + 1, rot))
        for r in range(3):
            a, b = Q[r][k], Q[r][k + 1]
            Q[r][k] = c * a - s * b
            Q[r][k + 1] = s * a + c * b
        k += 1


def eigh3(A):
    """A: 3 x 3 symmetric (nested lists / array, lower triangle read).  Returns (eigenvalues ascending, eigenvectors as
    columns of a 3 x 3 nested list)."""
    m = [[float(A[i][j]) if j <= i else float(A[j][i]) for j in range(3)] for i in range(3)]
    scale = max(abs(m[i][j]) for i in range(3) for j in range(i + 1))
    if scale == 0.0:
        scale = 1.0
    for i in range(3):
        for j in range(i + 1):
            m[i][j] /= scale
    # tridiagonalization_inplace_selector<MatrixType, 3, false>
    d = [m[0][0], 0.0, 0.0]
    e = [0.0, 0.0]
    v1norm2 = m[2][0] * m[2][0]
    if v1norm2 <= TINY:
        d[1], d[2] = m[1][1], m[2][2]
        e[0], e[1] = m[1][0], m[2][1]
        Q = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
    else:
        beta = math.sqrt(m[1][0] * m[1][0] + v1norm2)
        inv = 1.0 / beta
        m01, m02 = m[1][0] * inv, m[2][0] * inv
        q = 2.0 * m01 * m[2][1] + m02 * (m[2][2] - m[1][1])
        d[1] = m[1][1] + m02 * q
        d[2] = m[2][2] - m02 * q
        e[0] = beta
        e[1] = m[2][1] - m01 * q
        Q = [[1.0, 0.0, 0.0], [0.0, m01, m02], [0.0, m02, -m01]]
    # computeFromTridiagonal_impl
    n, end, start, it = 3, 2, 0, 0
    while end > 0:
        for i in range(start, end):
            if abs(e[i]) <= (abs(d[i]) + abs(d[i + 1])) * EPS or abs(e[i]) <= TINY:
                e[i] = 0.0
        while end > 0 and e[end - 1] == 0.0:
            end -= 1
        if end <= 0:
            break
        it += 1
        if it > 30 * n:
            raise RuntimeError("no convergence")
        start = end - 1
        while start > 0 and e[start - 1] != 0.0:
            start -= 1
        _qr_step(d, e, start, end, Q)
    # ascending selection sort, columns follow
    for i in range(2):
        k = min(range(i, 3), key=lambda j: d[j])
        if k != i:
            d[i], d[k] = d[k], d[i]
            for r in range(3):
                Q[r][i], Q[r][k] = Q[r][k], Q[r][i]
    return [x * scale for x in d], Q
