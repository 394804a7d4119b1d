"""CPU model of the plane fit's fast path (msf_loam_amd/csrc/msfl_math.cuh: plane_normal_centred).

The reference solves the 5 x 3 system A x = -1 by colPivHouseholderQr and only uses x / |x| (mapping_scan_matcher.cc:199-211).
The HIP kernel gets that direction as -adj(Q) c (Q: scatter of the centred points, c: centroid) and hands every neighbourhood
its two guards reject to the pivoted QR.  This test restates formula and guards in numpy f64 and checks on benign and
adversarial neighbourhoods (far from the origin, nearly collinear, grazing planes, planes through the origin, duplicates, exact
coplanarity) that a neighbourhood the guards ACCEPT has its normal within 1e-9 of the least-squares direction — the tolerance of
the GPU parity tests on records — and that ordinary neighbourhoods are accepted (the fallback stays rare)."""
import numpy as np


def plane_normal_centred(P):
    """P: (n, 5, 3) f64 (f32-exact values).  Returns (normal (n, 3), ok (n,)) like the device function."""
    c = P.sum(1) / 5.0
    q = P - c[:, None, :]
    Q = np.einsum("nji,njk->nik", q, q)
    Q00, Q01, Q02, Q11, Q12, Q22 = Q[:, 0, 0], Q[:, 0, 1], Q[:, 0, 2], Q[:, 1, 1], Q[:, 1, 2], Q[:, 2, 2]
    a00 = Q11 * Q22 - Q12 * Q12; a01 = Q02 * Q12 - Q01 * Q22; a02 = Q01 * Q12 - Q02 * Q11
    a11 = Q00 * Q22 - Q02 * Q02; a12 = Q01 * Q02 - Q00 * Q12; a22 = Q00 * Q11 - Q01 * Q01
    y = np.stack([a00 * c[:, 0] + a01 * c[:, 1] + a02 * c[:, 2],
                  a01 * c[:, 0] + a11 * c[:, 1] + a12 * c[:, 2],
                  a02 * c[:, 0] + a12 * c[:, 1] + a22 * c[:, 2]], 1)
    M00 = Q00 + 5 * c[:, 0] ** 2; M01 = Q01 + 5 * c[:, 0] * c[:, 1]; M02 = Q02 + 5 * c[:, 0] * c[:, 2]
    M11 = Q11 + 5 * c[:, 1] ** 2; M12 = Q12 + 5 * c[:, 1] * c[:, 2]; M22 = Q22 + 5 * c[:, 2] ** 2
    with np.errstate(all="ignore"):
        d0 = M00
        d1 = M11 - M01 * M01 / d0
        l21 = M12 - M01 * M02 / d0
        d2 = M22 - M02 * M02 / d0 - l21 * l21 / d1
        dmin = np.minimum(d0, np.minimum(d1, d2)); dmax = np.maximum(d0, np.maximum(d1, d2))
        trq = Q00 + Q11 + Q22
        cm = np.abs(c).max(1); ym = np.abs(y).max(1)
        ok = (dmin > 1e-10 * dmax) & (ym > 1e-6 * (trq * trq * cm))
        n = -y / np.linalg.norm(y, axis=1, keepdims=True)
    return n, ok


def lstsq_direction(P):
    out = np.zeros((len(P), 3))
    for i, A in enumerate(P):
        x = np.linalg.lstsq(A, -np.ones(5), rcond=None)[0]
        out[i] = x / np.linalg.norm(x)
    return out


def neighbourhoods(rng, n, centre_range, spread, aniso, noise, offset_scale=1.0):
    """n five-point sets near planes: in-plane spread `spread`, second in-plane axis scaled by `aniso`, out-of-plane `noise`."""
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    u = np.cross(nrm, rng.normal(size=(n, 3))); u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = np.cross(nrm, u)
    c0 = rng.uniform(-centre_range, centre_range, (n, 3)) * offset_scale
    a = rng.uniform(-spread, spread, (n, 5, 1)); b = rng.uniform(-spread, spread, (n, 5, 1)) * aniso
    e = rng.normal(0, 1, (n, 5, 1)) * noise
    P = c0[:, None, :] + a * u[:, None, :] + b * v[:, None, :] + e * nrm[:, None, :]
    return P.astype(np.float32).astype(np.float64)


def test_accepted_neighbourhoods_match_the_least_squares_direction():
    rng = np.random.default_rng(20260930)
    families = {
        "map-like": neighbourhoods(rng, 4000, 40.0, 0.5, 1.0, 0.01),
        "noisy": neighbourhoods(rng, 2000, 40.0, 0.5, 1.0, 0.08),
        "far": neighbourhoods(rng, 2000, 150.0, 0.3, 1.0, 0.005),
        "exactly coplanar (f32 grid)": np.round(neighbourhoods(rng, 1000, 30.0, 0.5, 1.0, 0.0) * 4) / 4,
        "nearly collinear": neighbourhoods(rng, 2000, 40.0, 0.5, 0.01, 0.002),
        "collinear": neighbourhoods(rng, 500, 40.0, 0.5, 0.0, 0.0),
        "near the origin": neighbourhoods(rng, 2000, 1.0, 0.4, 1.0, 0.01),
        "tiny": neighbourhoods(rng, 1000, 40.0, 0.01, 1.0, 0.0005),
    }
    # planes through the origin: n.p = 0 has no solution of n.p = -1 (rank-deficient for exactly coplanar points)
    P = neighbourhoods(rng, 1000, 30.0, 0.5, 1.0, 0.0)
    nrm = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    P = P - (np.einsum("nk,nk->n", P.mean(1), nrm))[:, None, None] * nrm[:, None, :]
    families["through the origin"] = P.astype(np.float32).astype(np.float64)
    dup = neighbourhoods(rng, 500, 40.0, 0.5, 1.0, 0.01); dup[:, 4] = dup[:, 3]; dup[:, 2] = dup[:, 1]
    families["duplicates"] = dup
    accepted = {}
    for name, P in families.items():
        n, ok = plane_normal_centred(P)
        ref = lstsq_direction(P[ok])
        err = np.abs(n[ok] - ref).max(1) if ok.any() else np.zeros(0)
        assert err.size == 0 or err.max() < 1e-9, (name, err.max())
        accepted[name] = ok.mean()
    assert accepted["map-like"] > 0.995 and accepted["noisy"] > 0.995 and accepted["far"] > 0.98, accepted
    assert accepted["near the origin"] > 0.95, accepted
    assert accepted["collinear"] == 0.0, accepted            # rank-deficient: always the reference's QR


def test_guards_reject_non_finite_input():
    P = np.zeros((3, 5, 3)); P[0] = np.nan; P[1, 2, 1] = np.inf; P[2] = 1.0     # NaN, Inf, five identical points
    with np.errstate(all="ignore"):
        _, ok = plane_normal_centred(P)
    assert not ok.any()
