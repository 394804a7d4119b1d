"""The C++ host-side mirror (include/msfl/scan_matcher.hpp: OdometryScanMatcher::MatchScan2Scan,
MappingScanMatcher::MatchScan2Map, ScanRegistration::Extract with the reference's names and
conventions) compiles with plain g++ against the C ABI (CPU) and, on the GPU, reproduces the ctypes
path bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

from msf_loam_amd import synth
from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_api_check")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s"])
    assert os.path.exists(EXE)


def test_cpp_host_mirror_compiles_and_links():
    _build()


def test_reference_adapter_compiles_against_both_type_families():
    """include/msfl/reference_adapter.hpp (the bodies INTEGRATION.md pastes into the reference as three call sites) compiles
    unchanged against the mirror PODs and against PCL / Eigen-shaped test types with the reference's 32-byte point layout
    (common/common.h:44-62), Eigen-style accessors and a non-const ToVector7 (tests/cpp/adapter_check.cpp)."""
    _build()
    for exe in ("adapter_check_mirror", "adapter_check_pcl"):
        assert os.path.exists(os.path.join(ROOT, "tests", "cpp", exe))


def test_header_is_plain_c_and_links():
    """include/msfl_c_api.h through gcc -std=c99: what a cgo / JNI / N-API binding generator sees."""
    _build()
    exe = os.path.join(ROOT, "tests", "cpp", "c_header_check")
    assert os.path.exists(exe)
    out = subprocess.check_output([exe]).decode()         # pure host calls: default params, API version, symbol addresses
    assert "29 entry points" in out and "sizeof(msfl_point)=16" in out and "sizeof(msfl_slam_result)=496" in out


@pytest.mark.gpu
def test_cpp_host_mirror_matches_ctypes_path(gpu, tmp_path):
    _build()
    _, mc, ms = common.small_world()
    pts, ring, truth, guess = common.scans(1)[0]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<i", len(pts))); f.write(pts.astype("<f4").tobytes()); f.write(ring.astype("<u2").tobytes())
        f.write(struct.pack("<i", len(mc))); f.write(mc.astype("<f4").tobytes())
        f.write(struct.pack("<i", len(ms))); f.write(ms.astype("<f4").tobytes())
        f.write(np.asarray(guess, "<f8").tobytes())
        from tests.test_deskew import _preintegration
        t, dq, dp = _preintegration(n=60, span=0.25)           # relative times of a scan reach 0.2 s when a ring wraps (msf_loam_node.cc:145-148)
        vel, grav = np.array([0.4, -0.2, 0.05]), np.array([0.0, 0.0, 9.81])
        f.write(struct.pack("<i", len(t)))
        for a in (t, dq, dp, vel, grav):
            f.write(np.asarray(a, "<f8").tobytes())
    subprocess.check_call([EXE, str(fin), str(fout)])
    raw = open(fout, "rb").read()
    pose_cpp = np.frombuffer(raw[:56], "<f8"); rel_cpp = np.frombuffer(raw[56:112], "<f8")
    odo_ok, n_sharp, n_ls, n_flat, n_lf = struct.unpack("<5i", raw[112:132])
    f = gpu.extract_features(pts, ring, extrinsic=np.array([0, 0, 0, 0, 0, 0, 1.0]))
    assert (n_sharp, n_ls, n_flat, n_lf) == (len(f["sharp"]), len(f["less_sharp"]), len(f["flat"]), len(f["less_flat"]))
    gpu.set_map(mc, ms)
    s, pose_py, _ = gpu.match_scan2map(f["full"][f["less_sharp"]], f["full"][f["less_flat"]], guess)
    assert np.array_equal(pose_cpp, pose_py)
    assert odo_ok == 1
    dt, dr = synth.pose_error(rel_cpp, np.array([0, 0, 0, 0, 0, 0, 1.0]))
    assert dt < 2e-3 and dr < 2e-4      # a scan registered against itself returns to identity
    # the reference-signature (8-argument) overload: LiDAR-only branch == the 6-argument call, deskew branch == the ctypes
    # chain msfl_delta_qp -> msfl_match_scan2map_deskew, both bit for bit
    pose8 = np.frombuffer(raw[132:188], "<f8"); pose8d = np.frombuffer(raw[188:244], "<f8")
    assert np.array_equal(pose8, pose_py)
    corner, surf = f["full"][f["less_sharp"]], f["full"][f["less_flat"]]
    sc, cdq, cdp = gpu.delta_qp(t, dq, dp, corner)
    ss, sdq, sdp = gpu.delta_qp(t, dq, dp, surf)
    assert sc == 0 and ss == 0
    s, pose_d, _ = gpu.match_scan2map_deskew(corner, surf, cdq, cdp, sdq, sdp, vel, grav, guess)
    assert s == 0 and np.array_equal(pose8d, pose_d)
    assert not np.array_equal(pose8d, pose8)
    # LaserSlam (msfl_slam_* behind the reference's AddLaserScan shape) == the ctypes binding on the same three scans
    from msf_loam_amd import capi
    sl = capi.Slam(0, max_scan_points=len(pts), max_rings=16, pose_odom2map=guess)
    recs = [sl.add_scan(pts, ring) for _ in range(3)]
    sl.close()
    for k, r in enumerate(recs):
        o = 244 + 112 * k
        assert np.array_equal(np.frombuffer(raw[o:o + 56], "<f8"), np.array(r.pose_odom[:])), k
        assert np.array_equal(np.frombuffer(raw[o + 56:o + 112], "<f8"), np.array(r.pose_map[:])), k


@pytest.mark.gpu
def test_reference_adapter_both_type_families_match_the_ctypes_path(gpu, tmp_path):
    """the adapter templates instantiated with the mirror PODs and with PCL / Eigen-shaped types run the same
    extraction -> scan-to-scan -> scan-to-map (both branches) -> map-store -> de-skew chain on the GPU: the two binaries write the
    same bytes, and those equal the ctypes path's results bit for bit."""
    _build()
    _, mc, ms = common.small_world()
    pts, ring, truth, guess = common.scans(1)[0]
    from tests.test_deskew import _preintegration
    t, dq, dp = _preintegration(n=60, span=0.25)
    vel, grav = np.array([0.4, -0.2, 0.05]), np.array([0.0, 0.0, 9.81])
    fin = tmp_path / "in.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<i", len(pts))); f.write(pts.astype("<f4").tobytes()); f.write(ring.astype("<u2").tobytes())
        f.write(struct.pack("<i", len(mc))); f.write(mc.astype("<f4").tobytes())
        f.write(struct.pack("<i", len(ms))); f.write(ms.astype("<f4").tobytes())
        f.write(np.asarray(guess, "<f8").tobytes())
        f.write(struct.pack("<i", len(t)))
        for a in (t, dq, dp, vel, grav):
            f.write(np.asarray(a, "<f8").tobytes())
    outs = []
    for exe in ("adapter_check_mirror", "adapter_check_pcl"):
        fout = tmp_path / (exe + ".bin")
        subprocess.check_call([os.path.join(ROOT, "tests", "cpp", exe), str(fin), str(fout)])
        outs.append(open(fout, "rb").read())
    assert outs[0] == outs[1], "the two instantiations of the adapter disagree"
    raw = outs[1]
    rel = np.frombuffer(raw[:56], "<f8")
    odo_ok, n_full, n_sharp, n_ls, n_flat, n_lf = struct.unpack("<6i", raw[56:80])
    pose = np.frombuffer(raw[80:136], "<f8"); pose_d = np.frombuffer(raw[136:192], "<f8")
    n_around, = struct.unpack("<i", raw[192:196])
    around4 = np.frombuffer(raw[196:260], "<f4").reshape(4, 4)
    lf2 = np.frombuffer(raw[260:292], "<f4").reshape(2, 4)
    f = gpu.extract_features(pts, ring, extrinsic=np.array([0, 0, 0, 0, 0, 0, 1.0]))
    assert (n_full, n_sharp, n_ls, n_flat, n_lf) == (len(f["full"]), len(f["sharp"]), len(f["less_sharp"]), len(f["flat"]), len(f["less_flat"]))
    corner, surf = f["full"][f["less_sharp"]], f["full"][f["less_flat"]]
    s, rel_py, _ = gpu.match_scan2scan(corner, f["ring"][f["less_sharp"]], surf, f["ring"][f["less_flat"]],
                                       f["full"][f["sharp"]], f["full"][f["flat"]], np.array([0.05, -0.03, 0.01, 0, 0, 0.005, 0.9999875]))
    assert odo_ok == 1 and s == 0 and np.array_equal(rel, rel_py)
    gpu.set_map(mc, ms)
    s, pose_py, _ = gpu.match_scan2map(corner, surf, guess)
    assert s == 0 and np.array_equal(pose, pose_py)
    sc, cdq, cdp = gpu.delta_qp(t, dq, dp, corner)
    ss, sdq, sdp = gpu.delta_qp(t, dq, dp, surf)
    s, pose_dpy, _ = gpu.match_scan2map_deskew(corner, surf, cdq, cdp, sdq, sdp, vel, grav, guess)
    assert sc == 0 and ss == 0 and s == 0 and np.array_equal(pose_d, pose_dpy) and not np.array_equal(pose_d, pose)
    from msf_loam_amd import capi
    g = capi.Grid(gpu, 3.0, 0.4)
    g.insert_scan(ms)
    want = g.get_surrounded(surf, pose_py)
    assert n_around == len(want) and np.array_equal(around4, want[:4])
    s, want_lf = gpu.deskew_cloud(t, dq, dp, surf, np.array([0, 0, 0, 1.0]), vel, grav)
    assert s == 0 and np.array_equal(lf2, want_lf[:2])
