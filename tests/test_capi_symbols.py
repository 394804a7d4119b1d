"""CPU: the C-ABI shared library loads and exports every symbol include/msfl_c_api.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "msfl_c_api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msfl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from msf_loam_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(capi.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert sorted(capi.EXPORTED) == names, "capi.EXPORTED out of sync with the header"


def test_params_struct_matches_header_defaults():
    from msf_loam_amd import capi
    p = capi.default_params()
    assert p.scan_period == 0.1 and p.min_range == 0.3 and p.sectors_per_ring == 6
    assert (p.max_sharp_per_sector, p.max_less_sharp_per_sector, p.max_flat_per_sector) == (2, 20, 4)
    assert p.odom_distance_sq_threshold == 25.0 and p.odom_nearby_scan == 2.5 and p.odom_min_correspondences == 10
    assert p.map_knn == 5 and p.map_knn_max_sq_dist == 1.0 and p.line_eigen_ratio == 3.0 and p.plane_tolerance == 0.2
    assert p.outer_iterations == 2 and p.max_lm_iterations == 6 and p.huber_delta == 0.1
    assert p.initial_trust_region_radius == 1e4 and p.min_relative_decrease == 1e-3
    assert p.function_tolerance == 1e-6 and p.gradient_tolerance == 1e-10 and p.parameter_tolerance == 1e-8
    assert p.max_consecutive_invalid_steps == 5
    assert capi.load().msfl_api_version() == 1
    assert capi.status_string(capi.MAP_TOO_SMALL) == "MAP_TOO_SMALL"


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a GPU msfl_create must fail (HIP_ERROR); the product has no CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from msf_loam_amd import capi
    with pytest.raises(capi.MsflError):
        capi.Handle(0)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "msf_loam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".inc", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for pat in (r"^\s*from\s+oracle", r"^\s*import\s+oracle", r"libmsfl_oracle", r"msfl_oracle\.h", r"\borc_[a-z]"):
                    assert not re.search(pat, text, flags=re.M), f"{f} references the oracle ({pat})"
