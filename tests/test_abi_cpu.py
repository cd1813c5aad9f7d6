"""CPU-only checks of the drop-in boundary: the library loads, exports every symbol include/gpis.h
declares, the struct mirrors agree with the C sizes, and the product fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(pkg):
    import __graft_entry__ as g
    g.build_hip()
    return pkg.load_library()


def test_header_symbols_exported(lib):
    header = open(os.path.join(ROOT, "include", "gpis.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char \*)\s*(gpis_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib.lib, name), "libgpis_hip.so does not export %s" % name
    assert declared <= set(lib.SYMBOLS) | {"gpis_abi_sizes"}


def test_struct_sizes_match_c(lib, pkg):
    got = dict(kv.split("=") for kv in lib.lib.gpis_abi_sizes().decode().split(","))
    assert int(got["gpis_ray_in"]) == pkg.RAY_IN.itemsize == 128
    assert int(got["gpis_seg_out"]) == pkg.SEG_OUT.itemsize == 96
    assert int(got["gpis_params"]) == pkg.PARAMS.itemsize
    assert int(got["gpis_scene_s"]) == pkg.SCENE_S.itemsize


def test_default_params_match_reference_defaults(lib, pkg):
    p = np.zeros((), dtype=pkg.PARAMS)
    lib.lib.gpis_default_params(p.ctypes.data)
    q = pkg.default_params()
    for f in p.dtype.names:
        if f.startswith("_pad"):
            continue
        assert np.array_equal(p[f], q[f]), f
    assert p["step_size"] == np.float32(0.01) and p["min_step"] == 8 and p["local_scale"] == 3.0


def test_no_cpu_fallback(lib, pkg):
    """Without a GPU gpis_create must fail with GPIS_ERR_NO_DEVICE, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback|hip"):
        pkg.Medium(pkg.params_for_config("C0"))


def test_product_does_not_reference_oracle():
    pkgdir = os.path.join(ROOT, "sparse-conv-gpis-tungsten_amd")
    for base, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".hh")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="replace").read()
                body = "\n".join(l for l in text.splitlines() if "never imports anything from" not in l)
                assert "oracle_bindings" not in body and "gpis_oracle" not in body and "liboracle" not in body, \
                    os.path.join(base, f)
