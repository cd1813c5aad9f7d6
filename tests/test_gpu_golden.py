"""HIP path against the committed golden fixtures (no oracle call for the expected values)."""
import ctypes
import glob
import os

import numpy as np
import pytest

from gpu_util import to_dev, dev_empty, to_host, stream_ptr

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_gpu_reference_primitives(pkg):
    g = np.load(os.path.join(GOLD, "ref_primitives.npz"))
    med = pkg.Medium(pkg.params_for_config("C0"))
    for arity in (1, 2, 3, 4):
        w = g["hash%d_in" % arity]
        d_w, d_o = to_dev(w), dev_empty(4 * len(w))
        med.call("gpis_xxhash32_batch", ctypes.c_size_t(len(w)), arity, d_w.data_ptr(), d_o.data_ptr(), stream_ptr())
        assert np.array_equal(to_host(d_o, np.uint32), g["hash%d_out" % arity])
    st = g["pcg_state"]
    d_s, d_o = to_dev(st), dev_empty(4 * 96 * len(st))
    med.call("gpis_pcg32_stream_batch", ctypes.c_size_t(len(st)), d_s.data_ptr(), ctypes.c_uint32(96), d_o.data_ptr(), stream_ptr())
    assert np.array_equal(to_host(d_o, np.uint32, (len(st), 96)), g["pcg_stream"])


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLD, "oracle_C*.npz")) if "image64" not in p))
def test_gpu_vs_golden(pkg, path):
    g = np.load(path)
    med = pkg.Medium(g["params"])
    exact = True      # rounds 1-2: not for "1d" / "multires" (double-precision libm); the device now evaluates the host libm bit for bit
    val, gid = med.eval_value(g["q"])
    grad = med.eval_gradient(g["q"])
    seg, coeff = med.sample_distance(g["rays"], want_coeff=True)
    seg2, coeff2 = med.sample_distance(g["shadow"], want_coeff=True)
    vis = med.transmittance(g["shadow"])
    assert np.array_equal(gid, g["gp_id"])
    if exact:
        assert np.array_equal(val, g["value"]) and np.array_equal(grad, g["grad"], equal_nan=True)
        for f in seg.dtype.names:
            assert np.array_equal(seg[f], g["seg"][f], equal_nan=True), f
            assert np.array_equal(seg2[f], g["seg2"][f], equal_nan=True), f
        assert np.array_equal(vis, g["vis"])
        assert np.array_equal(coeff2["value_scale"], g["coeff2"]["value_scale"])
        assert np.array_equal(coeff2["gradient_scale"], g["coeff2"]["gradient_scale"])
    else:
        assert np.allclose(val, g["value"], rtol=2e-5, atol=2e-6)
        assert np.allclose(grad, g["grad"], rtol=1e-4, atol=1e-4, equal_nan=True)
        assert (seg["exited"] != g["seg"]["exited"]).sum() <= 1
        same = seg["exited"] == g["seg"]["exited"]
        assert np.allclose(seg["t"][same], g["seg"]["t"][same], rtol=1e-4, atol=1e-4)
        assert (vis != g["vis"]).sum() <= 2


def test_gpu_image_fixture(pkg):
    import torch
    g = np.load(os.path.join(GOLD, "oracle_C0_image64.npz"))
    med = pkg.Medium(pkg.params_for_config("C0"))
    scene = np.zeros((), dtype=pkg.SCENE_S)
    med.L.lib.gpis_default_scene_s(scene.ctypes.data, 64, 64, 4)
    rad = torch.zeros(64 * 64, dtype=torch.float32, device="cuda")
    hits = torch.zeros(64 * 64, dtype=torch.int32, device="cuda")
    med.call("gpis_render_scene_s", scene.ctypes.data_as(ctypes.c_void_p), rad.data_ptr(), hits.data_ptr(), stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(rad.cpu().numpy().reshape(64, 64), g["radiance_sum"])
    assert np.array_equal(hits.cpu().numpy().reshape(64, 64).astype(np.uint32), g["hits"])
    rad.zero_()
    med.call("gpis_render_scene_s_paths", scene.ctypes.data_as(ctypes.c_void_p), 4, 0.8, rad.data_ptr(), stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(rad.cpu().numpy().reshape(64, 64), g["paths_radiance_sum"])
