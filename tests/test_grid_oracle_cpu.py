"""GridNonstationaryCovariance (GPF.cpp:1326-1427) in the CPU restatement: the voxel lookup against a plain numpy statement of
VdbGrid::density (clamp to bounds + 2 / - 3, OpenVDB's PointSampler / BoxSampler — absent dependency, restated from its published
source: parity unpinned for the lookup), and the wrapper's variance / kernel-scale logic through the evaluator."""
import numpy as np
import pytest


def _grid(n=20, seed=3):
    rng = np.random.default_rng(seed)
    k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    blob = np.exp(-((i - n / 2) ** 2 + (j - n / 2) ** 2 + (k - n / 2) ** 2) / (0.1 * n * n))
    return (0.3 + 1.5 * blob + 0.05 * rng.random((n, n, n))).astype(np.float32)


def _world_to_index(n, half=1.5):
    s = (n - 1) / (2 * half)
    return np.array([[s, 0, 0, half * s], [0, s, 0, half * s], [0, 0, s, half * s], [0, 0, 0, 1]], dtype=np.float32)


def _params(pkg, separate):
    p = pkg.params_for_config("C3")
    p["impulse_density"] = 10
    p["grid_nonstationary"] = 1
    p["grid_offset"], p["grid_scale"] = 0.1, 1.25
    p["grid_surf_vol_amp_separate"] = separate
    p["grid_surf_vol_amp_thresh"] = 1.4
    p["grid_surf_amp_scale"], p["grid_vol_amp_scale"] = 0.8, 1.3
    p["grid_surf_ls_scale"], p["grid_vol_ls_scale"] = 0.7, 1.6
    return p


def _numpy_density(vox, T, pts, interpolate):
    n = vox.shape[0]
    f32 = np.float32
    x, y, z = (pts[:, c].astype(f32) for c in range(3))
    q = [((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3] for r in range(3)]
    q = [np.minimum(np.maximum(c, f32(0 + 2)), f32(n - 1 - 3)) for c in q]
    qd = [c.astype(np.float64) for c in q]
    if interpolate == "point":
        idx = [np.floor(c + 0.5).astype(int) for c in qd]
        return vox[idx[2], idx[1], idx[0]]
    fl = [np.floor(c) for c in qd]
    i, j, k = (c.astype(int) for c in fl)
    u, v, w = (qd[c] - fl[c] for c in range(3))

    def lerp(a, b, t):
        return (a + ((b - a).astype(np.float64) * t).astype(f32)).astype(f32)

    def at(di, dj, dk):
        return vox[k + dk, j + dj, i + di]
    return lerp(lerp(lerp(at(0, 0, 0), at(0, 0, 1), w), lerp(at(0, 1, 0), at(0, 1, 1), w), v),
                lerp(lerp(at(1, 0, 0), at(1, 0, 1), w), lerp(at(1, 1, 0), at(1, 1, 1), w), v), u)


@pytest.mark.parametrize("interpolate", ["linear", "point"])
def test_voxel_lookup_equals_the_numpy_statement(pkg, ob, interpolate):
    vox, T = _grid(), _world_to_index(20)
    p = _params(pkg, 0)
    orc = ob.Oracle(p)
    pts = np.random.default_rng(1).uniform(-1.9, 1.9, (5000, 3))          # beyond the box too: the clamp
    assert np.all(orc.grid_unscaled_variance(pts) == 1.0)                  # no grid yet: getUnscaledVariance = 1 (GPF.cpp:1390)
    orc.set_variance_grid(vox, T, interpolate)
    want = ((_numpy_density(vox, T, pts, interpolate) + np.float32(0.1)) * np.float32(1.25)).astype(np.float32)
    assert np.array_equal(orc.grid_unscaled_variance(pts), want)


def test_variance_scales_the_noise_and_the_threshold_picks_the_kernel_scale(pkg, ob):
    vox, T = _grid(), _world_to_index(20)
    rng = np.random.default_rng(2)
    q = np.zeros(256, dtype=pkg.QUERY)
    q["p"] = rng.uniform(-1.2, 1.2, (256, 3))
    q["dir"] = (0.0, 0.0, 1.0)
    q["pixel"][:, 0] = np.arange(256)
    # constant grids: the value is exactly proportional to the variance (amplitude = variance * sigma, GPF.cpp:1235-1237)
    p = _params(pkg, 0)
    p["grid_offset"], p["grid_scale"] = 0.0, 1.0
    p["mean"]["type"], p["mean"]["offset"] = pkg.MEAN_TYPE.HOMOGENEOUS, 0.0           # value = amplitude * noise + 0
    a, b = ob.Oracle(p), ob.Oracle(p)
    a.set_variance_grid(np.full((8, 8, 8), 1.0, np.float32), _world_to_index(8))
    b.set_variance_grid(np.full((8, 8, 8), 2.0, np.float32), _world_to_index(8))
    va, vb = a.eval_value(q)[0], b.eval_value(q)[0]
    assert np.abs(va).max() > 0 and np.array_equal(vb, 2 * va)
    # with the surface / volume split, the kernel scale follows the threshold: a medium whose grid lies entirely below the
    # threshold equals a plain ramp-less wrapper of kernel scale surf_ls (here through maxVal: both reduce to scale 1 of the max)
    p = _params(pkg, 1)
    s = ob.Oracle(p)
    s.set_variance_grid(vox, T)
    amp = s.grid_unscaled_variance(q["p"].astype(np.float64))
    assert (amp < 1.4).any() and (amp >= 1.4).any()                       # both branches of getVariance / getKernelScale are taken
    assert np.isfinite(s.eval_value(q)[0]).all()
    # the phase split reads the same unscaled variance (SCN.cpp:81-86)
    p["surf_vol_phase_separate"], p["surf_vol_phase_amp_thresh"] = 1, 1.4
    s2 = ob.Oracle(p)
    s2.set_variance_grid(vox, T)
    assert np.array_equal(s2.eval_value(q)[1], (amp >= 1.4).astype(np.int32))
    bad = _params(pkg, 0)
    bad["nonstationary"] = 0
    with pytest.raises(ValueError):
        ob.Oracle(bad)
