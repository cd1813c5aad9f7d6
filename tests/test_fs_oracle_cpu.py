"""Function-space comparison path, oracle side: PARITY UNPINNED against the reference (its function-space TUs
include Boost and cannot be built here, and it holds no fixture for them), so the restatement is checked against what the
algorithm must produce: numpy's eigen-decomposition, the covariance's own derivatives, and the statistics of the samples."""
import numpy as np
import pytest


def _params(pkg, ctx, n=32, step=0.0):
    p = pkg.params_for_config("C0")
    p["single_realization"] = 0
    p["correlation_context"] = getattr(pkg.CTX, ctx)
    p["mean"]["type"] = pkg.MEAN_TYPE.HOMOGENEOUS
    p["mean"]["offset"] = 1.0                      # 10 sigma above zero: a segment never crosses, its 'values' are the sample
    p["sigma"], p["length_scale"] = 0.1, 0.05
    p["aniso"] = (1.0, 1.0, 1.0)
    p["fs_sample_points"], p["fs_step_size"] = n, step
    return p


def _rays(pkg, n, near=0.0, far=0.32, first=1, seed=5):
    rng = np.random.default_rng(seed)
    r = np.zeros(n, dtype=pkg.RAY_IN)
    r["pos"] = (0.1, -0.2, 0.3)
    r["dir"] = (0.0, 0.0, 1.0)
    r["near_t"], r["far_t"] = near, far
    r["first_scatter"] = first
    r["pixel"][:, 0] = np.arange(n) % 1024
    r["pixel"][:, 1] = np.arange(n) // 1024
    st = np.zeros(n, dtype=pkg.FS_STATE)
    st["sampler_state"] = rng.integers(1, 2**63, size=n, dtype=np.uint64)
    return r, st


def test_eigh_and_square_root_against_numpy(pkg, ob):
    orc = ob.Oracle(_params(pkg, "NONE"), threads=1)
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 7, 33, 64, 66):
        a = rng.standard_normal((n, n))
        a = a + a.T
        w, v = orc.fs_eigh(a)
        w_np = np.linalg.eigvalsh(a)
        assert np.allclose(w, w_np, rtol=1e-10, atol=1e-10 * max(1.0, np.abs(w_np).max())), n
        assert np.allclose(a @ v, v * w, atol=1e-9 * max(1.0, np.abs(w_np).max())), n
        assert np.allclose(v.T @ v, np.eye(n), atol=1e-10), n
    # a numerically singular covariance (squared exponential on a fine grid): LLT fails, the eigen square root takes over
    x = np.linspace(0, 0.3, 64)
    s = 0.01 * np.exp(-(x[:, None] - x[None, :]) ** 2 / (2 * 0.05 ** 2))
    t = orc.fs_norm_transform(s)
    assert np.allclose(t @ t.T, s, atol=1e-9)
    assert np.abs(np.triu(t, 1)).max() > 1e-6                  # not a Cholesky factor
    # a well-conditioned one: the Cholesky factor
    s2 = s + 1e-3 * np.eye(64)
    t2 = orc.fs_norm_transform(s2)
    assert np.allclose(t2, np.linalg.cholesky(s2), atol=1e-10)


def test_covariance_derivatives(pkg, ob):
    p = _params(pkg, "NONE")
    p["aniso"] = (1.0, 2.0, 0.5)
    orc = ob.Oracle(p, threads=1)
    a, b = np.array([0.1, 0.2, -0.1]), np.array([0.13, 0.17, -0.06])
    da, db = np.array([0.3, -0.5, 0.8]), np.array([-0.2, 0.9, 0.4])
    h = 1e-6
    k = lambda x, y: orc.fs_cov(0, 0, x, y, da, db)
    d = b - a
    s2, l2 = float(np.float32(0.1) * np.float32(0.1)), float(np.float32(2) * (np.float32(0.05) * np.float32(0.05)))   # float members
    assert abs(k(a, b) - s2 * np.exp(-(d[0] ** 2 + 2 * d[1] ** 2 + 0.5 * d[2] ** 2) / l2)) < 1e-15
    assert abs(orc.fs_cov(1, 0, a, b, da, db) - (k(a + h * da, b) - k(a - h * da, b)) / (2 * h)) < 1e-7
    assert abs(orc.fs_cov(0, 1, a, b, da, db) - (k(a, b + h * db) - k(a, b - h * db)) / (2 * h)) < 1e-7
    fd2 = (orc.fs_cov(1, 0, a, b + h * db, da, db) - orc.fs_cov(1, 0, a, b - h * db, da, db)) / (2 * h)
    assert abs(orc.fs_cov(1, 1, a, b, da, db) - fd2) < 1e-5


def test_prior_samples_have_the_gp_statistics(pkg, ob):
    """first segment of a path (no context): the 32 values are one draw of N(mean, K) at the segment's points"""
    p = _params(pkg, "NONE")
    orc = ob.Oracle(p, threads=8)
    rays, st = _rays(pkg, 20000)
    out, st2 = orc.fs_sample_distance(rays, st)
    assert np.all(out["ok"] == 1) and np.all(out["exited"] == 1) and np.all(st2["has_context"] == 1)
    assert np.all(st2["n_points"] == 32) and np.all(st2["is_intersect"] == 0)
    assert np.all(st2["sampler_state"] != st["sampler_state"])
    v = (st2["values"][:, :32] - 1.0) / 0.1
    z = st2["points"][:, :32, 2] - 0.3                       # march parameter of every point
    assert np.allclose(z[:, 0], 0.001) and np.allclose(z[:, -1], 0.32)
    assert np.all(np.diff(z, axis=1) >= 0)
    assert abs(v.mean()) < 0.02 and np.all(np.abs(v.std(axis=0) - 1.0) < 0.03)
    # covariance between the first point and the others follows exp(-d^2 / (2 l^2)) — the points move with tOffset, so bin by distance
    dz = z[:, 5] - z[:, 0]
    want = np.exp(-dz ** 2 / (2 * 0.05 ** 2))
    assert abs(np.mean(v[:, 0] * v[:, 5]) - want.mean()) < 0.03
    # the sampled gradient at the segment end is N(0, sigma^2 / l^2) per component around the conditional mean: finite, varied
    assert np.all(np.isfinite(out["aniso"])) and out["aniso"].std(axis=0).min() > 0.1


@pytest.mark.parametrize("ctx", ["RENEWAL", "RENEWAL_PLUS", "GLOBAL"])
def test_conditioning_continues_the_field(pkg, ob, ctx):
    """second segment of a path, starting where the first ended: with a memory the first new value (0.1 steps beyond the
    conditioning point) continues the last one; without one it does not"""
    p = _params(pkg, ctx)
    orc = ob.Oracle(p, threads=8)
    rays, st = _rays(pkg, 4000)
    out, st1 = orc.fs_sample_distance(rays, st)
    last = st1["values"][:, 31].copy()
    r2 = rays.copy()
    r2["near_t"], r2["far_t"], r2["first_scatter"], r2["bounce"] = 0.32, 0.64, 0, 1
    out2, st2 = orc.fs_sample_distance(r2, st1)
    assert np.all(out2["ok"] == 1)
    first_new = st2["values"][:, 0]
    assert np.corrcoef(last, first_new)[0, 1] > 0.995, ctx
    assert np.abs(first_new - last).max() < 0.02
    none = ob.Oracle(_params(pkg, "NONE"), threads=8)
    _, s1 = none.fs_sample_distance(rays, st)
    _, s2 = none.fs_sample_distance(r2, s1)
    assert abs(np.corrcoef(s1["values"][:, 31], s2["values"][:, 0])[0, 1]) < 0.1


def test_crossing_and_context_records(pkg, ob):
    """a zero-mean field crosses: the state then ends with the crossing point twice (value, derivative), the value there is the
    interpolated zero, the derivative is the slope, and the sampled normal's component along the ray is that slope"""
    p = _params(pkg, "RENEWAL_PLUS", n=64, step=0.01)
    p["mean"]["offset"] = 0.0
    orc = ob.Oracle(p, threads=8)
    rays, st = _rays(pkg, 3000, far=2.0)
    out, st2 = orc.fs_sample_distance(rays, st)
    hit = (out["exited"] == 0) & (out["ok"] == 1)
    assert hit.sum() > 1000
    i = np.nonzero(hit)[0]
    npnt = st2["n_points"][i]
    assert np.all(st2["is_intersect"][i] == 1) and np.all(npnt >= 3)
    vals = st2["values"][i, :]
    assert np.abs(vals[np.arange(len(i)), npnt - 2]).max() < 1e-12             # lerp(prev, curr, offset) = 0 up to rounding
    slope = vals[np.arange(len(i)), npnt - 1]
    assert np.allclose(out["aniso"][i, 2], slope, rtol=1e-12)                   # ray along +z: the normal component is the slope
    assert np.all(st2["derivs"][i, npnt - 1] == 1) and np.all(st2["derivs"][i, npnt - 2] == 0)
    assert np.allclose(st2["points"][i, npnt - 1, 2] - 0.3, out["t"][i])
    # rejected samples (gradient pointing along the ray) are the ok = 0 ones
    assert np.all(out["aniso"][(out["ok"] == 0) & (out["exited"] == 0), 2] > 0)
    # segments longer than 64 steps of 0.01 are marched in several batches: some hits lie beyond the first batch
    assert (out["t"][i] > 0.64).sum() >= 1
    vis, _ = orc.fs_transmittance(rays, st)
    assert 0.02 < vis.mean() < 0.98 or vis.mean() < 0.02
