"""Function-space comparison path on the GPU against the CPU restatement: the same matrices, factorisations and
variates in the same order of IEEE operations — only exp / log / sin / cos differ (ocml against glibc), so sampled values agree
to ~1e-9 wherever the two sides take the same branches.  The branches are fragile BY CONSTRUCTION of the reference's algorithm:
a squared-exponential covariance on 32-64 points is numerically singular, so Eigen::LLT fails on a pivot that is rounding noise
(its sign decides between the Cholesky factor and the eigen square root, Gaussian.cpp:139-160 — two different realisations of
the same distribution), and the pseudo-inverse cuts eigenvalues at 1e6 eps (GaussianProcess.cpp:645-662).  A last-bit difference
in exp() flips such a decision, or is amplified through a noise-sized pivot, on ~5 % of the segments of such a configuration
(measured: 17 of 384 at 32 points 0.02 apart, l = 0.05).  The test therefore runs well-conditioned configurations (points
0.8 l apart: Cholesky succeeds with a margin) with a 1 % bound on the first segment.  The singular ones — which are what
exercises the eigen square root and the cut-off — are compared as distributions (hit rate, mean free path) plus the requirement
that at least half of the segments still agree value for value to 1 % of sigma; whatever agrees in its sampled values to 1e-7
must agree tightly in everything derived from them.  Parity of the restatement itself against the reference: unpinned (tests/test_fs_oracle_cpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(pkg, ctx, n, step, offset, aniso=(1.0, 1.0, 1.0), mean="HOMOGENEOUS"):
    p = pkg.params_for_config("C0")
    p["single_realization"] = 0
    p["correlation_context"] = getattr(pkg.CTX, ctx)
    p["mean"]["type"] = getattr(pkg.MEAN_TYPE, mean)
    p["mean"]["offset"] = offset
    p["sigma"], p["length_scale"] = 0.1, 0.05
    p["aniso"] = aniso
    p["fs_sample_points"], p["fs_step_size"] = n, step
    return p


def _rays(pkg, n, seed, near=0.0, far=0.5):
    rng = np.random.default_rng(seed)
    r = np.zeros(n, dtype=pkg.RAY_IN)
    r["pos"] = rng.uniform(-0.5, 0.5, (n, 3))
    d = rng.standard_normal((n, 3))
    r["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    r["near_t"] = near
    r["far_t"] = far + rng.uniform(0, 0.3, n)
    r["first_scatter"] = 1
    r["pixel"][:, 0] = np.arange(n)
    st = np.zeros(n, dtype=pkg.FS_STATE)
    st["sampler_state"] = rng.integers(1, 2**63, size=n, dtype=np.uint64)
    return r, st


def _compare(got, want, st_g, st_o, tag, frac, vtol=1e-7):
    go, wo = got, want
    n = len(go)
    same = (go["exited"] == wo["exited"]) & (go["ok"] == wo["ok"]) & (st_g["n_points"] == st_o["n_points"])
    flips = int((~same).sum())
    i = np.nonzero(same)[0]
    # a segment whose two sides took different square roots (Cholesky here, eigen there) can land in the same outcome class by
    # chance: its sampled values differ at O(sigma); such segments count as flips too, the rest must agree tightly
    k = np.arange(st_g["values"].shape[1])[None, :] < st_g["n_values"][i][:, None]
    dv = np.abs(np.where(k, st_g["values"][i] - st_o["values"][i], 0.0)).max(axis=1)
    singular = frac >= 0.125
    if singular:
        vtol = max(vtol, 1e-3)      # 1 % of sigma: eigenvectors of near-degenerate eigenvalues turn within their cluster
    bad = dv > vtol
    if singular:
        # numerically singular configuration: distributions, and a majority of identical segments
        assert flips + bad.sum() <= n // 2, (tag, "sampled values differ on %d, outcomes on %d of %d segments" % (bad.sum(), flips, n))
        assert abs(float((go["exited"] == 0).mean()) - float((wo["exited"] == 0).mean())) < 0.05, tag
        hg, hw = go["t"][go["exited"] == 0], wo["t"][wo["exited"] == 0]
        if len(hw) > 50:
            assert abs(hg.mean() - hw.mean()) < 0.15 * hw.mean(), (tag, hg.mean(), hw.mean())
    else:
        assert flips + bad.sum() <= max(2, int(n * frac)), (tag, "sampled values differ on %d, outcomes on %d of %d segments" % (bad.sum(), flips, n), dv.max())
    j = i[~bad]
    assert np.array_equal(st_g["sampler_state"][j], st_o["sampler_state"][j]), tag
    assert np.array_equal(st_g["derivs"][j], st_o["derivs"][j]) and np.array_equal(st_g["is_intersect"][j], st_o["is_intersect"][j])
    assert np.array_equal(go["gp_id"][j], wo["gp_id"][j])
    if vtol > 1e-7:
        # the GLOBAL context conditions on every point of the previous segment plus its crossing point (a 20-66 entry, numerically
        # singular system): last-bit differences of exp() are amplified to ~1e-3 sigma through the pseudo-inverse.  Crossing
        # positions then agree to a fraction of a step only.
        assert np.allclose(go["t"][j], wo["t"][j], atol=0.02), tag
        return j
    assert np.allclose(go["t"][j], wo["t"][j], rtol=1e-7, atol=1e-9), tag
    # the sampled normal is conditioned on ALL points of the segment plus the crossing point, which may lie arbitrarily close to
    # one of them: its pseudo-inverse is in the fragile regime for every configuration — the 12.5 % bound applies
    gbad = ~np.all(np.isclose(go["aniso"][j], wo["aniso"][j], rtol=1e-6, atol=1e-6), axis=1)
    assert gbad.sum() <= max(2, n // 8), (tag, "sampled normals differ on %d of %d segments" % (gbad.sum(), n))
    j = j[~gbad]
    assert np.allclose(st_g["points"][j], st_o["points"][j], rtol=1e-9, atol=1e-9), tag
    assert np.allclose(go["weight"][j], wo["weight"][j])
    assert np.allclose(go["sample_t"][j], wo["sample_t"][j], rtol=1e-6, atol=1e-7) and np.allclose(go["p"][j], wo["p"][j], atol=1e-6)
    return j


@pytest.mark.parametrize("ctx,n,step,offset,frac", [
    ("NONE", 12, 0.0, 0.0, 0.01), ("RENEWAL", 16, 0.04, 0.05, 0.01), ("RENEWAL_PLUS", 14, 0.0, 0.0, 0.01), ("GLOBAL", 14, 0.05, 0.1, 0.01),
    ("GLOBAL", 17, 0.0, 1.0, 0.01), ("NONE", 32, 0.0, 0.0, 0.125), ("RENEWAL_PLUS", 64, 0.01, 0.05, 0.125), ("GLOBAL", 64, 0.01, 0.1, 0.125)])
def test_function_space_path(pkg, ob, ctx, n, step, offset, frac):
    params = _params(pkg, ctx, n, step, offset, aniso=(1.0, 0.7, 1.4))
    med, orc = pkg.Medium(params), ob.Oracle(params, threads=16)
    rays, st = _rays(pkg, 384, seed=11 + n)
    got, st_g = med.fs_sample_distance(rays, st)
    want, st_o = orc.fs_sample_distance(rays, st)
    assert (want["exited"] == 0).sum() > 20 or offset >= 1.0
    j = _compare(got, want, st_g, st_o, (ctx, "first"), frac)
    # second segment of the path: leaves the point where the first ended, conditioned on the context just written
    r2 = rays[j].copy()
    t = want["sample_t"][j]
    r2["pos"] = rays["pos"][j] + rays["dir"][j] * t[:, None]
    rng = np.random.default_rng(5)
    d = rng.standard_normal((len(j), 3))
    r2["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    r2["near_t"], r2["far_t"] = 0.0, 0.4
    r2["first_scatter"], r2["bounce"] = 0, 1
    r2["last_aniso"] = want["aniso"][j]
    ok = want["ok"][j] == 1
    got2, st_g2 = med.fs_sample_distance(r2[ok], st_o[j][ok])
    want2, st_o2 = orc.fs_sample_distance(r2[ok], st_o[j][ok])
    _compare(got2, want2, st_g2, st_o2, (ctx, "second"), max(frac, 0.125) if ctx == "GLOBAL" else frac, 5e-4 if ctx == "GLOBAL" else 1e-7)
    # shadow segments on a copy of the state
    vis_g, sv_g = med.fs_transmittance(r2[ok], st_o[j][ok])
    vis_o, sv_o = orc.fs_transmittance(r2[ok], st_o[j][ok])
    assert (vis_g != vis_o).sum() <= max(2, int(len(vis_o) * frac))


def test_function_space_spherical_mean_and_errors(pkg, ob):
    params = _params(pkg, "RENEWAL_PLUS", 48, 0.02, 0.0, mean="SPHERICAL")
    params["mean"]["radius"] = 0.4
    med, orc = pkg.Medium(params), ob.Oracle(params, threads=16)
    rays, st = _rays(pkg, 256, seed=3, far=1.5)
    rays["pos"] = rays["pos"] * 2.4
    got, st_g = med.fs_sample_distance(rays, st)
    want, st_o = orc.fs_sample_distance(rays, st)
    _compare(got, want, st_g, st_o, "spherical", 0.125)
    bad = params.copy()
    bad["fs_sample_points"] = 65
    with pytest.raises(RuntimeError):
        pkg.Medium(bad).fs_sample_distance(rays[:4], st[:4])
    bad = params.copy()
    bad["nonstationary"] = 1
    with pytest.raises(RuntimeError):
        pkg.Medium(bad).fs_sample_distance(rays[:4], st[:4])
    with pytest.raises(RuntimeError):
        ob.Oracle(bad).fs_sample_distance(rays[:4], st[:4])


def test_dense_linear_algebra_bit_equal_to_the_reference_eigen(pkg):
    """The device's eigen-solver, LLT / normTransform and pseudo-inverse (gpis_fs_linalg_batch, one wave per matrix) on the matrices
    of tests/golden/ref_fs_primitives.npz — produced by the reference's vendored Eigen compiled in place under the reference's flags
    (tests/golden/make_golden.py: ref_fs_primitives).  Only +, -, *, / and sqrt run in these routines and the device follows Eigen's
    own reduction orders (packet sums, gebp's split accumulators, the blocked LLT): every output bit equals the reference's."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fs_primitives.npz"))
    p = pkg.params_for_config("C4")
    med = pkg.Medium(p)
    for name in ("rand33", "se64", "se66"):
        vec, val = med.fs_linalg("eigh", g["eigh_%s_in" % name][None])
        assert np.array_equal(val[0], g["eigh_%s_val" % name]), name
        assert np.array_equal(vec[0], g["eigh_%s_vec" % name]), name
    for name in ("spd58", "se40_jitter", "se64"):
        t = med.fs_linalg("norm_transform", g["llt_%s_in" % name][None])
        assert np.array_equal(t[0], g["llt_%s_T" % name]), name
    pin = med.fs_linalg("pinv", g["eigh_se66_in"][None])
    assert np.array_equal(pin[0], g["pinv_se66"])
    # a batch larger than the resident grid, mixed sizes one at a time: the same bits for every copy
    stack = np.repeat(g["eigh_se64_in"][None], 1500, axis=0)
    vec, val = med.fs_linalg("eigh", stack)
    assert (vec == g["eigh_se64_vec"][None]).all() and (val == g["eigh_se64_val"][None]).all()

