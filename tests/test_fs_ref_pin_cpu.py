"""Pins the function-space path's linear-algebra / sampling layer and the fbm noise of the oracle BIT FOR BIT against the real
reference: tests/golden/ref_fs_primitives.npz was produced by the reference's own sources compiled in place (oracle/_ref:
Eigen::SelfAdjointEigenSolver / LLT of the vendored Eigen under the reference's flags, MultivariateNormalDistribution of
sampling/Gaussian.cpp with sampling/UniformPathSampler.hpp, the Eigen expression forms of pseudo_inverse and create_mvn_cond,
rand_truncated_normal, fbm / simplex3d / random3 of math/SdfFunctions.cpp; generator: tests/golden/make_golden.py).  Where
oracle/_ref itself is available (this container, or the prebuilt file on the GPU box) the same functions are also compared live
on fresh random inputs.  What stays unpinned is create_mvn_cond's covariance ASSEMBLY (GaussianProcess.cpp:664-690: its TU needs
Boost), i.e. which numbers enter these routines — not what the routines do with them."""
import ctypes
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fs_primitives.npz")
vp, ci, u64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_double
F = np.asfortranarray


def P(a):
    return a.ctypes.data_as(vp)


@pytest.fixture(scope="module")
def orc(ob):
    L = ob.oracle_lib()
    L.oracle_fs_eigh.argtypes = [ci, vp, vp]
    L.oracle_fs_norm_transform.argtypes = [ci, vp, vp]
    L.oracle_fs_pinv.argtypes = [ci, vp]
    L.oracle_fs_cond_forms.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp]
    L.oracle_fs_mvn_sample.argtypes = [ci, vp, vp, u64, ci, vp, vp, vp, vp]
    L.oracle_fs_rand_truncated_normal.restype = dbl
    L.oracle_fs_rand_truncated_normal.argtypes = [dbl, dbl, dbl, u64, vp]
    L.oracle_fbm.restype = dbl
    L.oracle_fbm.argtypes = [vp, ci]
    L.oracle_simplex3d.restype = ctypes.c_float
    L.oracle_simplex3d.argtypes = [vp]
    L.oracle_random3.argtypes = [vp, vp]
    L.oracle_bessel_k01.argtypes = [dbl, vp, vp]
    return L


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _eigh(orc, a):
    n = a.shape[0]
    A, w = F(a.copy()), np.zeros(n)
    orc.oracle_fs_eigh(n, P(A), P(w))
    return w, np.array(A)


def _norm_transform(orc, a):
    n = a.shape[0]
    T = np.zeros((n, n), order="F")
    orc.oracle_fs_norm_transform(n, P(F(a)), P(T))
    return np.array(T)


def test_eigen_solver_bit_equal_to_the_reference_eigen(orc, gold):
    for name in ("rand33", "se64", "se66"):
        w, v = _eigh(orc, gold["eigh_%s_in" % name])
        assert np.array_equal(w, gold["eigh_%s_val" % name]), name
        assert np.array_equal(v, gold["eigh_%s_vec" % name]), name
    # the C4 kind is numerically singular: its trailing eigenvalues are rounding noise, so equality above is a statement about
    # every rounding of the solver, not about well-conditioned quantities
    w = gold["eigh_se64_val"]
    assert (np.abs(w) < 1e-14 * w.max()).sum() > 20


def test_llt_and_norm_transform_bit_equal(orc, gold):
    for name, info in (("spd58", 0), ("se40_jitter", 0), ("se64", 1)):
        assert int(gold["llt_%s_info" % name]) == info           # Eigen::Success = 0, NumericalIssue = 1 (-> the eigen square root)
        t = _norm_transform(orc, gold["llt_%s_in" % name])
        assert np.array_equal(t, gold["llt_%s_T" % name]), name
    t = gold["llt_spd58_T"]
    assert np.abs(np.triu(t, 1)).max() == 0 and np.abs(np.triu(gold["llt_se64_T"], 1)).max() > 0


def test_pseudo_inverse_and_conditioning_products_bit_equal(orc, gold):
    a = F(gold["eigh_se66_in"].copy())
    orc.oracle_fs_pinv(66, P(a))
    assert np.array_equal(np.array(a), gold["pinv_se66"])
    nc, n = 66, 64
    m, c = np.zeros(n), np.zeros((n, n), order="F")
    orc.oracle_fs_cond_forms(nc, n, P(F(gold["pinv_se66"])), P(F(gold["cond_s12"])), P(F(gold["cond_s22"])), P(gold["cond_resid"]), P(gold["cond_mean"]), P(m), P(c))
    assert np.array_equal(m, gold["cond_mean_out"]) and np.array_equal(np.array(c), gold["cond_cov_out"])


def test_mvn_sample_with_constraints_and_truncated_normal_bit_equal(orc, gold):
    mats = {"se64": gold["eigh_se64_in"], "spd58": gold["llt_spd58_in"], "se40_jitter": gold["llt_se40_jitter_in"]}
    cidx, cmm = np.ascontiguousarray(gold["mvn_constraint_idx"]), np.ascontiguousarray(gold["mvn_constraint_minmax"])
    for name, st in (("se64", 0x1234567), ("spd58", 77), ("se40_jitter", 2 ** 40 + 5)):
        a = mats[name]
        n = a.shape[0]
        mu = np.linspace(-0.02, 0.03, n)
        for ncon in (0, 2):
            o, so = np.zeros(n), u64()
            orc.oracle_fs_mvn_sample(n, P(mu), P(F(a)), st, ncon, P(cidx), P(cmm), P(o), ctypes.byref(so))
            assert np.array_equal(o, gold["mvn_%s_c%d" % (name, ncon)]), (name, ncon)
            assert so.value == int(gold["mvn_%s_c%d_state" % (name, ncon)]), (name, ncon)       # same number of draws consumed
    tin, tst = gold["truncnorm_in"], gold["truncnorm_state"]
    for i in range(len(tin)):
        so = u64()
        r = orc.oracle_fs_rand_truncated_normal(tin[i, 0], tin[i, 1], tin[i, 2], int(tst[i]), ctypes.byref(so))
        assert r == gold["truncnorm_out"][i] and so.value == int(gold["truncnorm_state_out"][i]), i


def test_fbm_simplex_random3_bit_equal(orc, gold):
    uv = np.ascontiguousarray(gold["fbm_in"])
    assert np.array_equal(np.array([orc.oracle_fbm(P(u), 2) for u in uv]), gold["fbm_oct2"])
    assert np.array_equal(np.array([orc.oracle_fbm(P(u), 10) for u in uv]), gold["fbm_oct10"])
    pts = np.ascontiguousarray(gold["simplex_in"])
    assert np.array_equal(np.array([orc.oracle_simplex3d(P(q)) for q in pts], dtype=np.float32), gold["simplex_out"])
    cs, got = np.ascontiguousarray(gold["random3_in"]), np.zeros((512, 3), dtype=np.float32)
    for i in range(512):
        orc.oracle_random3(P(cs[i]), P(got[i]))
    assert np.array_equal(got, gold["random3_out"])
    assert 0.2 < gold["fbm_oct10"].mean() < 0.8


def test_live_against_oracle_ref_on_fresh_inputs(orc, ob):
    """the same pins on inputs no fixture has seen (needs oracle/_ref: skipped where it is absent)"""
    ref = ob.ref_lib()
    if ref is None or not hasattr(ref, "ref_fs_eigh"):
        pytest.skip("oracle/_ref not built and /root/reference absent")
    ref.ref_fs_eigh.argtypes = [ci, vp, vp, vp]
    ref.ref_mvn_norm_transform.argtypes = [ci, vp, vp, vp]
    ref.ref_fs_pinv_forms.argtypes = [ci, vp, vp]
    ref.ref_fbm.restype = dbl
    ref.ref_fbm.argtypes = [vp, ci]
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    for it in range(60):
        n = int(rng.integers(1, 67))
        x = np.sort(rng.uniform(0, rng.uniform(0.2, 3.0), n))
        a = 0.01 * np.exp(-(x[:, None] - x[None, :]) ** 2 / (2 * 0.05 ** 2)) + np.eye(n) * (10 ** rng.uniform(-19, -14) if it % 2 else 0.0)
        V, w = np.zeros((n, n), order="F"), np.zeros(n)
        assert ref.ref_fs_eigh(n, P(F(a)), P(V), P(w)) == 0
        w2, v2 = _eigh(orc, a)
        assert np.array_equal(w, w2) and np.array_equal(np.array(V), v2), n
        T = np.zeros((n, n), order="F")
        ref.ref_mvn_norm_transform(n, P(np.zeros(n)), P(F(a)), P(T))
        assert np.array_equal(np.array(T), _norm_transform(orc, a)), n
        o = np.zeros((n, n), order="F")
        ref.ref_fs_pinv_forms(n, P(F(a)), P(o))
        b = F(a.copy())
        orc.oracle_fs_pinv(n, P(b))
        assert np.array_equal(np.array(o), np.array(b)), n
    for u in rng.uniform(-5, 5, (300, 3)):
        assert ref.ref_fbm(P(u), 10) == orc.oracle_fbm(P(u), 10)


def test_bessel_k0_k1_for_matern_three_halves(orc):
    """Matern v = 3/2 (GPF.cpp:1053-1056, 1071-1074) calls boost::math::cyl_bessel_k; Boost is neither vendored nor installed, so
    the kernel uses its own K0 / K1 — PARITY UNPINNED VS BOOST.  Checked against the Wronskian I0 K1 + I1 K0 = 1/x, the
    recurrence-free identity K1 = -K0', and scipy, to 1e-14 relative over the range the kernel uses (0 < x <= 3.7) and beyond."""
    from scipy import special
    k0, k1 = dbl(), dbl()
    xs = np.concatenate([np.logspace(-9, np.log10(2.0), 500), np.linspace(2.0, 50.0, 1200), [2.0 - 1e-12, 2.0 + 1e-12]])
    e0 = e1 = ew = 0.0
    for x in xs:
        orc.oracle_bessel_k01(float(x), ctypes.byref(k0), ctypes.byref(k1))
        e0 = max(e0, abs(k0.value - special.k0(x)) / special.k0(x))
        e1 = max(e1, abs(k1.value - special.k1(x)) / special.k1(x))
        if x < 30:
            ew = max(ew, abs(special.i0(x) * k1.value + special.i1(x) * k0.value - 1 / x) * x)
    assert e0 < 1e-14 and e1 < 1e-14 and ew < 1e-14, (e0, e1, ew)
    h = 1e-5                                                         # K0' = -K1 (central difference, 1e-9 accurate)
    for x in (0.3, 1.0, 1.99, 2.01, 3.5):
        a, b, c = dbl(), dbl(), dbl()
        orc.oracle_bessel_k01(x + h, ctypes.byref(a), ctypes.byref(c))
        orc.oracle_bessel_k01(x - h, ctypes.byref(b), ctypes.byref(c))
        orc.oracle_bessel_k01(x, ctypes.byref(k0), ctypes.byref(k1))
        assert abs((a.value - b.value) / (2 * h) + k1.value) < 1e-8 * k1.value
