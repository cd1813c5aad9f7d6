#!/usr/bin/env python
"""Regenerates the golden fixtures in this directory.  Run in the build container:

    python tests/golden/make_golden.py

Two kinds of vectors (the reference ships no tests, fixtures or scenes for this path — SURVEY.md §4):

  ref_primitives.npz   produced by the REAL reference compiled in place (oracle/_ref: its own
                       MathUtil.hpp / UniformSampler.hpp / TangentFrame.hpp / Gaussian.cpp):
                       xxhash32 x4, PCG32 streams, the per-impulse draw order, Duff ONB, Box–Muller.
  ref_fs_primitives.npz  produced by the REAL reference compiled in place: Eigen's SelfAdjointEigenSolver / LLT, the
                       MultivariateNormalDistribution of sampling/Gaussian.cpp, pseudo-inverse / conditioning product forms,
                       rand_truncated_normal, fbm / simplex3d / random3 of math/SdfFunctions.cpp.
  reference_kat.json   the outputs the reference's evaluator printed in this image.
  oracle_*.npz         inputs + outputs of the oracle (CPU restatement) for every mode of the path.
                       They pin the HIP path and guard the oracle against regressions; they are
                       reference-derived only through the two files above ("parity unpinned"
                       beyond them).
"""
import ctypes
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gpis_pkg  # noqa: E402
import oracle_bindings as ob  # noqa: E402
from gpu_util import scene_rays, shadow_rays_from  # noqa: E402

pkg = _gpis_pkg.load_package()
f32 = np.float32


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def ref_primitives():
    ref = ob.ref_lib()
    assert ref is not None, "oracle/_ref must be built (needs /root/reference)"
    rng = np.random.default_rng(20261004)
    out = {}
    for arity in (1, 2, 3, 4):
        words = rng.integers(0, 2 ** 32, size=(512, arity), dtype=np.uint64).astype(np.uint32)
        words[:4] = [[0] * arity, [1] * arity, [0xFFFFFFFF] * arity, list(range(1, arity + 1))]
        fn = getattr(ref, "ref_xxhash32_%d" % arity)
        out["hash%d_in" % arity] = words
        out["hash%d_out" % arity] = np.array([fn(*[int(w) for w in row]) for row in words], dtype=np.uint32)
    states = rng.integers(0, 2 ** 63, size=64, dtype=np.uint64)
    states[:4] = [0, 1, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF]
    streams = np.zeros((64, 96), dtype=np.uint32)
    draws = np.zeros((64, 4 * 16), dtype=f32)
    normals = np.zeros((64, 8))
    for i, s in enumerate(states):
        ref.ref_pcg32_stream(ctypes.c_uint64(int(s)), 96, P(streams[i]))
        ref.ref_cell3d_draws(ctypes.c_uint64(int(s)), 16, P(draws[i]))
        ref.ref_sample_standard_normal2(ctypes.c_uint64(int(s)), 4, P(normals[i]))
    out.update(pcg_state=states, pcg_stream=streams, cell3d_draws=draws, box_muller=normals)
    ns = rng.standard_normal((128, 3)).astype(f32)
    ns[:4] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [0.3, -0.4, -0.0]]
    frames = np.zeros((128, 9), dtype=f32)
    for i, n in enumerate(ns):
        ref.ref_tangent_frame(P(n), P(frames[i]))
    out.update(frame_in=ns, frame_out=frames)
    # Fresnel.hpp:102-123 and SampleWarp.hpp:131-134, 189-192 (the conductor NEE estimator's arithmetic);
    # generated after everything above so the earlier vectors keep their random draws
    for n, k in (("ref_conductor_reflectance", 3), ("ref_power_heuristic", 2), ("ref_uniform_spherical_cap_pdf", 1)):
        getattr(ref, n).restype = ctypes.c_float
        getattr(ref, n).argtypes = [ctypes.c_float] * k
    fr_in = np.column_stack([rng.choice([0.2, 1.5, 0.0, 2.9, 0.05], 256), rng.choice([3.9, 0.0, 3.1, 7.0], 256), rng.uniform(-1, 1, 256)]).astype(f32)
    fr_in[:3] = [[0, 0, 0.5], [0.2, 3.9, 1.0], [0.2, 3.9, 0.0]]
    out.update(fresnel_in=fr_in, fresnel_out=np.array([ref.ref_conductor_reflectance(*map(float, r)) for r in fr_in], dtype=f32))
    ph_in = rng.uniform(0, 50, (256, 2)).astype(f32)
    out.update(power_heuristic_in=ph_in, power_heuristic_out=np.array([ref.ref_power_heuristic(*map(float, r)) for r in ph_in], dtype=f32))
    cap_in = rng.uniform(-0.99, 0.9999, 128).astype(f32)
    out.update(cap_pdf_in=cap_in, cap_pdf_out=np.array([ref.ref_uniform_spherical_cap_pdf(float(c)) for c in cap_in], dtype=f32))
    np.savez_compressed(os.path.join(HERE, "ref_primitives.npz"), **out)


def _se(x, y=None, l=0.05, s2=0.01):
    y = x if y is None else y
    return s2 * np.exp(-(x[:, None] - y[None, :]) ** 2 / (2 * l * l))


def ref_fs_primitives():
    """ref_fs_primitives.npz — the function-space path's linear-algebra / sampling layer and the fbm noise, produced by the REAL
    reference compiled in place (oracle/_ref): Eigen::SelfAdjointEigenSolver / LLT of the vendored Eigen under the reference's
    flags, MultivariateNormalDistribution (sampling/Gaussian.cpp:121-232) with sampling/UniformPathSampler.hpp, the Eigen
    expression forms of pseudo_inverse and create_mvn_cond (GaussianProcess.cpp:645-662, 693, 736-745), rand_truncated_normal,
    and fbm / simplex3d / random3 (math/SdfFunctions.cpp:199-296).  Matrices are stored as numpy arrays a[i, j]."""
    ref = ob.ref_lib()
    assert ref is not None, "oracle/_ref must be built (needs /root/reference)"
    vp, ci, u64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_double
    ref.ref_fs_eigh.argtypes = [ci, vp, vp, vp]
    ref.ref_fs_llt.argtypes = [ci, vp, vp]
    ref.ref_mvn_norm_transform.argtypes = [ci, vp, vp, vp]
    ref.ref_fs_pinv_forms.argtypes = [ci, vp, vp]
    ref.ref_fs_cond_forms.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp]
    ref.ref_mvn_sample.argtypes = [ci, vp, vp, u64, ci, vp, vp, ci, vp, vp]
    ref.ref_rand_truncated_normal.restype = dbl
    ref.ref_rand_truncated_normal.argtypes = [dbl, dbl, dbl, u64, vp]
    ref.ref_fbm.restype = dbl
    ref.ref_fbm.argtypes = [vp, ci]
    ref.ref_simplex3d.restype = ctypes.c_float
    ref.ref_simplex3d.argtypes = [vp]
    ref.ref_random3.argtypes = [vp, vp]
    rng = np.random.default_rng(20261005)
    F = np.asfortranarray
    out = {}
    # eigen-solver: a random symmetric matrix and the C4 kind (squared exponential on a fine grid: numerical rank ~20), n = 66 = the
    # Global context's largest system; a clustered grid (a crossing point between two of its own conditioning points)
    x66 = np.sort(np.concatenate([np.linspace(0, 0.64, 64), [0.3001, 0.30015]]))
    cases = {"rand33": (lambda a: a + a.T)(rng.standard_normal((33, 33))), "se64": _se(np.linspace(0, 0.64, 64)), "se66": _se(x66)}
    for name, a in cases.items():
        n = a.shape[0]
        A, V, w = F(a), np.zeros((n, n), order="F"), np.zeros(n)
        assert ref.ref_fs_eigh(n, P(A), P(V), P(w)) == 0
        out.update({"eigh_%s_in" % name: a, "eigh_%s_vec" % name: np.array(V), "eigh_%s_val" % name: w})
    # LLT: well conditioned (success, blocked path with a 2-row tail), barely positive definite, and singular (failure -> eigen square root)
    a = rng.standard_normal((58, 58))
    llt_cases = {"spd58": a @ a.T + 58 * np.eye(58), "se40_jitter": _se(np.linspace(0, 2.0, 40)) + 1e-12 * np.eye(40), "se64": cases["se64"]}
    for name, a in llt_cases.items():
        n = a.shape[0]
        S, L_, T = F(a), np.zeros((n, n), order="F"), np.zeros((n, n), order="F")
        info = ref.ref_fs_llt(n, P(S), P(L_))
        ref.ref_mvn_norm_transform(n, P(np.zeros(n)), P(S), P(T))
        out.update({"llt_%s_in" % name: a, "llt_%s_info" % name: np.int32(info), "llt_%s_T" % name: np.array(T)})
    # pseudo-inverse (66: the last two rows take gebp's split accumulators) and the conditioning products
    A, o = F(cases["se66"]), np.zeros((66, 66), order="F")
    ref.ref_fs_pinv_forms(66, P(A), P(o))
    out["pinv_se66"] = np.array(o)
    nc, n = 66, 64
    xc, xs = x66, np.linspace(0.64, 1.28, 64)
    pin, s12, s22 = F(np.array(o)), F(_se(xc, xs)), F(_se(xs))
    resid, mu = rng.standard_normal(nc) * 0.05, rng.standard_normal(n) * 0.05
    m1, c1 = np.zeros(n), np.zeros((n, n), order="F")
    ref.ref_fs_cond_forms(nc, n, P(pin), P(s12), P(s22), P(resid), P(mu), P(m1), P(c1))
    out.update(cond_s12=np.array(s12), cond_s22=np.array(s22), cond_resid=resid, cond_mean=mu, cond_mean_out=m1, cond_cov_out=np.array(c1))
    # MVN samples through the reference's own sampler, with and without constraints; the sampler state afterwards
    mv_in, mv_out = [], []
    for k, (name, st) in enumerate((("se64", 0x1234567), ("spd58", 77), ("se40_jitter", 2 ** 40 + 5))):
        a = cases.get(name, llt_cases.get(name))
        n = a.shape[0]
        mu = np.linspace(-0.02, 0.03, n)
        cidx, cmm = np.array([0, 3, n // 2, n // 2], dtype=np.int32), np.array([-0.3, 0.4, -50.0, 50.0], dtype=np.float32)
        for ncon in (0, 2):
            o, so = np.zeros(n), u64()
            ref.ref_mvn_sample(n, P(mu), P(F(a)), st, ncon, P(cidx), P(cmm), 1, P(o), ctypes.byref(so))
            out["mvn_%s_c%d" % (name, ncon)] = o
            out["mvn_%s_c%d_state" % (name, ncon)] = np.uint64(so.value)
    out.update(mvn_constraint_idx=cidx, mvn_constraint_minmax=cmm)
    tn_in = np.column_stack([rng.normal(size=64), np.abs(rng.normal(size=64)) + 0.01, rng.normal(size=64) * 2])
    tn_in[::3, 2] = tn_in[::3, 0]
    tn_state = rng.integers(1, 2 ** 63, size=64, dtype=np.uint64)
    tn_out, tn_so = np.zeros(64), np.zeros(64, dtype=np.uint64)
    for i in range(64):
        so = u64()
        tn_out[i] = ref.ref_rand_truncated_normal(tn_in[i, 0], tn_in[i, 1], tn_in[i, 2], int(tn_state[i]), ctypes.byref(so))
        tn_so[i] = so.value
    out.update(truncnorm_in=tn_in, truncnorm_state=tn_state, truncnorm_out=tn_out, truncnorm_state_out=tn_so)
    # fbm noise
    uv = rng.uniform(-4, 4, (512, 3))
    out.update(fbm_in=uv, fbm_oct2=np.array([ref.ref_fbm(P(u), 2) for u in uv]), fbm_oct10=np.array([ref.ref_fbm(P(u), 10) for u in uv]))
    pts = rng.uniform(-30, 30, (512, 3)).astype(f32)
    out.update(simplex_in=pts, simplex_out=np.array([ref.ref_simplex3d(P(q)) for q in pts], dtype=f32))
    cs = np.floor(rng.uniform(-60, 60, (512, 3))).astype(f32)
    r3 = np.zeros((512, 3), dtype=f32)
    for i in range(512):
        ref.ref_random3(P(cs[i]), P(r3[i]))
    out.update(random3_in=cs, random3_out=r3)
    np.savez_compressed(os.path.join(HERE, "ref_fs_primitives.npz"), **out)


def queries(n, seed, spread=1.4):
    rng = np.random.default_rng(seed)
    q = np.zeros(n, dtype=pkg.QUERY)
    q["p"] = rng.uniform(-spread, spread, (n, 3)).astype(f32)
    d = rng.standard_normal((n, 3))
    q["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
    q["t_segment"] = rng.uniform(0, 2, n).astype(f32)
    q["info_t"] = rng.uniform(0, 3, n).astype(f32)
    q["pixel"] = rng.integers(0, 1920, (n, 2))
    q["spp"] = rng.integers(0, 64, n)
    q["segment"] = rng.integers(0, 4, n)
    q["scene_seed"] = 0xBA5EBA11
    q["p"][0] = (0.9, 0.1, -0.2)
    q["p"][1] = (-1.2, -0.3, -0.9)
    q["p"][2] = (0.10606602, 0.21213204, -0.10606602)   # on cell faces of the world grid
    return q


CASES = {
    # name: (config, overrides)
    "C0_world": ("C0", {}),
    "C1_isoray": ("C1", {}),
    "C1_perpath_renewal_plus": ("C1", {"single_realization": 0, "correlation_context": pkg.CTX.RENEWAL_PLUS, "impulse_density": 12}),
    "C0_perpath_renewal": ("C0", {"single_realization": 0, "correlation_context": pkg.CTX.RENEWAL}),
    "C2_1d_mis": ("C2", {}),
    "C3_multires": ("C3", {"impulse_density": 16, "correlation_context": pkg.CTX.RENEWAL_PLUS}),
}


def params_of(case):
    cfg, over = CASES[case]
    p = pkg.params_for_config(cfg)
    for k, v in over.items():
        p[k] = v
    return p


def oracle_vectors():
    for case in CASES:
        params = params_of(case)
        orc = ob.Oracle(params, threads=8)
        q = queries(192, 100 + len(case))
        val, gid = orc.eval_value(q)
        grad = orc.eval_gradient(q)
        scene = ob.default_scene_s(96, 54, 1)
        rays, us = scene_rays(ob, orc, scene, step=5)
        seg, coeff = orc.sample_distance(rays, want_coeff=True)
        sh = shadow_rays_from(ob, scene, rays, us, seg)
        seg2, coeff2 = orc.sample_distance(sh, want_coeff=True)
        vis = orc.transmittance(sh)
        np.savez_compressed(os.path.join(HERE, "oracle_%s.npz" % case), params=params, q=q, value=val, gp_id=gid, grad=grad,
                            rays=rays, seg=seg, coeff=coeff, shadow=sh, seg2=seg2, coeff2=coeff2, vis=vis)
    # the C0 scene-S image: 64x64, 4 spp — radiance sums and per-pixel hit counts
    orc = ob.Oracle(pkg.params_for_config("C0"), threads=8)
    rad, hits = orc.render_scene_s(ob.default_scene_s(64, 64, 4), want_hits=True)
    # the same scene through the multi-bounce driver: 4 path bounces, albedo 0.8
    paths = orc.render_scene_s_paths(ob.default_scene_s(64, 64, 4), 4, 0.8)
    np.savez_compressed(os.path.join(HERE, "oracle_C0_image64.npz"), radiance_sum=rad, hits=hits, paths_radiance_sum=paths)
    # scene S with the specular NEE coupling on the C2 medium (1D sampling, MIS scheme): 48x48, 2 spp
    orc = ob.Oracle(pkg.params_for_config("C2"), threads=8)
    nee = orc.render_scene_s_nee(ob.default_scene_s(48, 48, 2), pkg.default_surface_s())
    np.savez_compressed(os.path.join(HERE, "image_C2_nee48.npz"), radiance_sum=nee, surface=pkg.default_surface_s())


def reference_kat():
    json.dump({
        "source": "SURVEY.md 8c 'Observed output': the reference's own evaluator compiled and run in this image",
        "config": "SE sigma=0.1 l=0.05 aniso=1, SphericalMean(0,1), ctx=none, seed=7, rho=8, single realization, world space, "
                  "pixelSampleSegment=(3,4,0,0)",
        "point": [0.9, 0.1, -0.2],
        "evaluateValue_9g": "-0.00919273123",
        "evaluateGradient_9g": ["-1.11124325", "1.23765218", "-2.80203676"],
        "xxhash32_Vec4u_1_2_3_4": 2694834884,
        "worldToLocal_diag_6g": "28.2843",
    }, open(os.path.join(HERE, "reference_kat.json"), "w"), indent=1)


if __name__ == "__main__":
    ob.build()
    ref_primitives()
    ref_fs_primitives()
    reference_kat()
    oracle_vectors()
    print(sorted(os.listdir(HERE)))
