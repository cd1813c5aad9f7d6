"""TEST INFRASTRUCTURE: ctypes wrapper of oracle/liboracle.so (the CPU restatement) and of
oracle/_ref/libgpis_ref.so (the real reference's primitives compiled in place).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import _gpis_pkg  # noqa: E402

_T = _gpis_pkg.load_package()
PARAMS, RAY_IN, SEG_OUT, COND_COEFF, QUERY, NEE_QUERY, DERIVED, SCENE_S, SURFACE_S = (
    _T.PARAMS, _T.RAY_IN, _T.SEG_OUT, _T.COND_COEFF, _T.QUERY, _T.NEE_QUERY, _T.DERIVED, _T.SCENE_S, _T.SURFACE_S)

ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libgpis_ref.so")


def build(force=False):
    """Compile the restatement (and, where /root/reference exists, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-f", os.path.join(_HERE, "Makefile")] + (["-B"] if force else []) + [ORACLE_SO], stdout=sys.stderr)      # make knows the dependencies
    # stdout belongs to the caller (bench.py prints exactly one JSON line there)
    subprocess.check_call(["make", "-s", "-f", os.path.join(_HERE, "Makefile"), "ref"], stdout=sys.stderr)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


_vp, _sz, _i32, _u32, _u64, _f32 = (ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32,
                                    ctypes.c_uint64, ctypes.c_float)


class Oracle:
    def __init__(self, params, threads=1):
        build()
        self.lib = ctypes.CDLL(ORACLE_SO)
        L = self.lib
        L.oracle_create.argtypes = [_vp, ctypes.POINTER(_vp)]
        L.oracle_destroy.argtypes = [_vp]
        L.oracle_set_threads.argtypes = [_vp, _i32]
        L.oracle_set_variance_grid.argtypes = [_vp, _vp, _vp]
        L.oracle_get_derived.argtypes = [_vp, _vp]
        L.oracle_sample_distance_batch.argtypes = [_vp, _sz, _vp, _vp, _vp]
        L.oracle_transmittance_batch.argtypes = [_vp, _sz, _vp, _vp]
        L.oracle_eval_value_batch.argtypes = [_vp, _sz, _vp, _vp, _vp]
        L.oracle_eval_gradient_batch.argtypes = [_vp, _sz, _vp, _vp]
        L.oracle_conditioning_batch.argtypes = [_vp, _sz, _vp, _vp, _vp, _vp]
        L.oracle_nee_pdf_batch.argtypes = [_vp, _sz, _vp, _vp]
        L.oracle_nee_grad_batch.argtypes = [_vp, _sz, _vp, _vp]
        L.oracle_mean_color_emission.argtypes = [_vp, _sz, _vp, _vp, _vp]
        L.oracle_get_counters.argtypes = [_vp, _vp, _vp]
        L.oracle_reset_counters.argtypes = [_vp]
        L.oracle_render_scene_s.argtypes = [_vp, _vp, _vp, _vp]
        L.oracle_scene_s_primary.argtypes = [_vp, _u32, _u32, _u32, _vp, _vp]
        L.oracle_render_scene_s_paths.argtypes = [_vp, _vp, _i32, _f32, _vp]
        L.oracle_render_scene_s_nee.argtypes = [_vp, _vp, _vp, _vp]
        self.params = _T.as_params(params)
        h = _vp()
        st = L.oracle_create(_p(self.params), ctypes.byref(h))
        if st != 0:
            raise ValueError("oracle_create failed (%d)" % st)
        self.h = h
        L.oracle_set_threads(h, int(threads))

    def __del__(self):
        try:
            if self.h:
                self.lib.oracle_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_variance_grid(self, voxels, world_to_index, interpolate="linear", origin=(0, 0, 0)):
        d, v = _T.variance_grid_desc(voxels, world_to_index, interpolate, origin)
        st = self.lib.oracle_set_variance_grid(self.h, d.ctypes.data_as(_vp), v.ctypes.data_as(_vp))
        if st != 0:
            raise ValueError("oracle_set_variance_grid failed (%d)" % st)

    def grid_unscaled_variance(self, points):
        p = np.ascontiguousarray(points, dtype=np.float64)
        out = np.empty(len(p), dtype=np.float32)
        self.lib.oracle_grid_unscaled_variance.argtypes = [_vp, _sz, _vp, _vp]
        self.lib.oracle_grid_unscaled_variance(self.h, len(p), p.ctypes.data_as(_vp), out.ctypes.data_as(_vp))
        return out

    def set_threads(self, n):
        self.lib.oracle_set_threads(self.h, int(n))

    def derived(self):
        d = np.zeros((), dtype=DERIVED)
        self.lib.oracle_get_derived(self.h, _p(d))
        return d

    def sample_distance(self, rays, want_coeff=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_IN)
        out = np.zeros(rays.shape[0], dtype=SEG_OUT)
        coeff = np.zeros(rays.shape[0], dtype=COND_COEFF) if want_coeff else None
        assert self.lib.oracle_sample_distance_batch(self.h, rays.shape[0], _p(rays), _p(out), _p(coeff)) == 0
        return (out, coeff) if want_coeff else out

    def transmittance(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_IN)
        vis = np.zeros(rays.shape[0], dtype=np.uint8)
        assert self.lib.oracle_transmittance_batch(self.h, rays.shape[0], _p(rays), _p(vis)) == 0
        return vis

    def eval_value(self, q):
        q = np.ascontiguousarray(q, dtype=QUERY)
        val = np.zeros(q.shape[0], dtype=np.float32)
        gid = np.zeros(q.shape[0], dtype=np.int32)
        assert self.lib.oracle_eval_value_batch(self.h, q.shape[0], _p(q), _p(val), _p(gid)) == 0
        return val, gid

    def eval_gradient(self, q):
        q = np.ascontiguousarray(q, dtype=QUERY)
        g = np.zeros((q.shape[0], 3), dtype=np.float32)
        assert self.lib.oracle_eval_gradient_batch(self.h, q.shape[0], _p(q), _p(g)) == 0
        return g

    def conditioning(self, q, target_val, target_grad):
        q = np.ascontiguousarray(q, dtype=QUERY)
        tv = np.ascontiguousarray(target_val, dtype=np.float32)
        tg = np.ascontiguousarray(target_grad, dtype=np.float32)
        co = np.zeros(q.shape[0], dtype=COND_COEFF)
        assert self.lib.oracle_conditioning_batch(self.h, q.shape[0], _p(q), _p(tv), _p(tg), _p(co)) == 0
        return co

    def nee_pdf(self, q):
        q = np.ascontiguousarray(q, dtype=NEE_QUERY)
        out = np.zeros(q.shape[0], dtype=np.float32)
        assert self.lib.oracle_nee_pdf_batch(self.h, q.shape[0], _p(q), _p(out)) == 0
        return out

    def nee_grad(self, q):
        q = np.ascontiguousarray(q, dtype=NEE_QUERY)
        out = np.zeros((q.shape[0], 3), dtype=np.float32)
        assert self.lib.oracle_nee_grad_batch(self.h, q.shape[0], _p(q), _p(out)) == 0
        return out

    # ---- function-space comparison path: states are updated in place and returned
    def fs_sample_distance(self, rays, states):
        rays = np.ascontiguousarray(rays, dtype=RAY_IN)
        states = np.ascontiguousarray(states, dtype=_T.FS_STATE).copy()
        out = np.zeros(rays.shape[0], dtype=SEG_OUT)
        rc = self.lib.oracle_fs_sample_distance_batch(self.h, rays.shape[0], _p(rays), _p(states), _p(out))
        if rc != 0:
            raise RuntimeError("oracle_fs_sample_distance_batch failed: %d" % rc)
        return out, states

    def fs_transmittance(self, rays, states):
        rays = np.ascontiguousarray(rays, dtype=RAY_IN)
        states = np.ascontiguousarray(states, dtype=_T.FS_STATE).copy()
        vis = np.zeros(rays.shape[0], dtype=np.uint8)
        rc = self.lib.oracle_fs_transmittance_batch(self.h, rays.shape[0], _p(rays), _p(states), _p(vis))
        if rc != 0:
            raise RuntimeError("oracle_fs_transmittance_batch failed: %d" % rc)
        return vis, states

    def fs_eigh(self, a):
        a = np.asarray(a, dtype=np.float64)
        n = a.shape[0]
        m = np.asfortranarray(a).copy(order="F")
        w = np.zeros(n, dtype=np.float64)
        self.lib.oracle_fs_eigh(n, m.ctypes.data_as(ctypes.c_void_p), _p(w))
        return w, m

    def fs_norm_transform(self, s):
        s = np.asfortranarray(np.asarray(s, dtype=np.float64))
        t = np.zeros_like(s, order="F")
        self.lib.oracle_fs_norm_transform(s.shape[0], s.ctypes.data_as(ctypes.c_void_p), t.ctypes.data_as(ctypes.c_void_p))
        return t

    def fs_cov(self, da, db, a, b, dir_a, dir_b):
        f = self.lib.oracle_fs_cov
        f.restype = ctypes.c_double
        v = [np.ascontiguousarray(x, dtype=np.float64) for x in (a, b, dir_a, dir_b)]
        return f(self.h, int(da), int(db), *[_p(x) for x in v])

    def mean_color_emission(self, points):
        p = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        col = np.zeros((p.shape[0], 3), dtype=np.float32)
        emi = np.zeros((p.shape[0], 3), dtype=np.float32)
        assert self.lib.oracle_mean_color_emission(self.h, p.shape[0], _p(p), _p(col), _p(emi)) == 0
        return col, emi

    def counters(self):
        e, s = _u64(), _u64()
        self.lib.oracle_get_counters(self.h, ctypes.byref(e), ctypes.byref(s))
        return e.value, s.value

    def reset_counters(self):
        self.lib.oracle_reset_counters(self.h)

    def render_scene_s(self, scene, want_hits=False):
        scene = np.array(scene, dtype=SCENE_S)
        rad = np.zeros((int(scene["height"]), int(scene["width"])), dtype=np.float32)
        hits = np.zeros_like(rad, dtype=np.uint32) if want_hits else None
        assert self.lib.oracle_render_scene_s(self.h, _p(scene), _p(rad), _p(hits)) == 0
        return (rad, hits) if want_hits else rad

    def render_scene_s_nee(self, scene, surface):
        scene = np.array(scene, dtype=SCENE_S)
        surface = np.array(surface, dtype=SURFACE_S)
        rad = np.zeros((int(scene["height"]), int(scene["width"])), dtype=np.float32)
        assert self.lib.oracle_render_scene_s_nee(self.h, _p(scene), _p(surface), _p(rad)) == 0
        return rad

    def render_scene_s_paths(self, scene, max_bounces, albedo):
        scene = np.array(scene, dtype=SCENE_S)
        rad = np.zeros((int(scene["height"]), int(scene["width"])), dtype=np.float32)
        assert self.lib.oracle_render_scene_s_paths(self.h, _p(scene), int(max_bounces), _f32(albedo), _p(rad)) == 0
        return rad

    def scene_s_primary(self, scene, x, y, spp):
        scene = np.array(scene, dtype=SCENE_S)
        ray = np.zeros((), dtype=RAY_IN)
        us = _f32()
        hit = self.lib.oracle_scene_s_primary(_p(scene), int(x), int(y), int(spp), _p(ray), ctypes.byref(us))
        return hit, ray, us.value


def oracle_lib():
    build()
    L = ctypes.CDLL(ORACLE_SO)
    L.oracle_normalized_uint.restype = _f32
    L.oracle_eig_dist2_ab.restype = _f32
    L.oracle_eig_dot_col.restype = _f32
    return L


def default_scene_s(width, height, spp):
    build()
    L = ctypes.CDLL(ORACLE_SO)
    s = np.zeros((), dtype=SCENE_S)
    L.oracle_default_scene_s(_p(s), _u32(width), _u32(height), _u32(spp))
    return s


def xxhash32(words):
    L = oracle_lib()
    words = np.ascontiguousarray(words, dtype=np.uint32)
    n, arity = words.shape
    out = np.zeros(n, dtype=np.uint32)
    assert L.oracle_xxhash32_batch(_sz(n), arity, _p(words), _p(out)) == 0
    return out


def pcg32_stream(states, count):
    L = oracle_lib()
    states = np.ascontiguousarray(states, dtype=np.uint64)
    out = np.zeros((states.shape[0], count), dtype=np.uint32)
    assert L.oracle_pcg32_stream_batch(_sz(states.shape[0]), _p(states), _u32(count), _p(out)) == 0
    return out


def ref_lib():
    """The reference's own primitives (None when oracle/_ref has not been built, e.g. on the GPU box
    if the prebuilt file did not travel)."""
    if not os.path.exists(REF_SO):
        try:
            build()
        except Exception:
            pass
    if not os.path.exists(REF_SO):
        return None
    L = ctypes.CDLL(REF_SO)
    for name in ("ref_xxhash32_1", "ref_xxhash32_2", "ref_xxhash32_3", "ref_xxhash32_4"):
        getattr(L, name).restype = _u32
        getattr(L, name).argtypes = [_u32] * int(name[-1])
    L.ref_normalized_uint.restype = _f32
    L.ref_normalized_uint.argtypes = [_u32]
    L.ref_bernoulli.restype = _f32
    L.ref_bernoulli.argtypes = [_f32]
    L.ref_eig_dist2_ab.restype = _f32
    L.ref_eig_dot_col.restype = _f32
    L.ref_vec3_length_sq.restype = _f32
    L.ref_vec3_dot.restype = _f32
    L.ref_pi_float.restype = _f32
    return L
