"""ctypes / numpy mirror of include/gpis.h (ABI version 1).

Struct layouts are checked against the C sizes at import time of the library
(``gpis_abi_sizes``) so a drift between this file and the header fails loudly.
"""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class CTX:
    GLOBAL, RENEWAL_PLUS, RENEWAL, NONE = 0, 1, 2, 3


class SCHEME:
    UNI, NEE, MIS = 0, 1, 2


class MEAN_TYPE:
    HOMOGENEOUS, SPHERICAL, LINEAR = 0, 1, 2


MEAN = np.dtype([
    ("type", "<i4"), ("radius", "<f4"), ("offset", "<f4"), ("scale", "<f4"), ("min", "<f4"), ("_pad", "<i4"),
    ("center", "<f8", 3), ("dir", "<f8", 3),
], align=True)

RAMP = np.dtype([
    ("enabled", "<i4"), ("type", "<i4"),
    ("min", "<f8"), ("max", "<f8"), ("start", "<f8"), ("end", "<f8"),
    ("min2", "<f8"), ("max2", "<f8"), ("start2", "<f8"), ("end2", "<f8"),
], align=True)

PARAMS = np.dtype([
    ("abi_version", "<u4"),
    ("step_size", "<f4"), ("min_step", "<u4"), ("seed", "<u4"), ("impulse_density", "<f4"),
    ("single_realization", "<i4"), ("isotropic_3d_sampling", "<i4"), ("sampling_1d", "<i4"),
    ("scheme_1d", "<i4"), ("correlation_xy", "<i4"), ("surf_vol_phase_separate", "<i4"),
    ("surf_vol_phase_amp_thresh", "<f4"),
    ("correlation_context", "<i4"), ("max_bounces", "<i4"),
    ("sigma_a", "<f4", 3), ("sigma_s", "<f4", 3), ("density", "<f4"),
    ("sigma", "<f4"), ("length_scale", "<f4"), ("aniso", "<f4", 3), ("use_aniso_mtx", "<i4"),
    ("aniso_mtx", "<f4", 9), ("local_scale", "<f4"),
    ("nonstationary", "<i4"), ("multi_resolution_grid", "<i4"), ("ls_ramp_type", "<i4"), ("_pad0", "<i4"),
    ("ls_min", "<f8"), ("ls_max", "<f8"), ("ls_start", "<f8"), ("ls_end", "<f8"),
    ("mean", MEAN), ("has_mean_additional", "<i4"), ("_pad1", "<i4"), ("mean_additional", MEAN),
    ("ls_min2", "<f8"), ("ls_max2", "<f8"), ("ls_start2", "<f8"), ("ls_end2", "<f8"),
    ("var", RAMP), ("mean_color", RAMP), ("mean_emission", RAMP),
    ("kernel_type", "<i4"), ("matern_v", "<f4"), ("gabor_a_inv", "<f4"), ("gabor_f_inv", "<f4"), ("gabor_omega", "<f4", 3), ("_pad2", "<i4"),
    ("aniso_field", RAMP),
    ("fs_sample_points", "<i4"), ("_pad3", "<i4"), ("fs_step_size", "<f8"),
    ("grid_nonstationary", "<i4"), ("grid_surf_vol_amp_separate", "<i4"), ("grid_offset", "<f4"), ("grid_scale", "<f4"),
    ("grid_surf_vol_amp_thresh", "<f4"), ("grid_surf_amp_scale", "<f4"), ("grid_vol_amp_scale", "<f4"),
    ("grid_surf_ls_scale", "<f4"), ("grid_vol_ls_scale", "<f4"), ("_pad4", "<i4"),
], align=True)

VARIANCE_GRID = np.dtype([
    ("dims", "<i4", 3), ("interpolate", "<i4"), ("origin", "<i4", 3), ("_pad", "<i4"),
    ("bounds_min", "<f4", 3), ("bounds_max", "<f4", 3), ("inv_natural_transform", "<f4", 16),
], align=True)


def variance_grid_desc(voxels, world_to_index, interpolate="linear", origin=(0, 0, 0)):
    """gpis_variance_grid for a dense array voxels[k, j, i] (z, y, x — x fastest in memory) whose voxel (0, 0, 0) sits at the
    index-space coordinate `origin`; bounds = the whole array, as VdbGrid::bounds() is the box of the active voxels."""
    v = np.ascontiguousarray(voxels, dtype=np.float32)
    d = np.zeros((), dtype=VARIANCE_GRID)
    d["dims"] = (v.shape[2], v.shape[1], v.shape[0])
    d["interpolate"] = {"point": 0, "linear": 1}[interpolate]
    d["origin"] = origin
    d["bounds_min"] = origin
    d["bounds_max"] = (origin[0] + v.shape[2] - 1, origin[1] + v.shape[1] - 1, origin[2] + v.shape[0] - 1)
    d["inv_natural_transform"] = np.asarray(world_to_index, dtype=np.float32).reshape(16)
    return d, v

FS_MAX_POINTS, FS_MAX_CTX = 64, 66
FS_STATE = np.dtype([
    ("sampler_state", "<u8"), ("has_context", "<i4"), ("is_intersect", "<i4"), ("n_points", "<i4"), ("n_values", "<i4"),
    ("sampled_grad", "<f8", 3), ("points", "<f8", (FS_MAX_CTX, 3)), ("values", "<f8", FS_MAX_CTX), ("derivs", "<i4", FS_MAX_CTX),
], align=True)

RAY_IN = np.dtype([
    ("pos", "<f4", 3), ("dir", "<f4", 3), ("near_t", "<f4"), ("far_t", "<f4"),
    ("pixel", "<u4", 2), ("spp", "<u4"), ("segment", "<u4"),
    ("scene_seed", "<u4"), ("info_t", "<f4"), ("u_jitter", "<f4"), ("first_scatter", "<u4"),
    ("bounce", "<i4"), ("last_val", "<f4"), ("last_gp_id", "<i4"), ("_pad", "<i4"),
    ("last_aniso", "<f8", 3), ("_reserved", "<f8", 3),
], align=True)

SEG_OUT = np.dtype([
    ("t", "<f8"), ("aniso", "<f8", 3),
    ("sample_t", "<f4"), ("continued_t", "<f4"), ("weight", "<f4", 3), ("continued_weight", "<f4", 3),
    ("p", "<f4", 3), ("last_val", "<f4"), ("exited", "<i4"), ("ok", "<i4"), ("gp_id", "<i4"), ("scheme", "<i4"),
], align=True)

COND_COEFF = np.dtype([
    ("value_scale", "<f4"), ("gradient_scale", "<f4", 3), ("ray_origin", "<f4", 3), ("n_evals", "<u4"),
], align=True)

QUERY = np.dtype([
    ("p", "<f4", 3), ("dir", "<f4", 3), ("t_segment", "<f4"), ("info_t", "<f4"),
    ("pixel", "<u4", 2), ("spp", "<u4"), ("segment", "<u4"),
    ("scene_seed", "<u4"), ("_pad", "<u4", 3),
    ("coeff", COND_COEFF),
], align=True)

NEE_QUERY = np.dtype([
    ("ray_dir", "<f4", 3), ("normal", "<f4", 3), ("p", "<f4", 3), ("t_segment", "<f4"), ("info_t", "<f4"),
    ("pixel", "<u4", 2), ("spp", "<u4"), ("segment", "<u4"), ("scene_seed", "<u4"),
    ("coeff", COND_COEFF),
], align=True)

DERIVED = np.dtype([
    ("world_to_local", "<f4", 9), ("local_to_world", "<f4", 9),
    ("kernel_radius_world", "<f4"), ("kernel_radius_iso", "<f4"),
    ("norm3d_world", "<f4"), ("norm3d_iso", "<f4"), ("norm1d", "<f4"),
    ("impulses_per_cell", "<u4"), ("activate_conditioning", "<i4"), ("effective_scheme_1d", "<i4"),
    ("multi_resolution", "<i4"), ("fast_path", "<i4"),
], align=True)

SCENE_S = np.dtype([
    ("width", "<u4"), ("height", "<u4"), ("spp_begin", "<u4"), ("spp_count", "<u4"),
    ("scene_seed", "<u4"), ("tile_size", "<u4"),
    ("cam_pos", "<f4", 3), ("cam_fov_deg", "<f4"), ("bound_radius", "<f4"),
    ("light_dir", "<f4", 3), ("light_radiance", "<f4"),
    ("y_begin", "<u4"), ("y_count", "<u4"), ("shard_index", "<u4"), ("shard_count", "<u4"),
], align=True)

SURFACE_S = np.dtype([
    ("eta", "<f4"), ("k", "<f4"), ("albedo", "<f4"), ("cap_cos", "<f4"), ("cap_radiance", "<f4"), ("_pad", "<f4", 3),
], align=True)


def default_scene_s(width, height, spp):
    """gpis_default_scene_s (scene S of SURVEY.md 8d) without loading the library."""
    s = np.zeros((), dtype=SCENE_S)
    s["width"], s["height"], s["spp_begin"], s["spp_count"] = width, height, 0, spp
    s["scene_seed"], s["tile_size"] = 0xBA5EBA11, 16
    s["cam_pos"], s["cam_fov_deg"], s["bound_radius"] = (0.0, 0.0, 4.0), 35.0, 1.5
    s["light_dir"], s["light_radiance"] = (0.5, 0.7, 0.5), 1.0
    s["y_begin"], s["y_count"], s["shard_index"], s["shard_count"] = 0, height, 0, 1
    return s


GUIDE_INFO = np.dtype([("half_extent_cells", np.int32), ("points_per_cell", np.int32), ("bricks_total", np.uint64), ("bricks_allocated", np.uint64),
                       ("bricks_usable", np.uint64), ("bytes_samples", np.uint64), ("bytes_bounds", np.uint64), ("bytes_dense", np.uint64),
                       ("selfcheck_points_tabulated", np.uint64)], align=True)


def default_surface_s():
    """Copper-like single-channel conductor and a 12-degree cap light."""
    s = np.zeros((), dtype=SURFACE_S)
    s["eta"], s["k"], s["albedo"] = 0.2, 3.9, 1.0
    s["cap_cos"], s["cap_radiance"] = 0.9781476, 20.0
    return s


_EXPECTED_SIZES = {
    "gpis_params": PARAMS.itemsize, "gpis_mean": MEAN.itemsize, "gpis_ray_in": RAY_IN.itemsize,
    "gpis_seg_out": SEG_OUT.itemsize, "gpis_cond_coeff": COND_COEFF.itemsize, "gpis_query": QUERY.itemsize,
    "gpis_nee_query": NEE_QUERY.itemsize, "gpis_derived": DERIVED.itemsize, "gpis_scene_s": SCENE_S.itemsize,
    "gpis_surface_s": SURFACE_S.itemsize, "gpis_ramp": RAMP.itemsize,
    "gpis_fs_state": FS_STATE.itemsize, "gpis_guide_info": GUIDE_INFO.itemsize,
}
assert RAY_IN.itemsize == 128 and SEG_OUT.itemsize == 96 and COND_COEFF.itemsize == 32
assert QUERY.itemsize == 96 and NEE_QUERY.itemsize == 96


def default_params():
    """Reference defaults (SCNM.cpp:17-34, GPM.cpp:86-95, GPF.hpp:1729,1784) as a PARAMS record."""
    p = np.zeros((), dtype=PARAMS)
    p["abi_version"] = 3
    p["grid_scale"], p["grid_surf_vol_amp_thresh"] = 1.0, 1.0                       # GPF.hpp:2327-2335
    p["grid_surf_amp_scale"], p["grid_vol_amp_scale"], p["grid_surf_ls_scale"], p["grid_vol_ls_scale"] = 1.0, 1.0, 1.0, 1.0
    p["matern_v"], p["gabor_a_inv"], p["gabor_f_inv"], p["gabor_omega"] = 0.5, 1.0, 1.0, (1.0, 0.0, 0.0)
    p["fs_sample_points"], p["fs_step_size"] = 32, 0.0
    p["step_size"] = 0.01
    p["min_step"] = 8
    p["impulse_density"] = 3.0
    p["correlation_context"] = CTX.RENEWAL_PLUS
    p["max_bounces"] = 1024
    p["density"] = 1.0
    p["sigma"] = 1.0
    p["length_scale"] = 1.0
    p["aniso"] = 1.0
    p["aniso_mtx"] = np.eye(3, dtype=np.float32).ravel()
    p["local_scale"] = 3.0
    p["ls_min"], p["ls_max"], p["ls_start"], p["ls_end"] = 1.0, 500.0, 0.0, 1.0
    p["ls_min2"], p["ls_max2"], p["ls_start2"], p["ls_end2"] = 1.0, 500.0, 0.0, 1.0
    for key in ("var", "mean_color", "mean_emission", "aniso_field"):
        p[key]["min"], p[key]["max"], p[key]["start"], p[key]["end"] = 1.0, 500.0, 0.0, 1.0
        p[key]["min2"], p[key]["max2"], p[key]["start2"], p[key]["end2"] = 1.0, 500.0, 0.0, 1.0
    for key in ("mean", "mean_additional"):
        p[key]["type"] = MEAN_TYPE.SPHERICAL
        p[key]["radius"] = 1.0
        p[key]["scale"] = 1.0
        p[key]["min"] = -np.finfo(np.float32).max
        p[key]["dir"] = (1.0, 0.0, 0.0)
    return p


def as_params(rec):
    """A parameter record of an older ABI (the golden fixtures of round 1 hold 352-byte records) widened to the current
    layout: the fields it has are copied, the new ones keep the reference's defaults."""
    rec = np.asarray(rec)
    if rec.dtype == PARAMS:
        return np.array(rec, dtype=PARAMS)
    p = default_params()

    def copy(dst, src):
        for name in src.dtype.names:
            if name in dst.dtype.names and name != "abi_version":
                if src.dtype[name].names:
                    copy(dst[name], src[name])
                else:
                    dst[name] = src[name]
    copy(p, rec)
    return p


def params_for_config(name):
    """Scene-S medium parameters of the BASELINE.json configs."""
    p = default_params()
    p["sigma_a"] = 0.0
    p["sigma_s"] = 1.0
    p["density"] = 1.0
    p["step_size"] = 0.01
    p["min_step"] = 8
    p["seed"] = 7
    p["max_bounces"] = 1024
    p["sigma"] = 0.1
    p["length_scale"] = 0.05
    p["aniso"] = 1.0
    p["local_scale"] = 3.0
    p["mean"]["type"] = MEAN_TYPE.SPHERICAL
    p["mean"]["center"] = 0.0
    p["mean"]["radius"] = 1.0
    name = name.upper()
    if name == "C0":      # CPU-reference config: world space, rho=8, ctx none, single realization
        p["impulse_density"] = 8
        p["correlation_context"] = CTX.NONE
        p["single_realization"] = 1
        p["isotropic_3d_sampling"] = 0
    elif name == "C1":    # headline config: 3D isotropic, rho=32, renewal, single realization
        p["impulse_density"] = 32
        p["correlation_context"] = CTX.RENEWAL
        p["single_realization"] = 1
        p["isotropic_3d_sampling"] = 1
    elif name == "C2":    # 1D sampling, MIS, Renewal+, correlationXY, per-path realizations
        p["impulse_density"] = 32
        p["correlation_context"] = CTX.RENEWAL_PLUS
        p["single_realization"] = 0
        p["isotropic_3d_sampling"] = 1
        p["sampling_1d"] = 1
        p["scheme_1d"] = SCHEME.MIS
        p["correlation_xy"] = 1
    elif name == "C3":    # multi-resolution non-stationary, rho=64, ensemble
        p["impulse_density"] = 64
        p["correlation_context"] = CTX.RENEWAL
        p["single_realization"] = 0
        p["isotropic_3d_sampling"] = 1
        p["nonstationary"] = 1
        p["multi_resolution_grid"] = 1
        p["ls_ramp_type"] = 0
        p["ls_min"], p["ls_max"], p["ls_start"], p["ls_end"] = 0.5, 2.0, -1.0, 1.0
    elif name == "C4":    # FunctionSpaceGaussianProcessMedium comparison path: 64 sample points, global memory
        p["correlation_context"] = CTX.GLOBAL
        p["single_realization"] = 0
        p["fs_sample_points"] = 64
        p["fs_step_size"] = 0.0
    else:
        raise ValueError("unknown config %r" % name)
    return p


def library_path():
    # GPIS_LIBRARY: a differently tuned build of the same sources (tools/ experiments); the ABI check below still applies
    return os.environ.get("GPIS_LIBRARY") or os.path.join(_HERE, "csrc", "libgpis_hip.so")


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(ctypes.c_void_p)
    return ctypes.c_void_p(int(a))   # raw device address (e.g. torch tensor.data_ptr())


class GpisLib:
    """Thin ctypes view of libgpis_hip.so.  Raises if the library is absent (no CPU fallback)."""

    SYMBOLS = [
        "gpis_create", "gpis_destroy", "gpis_get_derived", "gpis_last_error", "gpis_default_params",
        "gpis_sample_distance_batch", "gpis_transmittance_batch", "gpis_eval_value_batch",
        "gpis_eval_gradient_batch", "gpis_conditioning_batch", "gpis_nee_pdf_batch", "gpis_nee_grad_batch",
        "gpis_xxhash32_batch", "gpis_pcg32_stream_batch", "gpis_mean_color_emission_batch", "gpis_mean_color_emission_host",
        "gpis_fs_sample_distance_batch", "gpis_fs_transmittance_batch", "gpis_fs_linalg_batch", "gpis_libm_batch", "gpis_sort_pairs_u32",
        "gpis_fs_sample_distance_host", "gpis_fs_transmittance_host",
        "gpis_sample_distance_host", "gpis_transmittance_host", "gpis_eval_value_host", "gpis_eval_gradient_host",
        "gpis_conditioning_host", "gpis_nee_pdf_host", "gpis_nee_grad_host", "gpis_alloc_host", "gpis_free_host",
        "gpis_get_counters", "gpis_reset_counters", "gpis_set_profiling", "gpis_get_kernel_profile",
        "gpis_set_batch_order", "gpis_set_option", "gpis_get_option", "gpis_build_guide", "gpis_drop_guide", "gpis_get_guide_info", "gpis_get_guide_steps", "gpis_guide_selfcheck", "gpis_guide_raycheck",
        "gpis_set_variance_grid", "gpis_default_scene_s", "gpis_reserve_scene_workspace", "gpis_render_scene_s", "gpis_render_scene_s_paths", "gpis_render_scene_s_nee",
    ]

    def __init__(self, path=None):
        path = path or library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                "HIP extension %s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        self.path = path
        # torch wheels bundle their own HIP runtime; when torch is going to be used in this process it
        # has to initialise first so that this library binds to the same runtime instance
        try:
            import torch
            torch.cuda.is_available()
        except Exception:
            pass
        self.lib = ctypes.CDLL(path)
        L = self.lib
        L.gpis_last_error.restype = ctypes.c_char_p
        for name in self.SYMBOLS:
            getattr(L, name)   # AttributeError if a declared symbol is not exported
        vp, sz, i32, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32
        L.gpis_create.argtypes = [vp, i32, ctypes.POINTER(vp)]
        L.gpis_destroy.argtypes = [vp]
        L.gpis_get_derived.argtypes = [vp, vp]
        L.gpis_default_params.argtypes = [vp]
        L.gpis_default_params.restype = None
        L.gpis_sample_distance_batch.argtypes = [vp, sz, vp, vp, vp, vp]
        L.gpis_transmittance_batch.argtypes = [vp, sz, vp, vp, vp]
        L.gpis_eval_value_batch.argtypes = [vp, sz, vp, vp, vp, vp]
        L.gpis_eval_gradient_batch.argtypes = [vp, sz, vp, vp, vp]
        L.gpis_conditioning_batch.argtypes = [vp, sz, vp, vp, vp, vp, vp]
        L.gpis_nee_pdf_batch.argtypes = [vp, sz, vp, vp, vp]
        L.gpis_nee_grad_batch.argtypes = [vp, sz, vp, vp, vp]
        L.gpis_mean_color_emission_batch.argtypes = [vp, sz, vp, vp, vp, vp]
        L.gpis_fs_linalg_batch.argtypes = [vp, i32, i32, sz, vp, vp, vp, vp]
        L.gpis_libm_batch.argtypes = [i32, sz, vp, vp, vp, vp, vp]
        L.gpis_sort_pairs_u32.argtypes = [sz, vp, vp, vp, vp, vp]
        L.gpis_fs_sample_distance_host.argtypes = [vp, sz, vp, vp, vp]
        L.gpis_fs_transmittance_host.argtypes = [vp, sz, vp, vp, vp]
        L.gpis_mean_color_emission_host.argtypes = [vp, sz, vp, vp, vp]
        L.gpis_xxhash32_batch.argtypes = [vp, sz, i32, vp, vp, vp]
        L.gpis_pcg32_stream_batch.argtypes = [vp, sz, vp, u32, vp, vp]
        L.gpis_sample_distance_host.argtypes = [vp, sz, vp, vp, vp]
        L.gpis_transmittance_host.argtypes = [vp, sz, vp, vp]
        L.gpis_eval_value_host.argtypes = [vp, sz, vp, vp, vp]
        L.gpis_eval_gradient_host.argtypes = [vp, sz, vp, vp]
        L.gpis_conditioning_host.argtypes = [vp, sz, vp, vp, vp, vp]
        L.gpis_nee_pdf_host.argtypes = [vp, sz, vp, vp]
        L.gpis_nee_grad_host.argtypes = [vp, sz, vp, vp]
        L.gpis_alloc_host.argtypes = [sz]
        L.gpis_alloc_host.restype = vp
        L.gpis_free_host.argtypes = [vp]
        L.gpis_free_host.restype = None
        L.gpis_get_counters.argtypes = [vp, vp, vp]
        L.gpis_reset_counters.argtypes = [vp]
        L.gpis_set_profiling.argtypes = [vp, i32]
        L.gpis_get_kernel_profile.argtypes = [vp, i32, vp, vp, vp, vp]
        L.gpis_set_option.argtypes = [vp, i32, ctypes.c_longlong]
        L.gpis_get_option.argtypes = [vp, i32, vp]
        L.gpis_build_guide.argtypes = [vp, i32, i32]
        L.gpis_get_guide_info.argtypes = [vp, vp]
        L.gpis_drop_guide.argtypes = [vp]
        L.gpis_get_guide_steps.argtypes = [vp, vp]
        L.gpis_guide_selfcheck.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp]
        L.gpis_guide_raycheck.argtypes = [vp, sz, vp, u32, vp, vp, vp]
        L.gpis_default_scene_s.argtypes = [vp, u32, u32, u32]
        L.gpis_reserve_scene_workspace.argtypes = [vp, vp]
        L.gpis_set_variance_grid.argtypes = [vp, vp, vp]
        L.gpis_default_scene_s.restype = None
        L.gpis_render_scene_s.argtypes = [vp, vp, vp, vp, vp]
        L.gpis_render_scene_s_paths.argtypes = [vp, vp, i32, ctypes.c_float, vp, vp]
        L.gpis_render_scene_s_nee.argtypes = [vp, vp, vp, vp, vp]
        if hasattr(L, "gpis_abi_sizes"):
            L.gpis_abi_sizes.restype = ctypes.c_char_p
            got = dict(kv.split("=") for kv in L.gpis_abi_sizes().decode().split(","))
            for k, v in _EXPECTED_SIZES.items():
                if int(got[k]) != v:
                    raise RuntimeError("ABI drift: %s is %s bytes in C, %d in bindings.py" % (k, got[k], v))

    def last_error(self):
        return self.lib.gpis_last_error().decode(errors="replace")

    def check(self, status, what):
        if status != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, status, self.last_error()))


_LIB = None


def load_library(path=None):
    global _LIB
    if _LIB is None or (path and _LIB.path != path):
        _LIB = GpisLib(path)
    return _LIB


LIBM_FNS = {"exp": 0, "log": 1, "logf": 2, "sin": 3, "cos": 4, "sincos": 5, "pow": 6, "sincosf": 7}


def libm_eval(fn, x, y=None, device=0, lib=None):
    """Test surface: the device's restatement of the host libm (csrc/gpis_libm.hpp) applied to the array x (and y for "pow");
    returns the results as float64 ("sincos": the pair)."""
    import torch
    L = lib or load_library()
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    d_x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(dev)
    d_y = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64)).to(dev) if y is not None else None
    d_o, d_o2 = torch.zeros_like(d_x), torch.zeros_like(d_x)
    torch.cuda.synchronize(dev)
    L.check(L.lib.gpis_libm_batch(LIBM_FNS[fn], d_x.numel(), ctypes.c_void_p(d_x.data_ptr()), ctypes.c_void_p(d_y.data_ptr()) if d_y is not None else None,
                                  ctypes.c_void_p(d_o.data_ptr()), ctypes.c_void_p(d_o2.data_ptr()), None), "gpis_libm_batch")
    torch.cuda.synchronize(dev)
    return (d_o.cpu().numpy(), d_o2.cpu().numpy()) if fn in ("sincos", "sincosf") else d_o.cpu().numpy()


def sort_pairs_u32(keys, vals, device=0, lib=None):
    """Test surface: the library's radix sort on arrays of uint32 keys / values; returns (keys_sorted, vals_sorted)."""
    import torch
    L = lib or load_library()
    dev = torch.device("cuda", device)
    torch.cuda.set_device(dev)
    k = torch.from_numpy(np.ascontiguousarray(keys, dtype=np.uint32).view(np.int32)).to(dev)
    v = torch.from_numpy(np.ascontiguousarray(vals, dtype=np.uint32).view(np.int32)).to(dev)
    ko, vo = torch.zeros_like(k), torch.zeros_like(v)
    torch.cuda.synchronize(dev)
    L.check(L.lib.gpis_sort_pairs_u32(k.numel(), ctypes.c_void_p(k.data_ptr()), ctypes.c_void_p(v.data_ptr()), ctypes.c_void_p(ko.data_ptr()),
                                      ctypes.c_void_p(vo.data_ptr()), None), "gpis_sort_pairs_u32")
    torch.cuda.synchronize(dev)
    return ko.cpu().numpy().view(np.uint32), vo.cpu().numpy().view(np.uint32)


class Medium:
    """gpis_medium handle + the host-pointer convenience calls (numpy in / numpy out)."""

    def __init__(self, params, device=0, lib=None):
        self.L = lib or load_library()
        self.params = as_params(params)
        self.device = int(device)
        h = ctypes.c_void_p()
        st = self.L.lib.gpis_create(_ptr(self.params), int(device), ctypes.byref(h))
        self.L.check(st, "gpis_create")
        self.h = h

    def close(self):
        if self.h:
            self.L.lib.gpis_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def derived(self):
        d = np.zeros((), dtype=DERIVED)
        self.L.check(self.L.lib.gpis_get_derived(self.h, _ptr(d)), "gpis_get_derived")
        return d

    # ---- host-pointer entries --------------------------------------------------------
    def sample_distance(self, rays, want_coeff=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_IN)
        out = np.zeros(rays.shape[0], dtype=SEG_OUT)
        coeff = np.zeros(rays.shape[0], dtype=COND_COEFF) if want_coeff else None
        st = self.L.lib.gpis_sample_distance_host(self.h, rays.shape[0], _ptr(rays), _ptr(out), _ptr(coeff))
        self.L.check(st, "gpis_sample_distance_host")
        return (out, coeff) if want_coeff else out

    def transmittance(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_IN)
        vis = np.zeros(rays.shape[0], dtype=np.uint8)
        self.L.check(self.L.lib.gpis_transmittance_host(self.h, rays.shape[0], _ptr(rays), _ptr(vis)),
                     "gpis_transmittance_host")
        return vis

    # ---- function-space comparison path: device-pointer entries, fed through torch buffers here
    def _fs_call(self, fn_name, rays, states, want_out):
        import torch
        rays = np.ascontiguousarray(rays, dtype=RAY_IN)
        states = np.ascontiguousarray(states, dtype=FS_STATE)
        n = rays.shape[0]
        dev = torch.device("cuda", self.device)
        d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).to(dev)
        d_st = torch.from_numpy(states.view(np.uint8).reshape(-1).copy()).to(dev)
        d_out = torch.zeros(max(n, 1) * (SEG_OUT.itemsize if want_out else 1), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        rc = getattr(self.L.lib, fn_name)(self.h, n, ctypes.c_void_p(d_rays.data_ptr()), ctypes.c_void_p(d_st.data_ptr()),
                                          ctypes.c_void_p(d_out.data_ptr()), None)
        self.L.check(rc, fn_name)
        torch.cuda.synchronize(dev)
        st = d_st.cpu().numpy().view(FS_STATE).copy()
        raw = d_out.cpu().numpy()
        return (raw[:n * SEG_OUT.itemsize].view(SEG_OUT).copy() if want_out else raw[:n].copy()), st

    FS_OPS = {"eigh": 0, "norm_transform": 1, "pinv": 2}

    def fs_linalg(self, op, mats):
        """Test surface: the function-space path's eigen-solver / normTransform / pseudo-inverse on a stack of matrices
        (count, n, n), numpy convention mats[k][i, j]; returns the result stack (and the eigenvalues for "eigh")."""
        import torch
        mats = np.asarray(mats, dtype=np.float64)
        count, n = mats.shape[0], mats.shape[1]
        colmajor = np.ascontiguousarray(np.transpose(mats, (0, 2, 1)))        # column-major storage of every matrix
        dev = torch.device("cuda", self.device)
        d_in = torch.from_numpy(colmajor.reshape(-1)).to(dev)
        d_out = torch.zeros_like(d_in)
        d_w = torch.zeros(count * n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        self.L.check(self.L.lib.gpis_fs_linalg_batch(self.h, self.FS_OPS[op], n, count, ctypes.c_void_p(d_in.data_ptr()), ctypes.c_void_p(d_out.data_ptr()),
                                                     ctypes.c_void_p(d_w.data_ptr()), None), "gpis_fs_linalg_batch")
        torch.cuda.synchronize(dev)
        out = np.transpose(d_out.cpu().numpy().reshape(count, n, n), (0, 2, 1)).copy()
        return (out, d_w.cpu().numpy().reshape(count, n)) if op == "eigh" else out

    def fs_sample_distance(self, rays, states):
        return self._fs_call("gpis_fs_sample_distance_batch", rays, states, True)

    def fs_transmittance(self, rays, states):
        return self._fs_call("gpis_fs_transmittance_batch", rays, states, False)

    def mean_color_emission(self, points):
        """(color, emission) of the mean at double-precision points (n, 3)"""
        p = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        col = np.zeros((p.shape[0], 3), dtype=np.float32)
        emi = np.zeros((p.shape[0], 3), dtype=np.float32)
        self.L.check(self.L.lib.gpis_mean_color_emission_host(self.h, p.shape[0], _ptr(p), _ptr(col), _ptr(emi)), "gpis_mean_color_emission_host")
        return col, emi

    def eval_value(self, q):
        q = np.ascontiguousarray(q, dtype=QUERY)
        val = np.zeros(q.shape[0], dtype=np.float32)
        gid = np.zeros(q.shape[0], dtype=np.int32)
        self.L.check(self.L.lib.gpis_eval_value_host(self.h, q.shape[0], _ptr(q), _ptr(val), _ptr(gid)),
                     "gpis_eval_value_host")
        return val, gid

    def eval_gradient(self, q):
        q = np.ascontiguousarray(q, dtype=QUERY)
        g = np.zeros((q.shape[0], 3), dtype=np.float32)
        self.L.check(self.L.lib.gpis_eval_gradient_host(self.h, q.shape[0], _ptr(q), _ptr(g)),
                     "gpis_eval_gradient_host")
        return g

    def conditioning(self, q, target_val, target_grad):
        q = np.ascontiguousarray(q, dtype=QUERY)
        tv = np.ascontiguousarray(target_val, dtype=np.float32)
        tg = np.ascontiguousarray(target_grad, dtype=np.float32)
        co = np.zeros(q.shape[0], dtype=COND_COEFF)
        self.L.check(self.L.lib.gpis_conditioning_host(self.h, q.shape[0], _ptr(q), _ptr(tv), _ptr(tg), _ptr(co)),
                     "gpis_conditioning_host")
        return co

    def nee_pdf(self, q):
        q = np.ascontiguousarray(q, dtype=NEE_QUERY)
        out = np.zeros(q.shape[0], dtype=np.float32)
        self.L.check(self.L.lib.gpis_nee_pdf_host(self.h, q.shape[0], _ptr(q), _ptr(out)), "gpis_nee_pdf_host")
        return out

    def nee_grad(self, q):
        q = np.ascontiguousarray(q, dtype=NEE_QUERY)
        out = np.zeros((q.shape[0], 3), dtype=np.float32)
        self.L.check(self.L.lib.gpis_nee_grad_host(self.h, q.shape[0], _ptr(q), _ptr(out)), "gpis_nee_grad_host")
        return out

    def counters(self):
        e = ctypes.c_uint64()
        s = ctypes.c_uint64()
        self.L.check(self.L.lib.gpis_get_counters(self.h, ctypes.byref(e), ctypes.byref(s)), "gpis_get_counters")
        return e.value, s.value

    def reset_counters(self):
        self.L.check(self.L.lib.gpis_reset_counters(self.h), "gpis_reset_counters")

    def set_variance_grid(self, voxels, world_to_index, interpolate="linear", origin=(0, 0, 0)):
        """GridNonstationaryCovariance: the voxel grid the variance is read from (voxels[k, j, i], world_to_index = a 4x4 matrix)"""
        d, v = variance_grid_desc(voxels, world_to_index, interpolate, origin)
        self.L.check(self.L.lib.gpis_set_variance_grid(self.h, d.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p)), "gpis_set_variance_grid")

    def build_guide(self, half_extent_cells=16, points_per_cell=32):
        self.L.check(self.L.lib.gpis_build_guide(self.h, int(half_extent_cells), int(points_per_cell)), "gpis_build_guide")

    def guide_info(self):
        """dict of the guide field's brick counts and bytes (zeros without a guide field)"""
        info = np.zeros((), dtype=GUIDE_INFO)
        self.L.check(self.L.lib.gpis_get_guide_info(self.h, info.ctypes.data_as(ctypes.c_void_p)), "gpis_get_guide_info")
        return {k: int(info[k]) for k in GUIDE_INFO.names}

    def drop_guide(self):
        self.L.check(self.L.lib.gpis_drop_guide(self.h), "gpis_drop_guide")

    def guide_steps(self):
        n = ctypes.c_uint64()
        self.L.check(self.L.lib.gpis_get_guide_steps(self.h, ctypes.byref(n)), "gpis_get_guide_steps")
        return n.value

    def guide_selfcheck(self, points3_dev_ptr, n, stream=None):
        c, v = ctypes.c_uint64(), ctypes.c_uint64()
        r, b = ctypes.c_float(), ctypes.c_float()
        self.L.check(self.L.lib.gpis_guide_selfcheck(self.h, int(n), ctypes.c_void_p(int(points3_dev_ptr)), ctypes.byref(c), ctypes.byref(v),
                                                     ctypes.byref(r), ctypes.byref(b), stream), "gpis_guide_selfcheck")
        return c.value, v.value, r.value, b.value

    def guide_raycheck(self, rays_dev_ptr, n, steps, stream=None):
        """(certified steps, violations) over the first `steps` march positions of n device-resident rays."""
        c, v = ctypes.c_uint64(), ctypes.c_uint64()
        self.L.check(self.L.lib.gpis_guide_raycheck(self.h, int(n), ctypes.c_void_p(int(rays_dev_ptr)), int(steps), ctypes.byref(c),
                                                    ctypes.byref(v), stream), "gpis_guide_raycheck")
        return c.value, v.value

    def set_batch_order(self, scattered):
        self.L.check(self.L.lib.gpis_set_batch_order(self.h, 1 if scattered else 0), "gpis_set_batch_order")

    OPTIONS = {"march_form": 0, "wave_tail": 1, "paths_sort": 2, "paths_presort": 3, "chunk_log2": 4, "persistent": 5, "solo_max": 6, "range_len": 7, "defer_grad": 8}
    MARCH_FORMS = {"auto": 0, "resident": 1, "wave": 2}

    def set_option(self, name, value):
        if name == "march_form" and isinstance(value, str):
            value = self.MARCH_FORMS[value]
        self.L.check(self.L.lib.gpis_set_option(self.h, self.OPTIONS[name], int(value)), "gpis_set_option")

    def get_option(self, name):
        v = ctypes.c_longlong()
        self.L.check(self.L.lib.gpis_get_option(self.h, self.OPTIONS[name], ctypes.byref(v)), "gpis_get_option")
        return v.value

    def set_profiling(self, on):
        self.L.check(self.L.lib.gpis_set_profiling(self.h, int(bool(on))), "gpis_set_profiling")

    def kernel_profile(self, which):
        """(total_ms, launches, n_eval, n_seg) of the sampleDistance (0) / transmittance (1) kernel."""
        ms = ctypes.c_double()
        n, e, s = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self.L.check(self.L.lib.gpis_get_kernel_profile(self.h, int(which), ctypes.byref(ms), ctypes.byref(n),
                                                        ctypes.byref(e), ctypes.byref(s)), "gpis_get_kernel_profile")
        return ms.value, n.value, e.value, s.value

    # ---- device-pointer entries (raw addresses, e.g. torch tensors) ------------------
    def call(self, name, *args):
        fn = getattr(self.L.lib, name)
        conv = [(_ptr(a) if (a is None or isinstance(a, np.ndarray)) else a) for a in args]
        self.L.check(fn(self.h, *conv), name)
