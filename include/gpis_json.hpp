// gpis_json.hpp — the ONE mapping from the reference's JSON keys to gpis_params.
//
// Included by both adapters: integration/HipSparseConvNoiseMedium.cpp (Tungsten's JsonPtr, compiled against the reference's
// headers) and host/HipSparseConvNoiseMedium.cpp (the stand-alone adapter's own parser).  Until round 3 each kept its own copy of
// this table and they had drifted (neither knew the sandstone / rust noises, Matern's "v" = 1.5 or the grid flavour of the
// non-stationary wrapper once the library did).  What differs between the two — how a JSON node is asked for a child, a number, a
// string, a vector, and how an error is reported — is the accessor type A:
//
//   struct A {
//       using Node = ...;                                             // cheap to copy
//       static bool child(const Node &o, const char *key, Node &out); // object member, if present
//       template <class T> static void num(const Node &o, const char *key, T &dst);   // number (or bool -> 0 / 1); dst untouched when absent
//       static void flag(const Node &o, const char *key, int32_t &dst);               // bool (or number) -> 0 / 1
//       static void str(const Node &o, const char *key, std::string &dst);
//       static void vec3f(const Node &o, const char *key, float *dst);                // scalar or 3-array, as JsonPtr reads a Vec3f
//       static void vec3d(const Node &o, const char *key, double *dst);
//       static void mat3f(const Node &o, const char *key, float *dst9);               // row-major 3x3
//       [[noreturn]] static void fail(const std::string &what);
//   };
//
// Every function cites the reference's fromJson it follows.
#pragma once
#include <string>

#include "gpis.h"

namespace gpis_json {

// GaussianProcessMedium.cpp:30-41
template <class A> int correlationContext(const std::string &name)
{
    if (name == "global") return GPIS_CTX_GLOBAL;
    if (name == "renewal+") return GPIS_CTX_RENEWAL_PLUS;
    if (name == "renewal") return GPIS_CTX_RENEWAL;
    if (name == "none") return GPIS_CTX_NONE;
    A::fail("Invalid correlation context: '" + name + "'");
}
// SparseConvolutionNoiseMedium.cpp:36-45
template <class A> int samplingScheme1D(const std::string &name)
{
    if (name == "uni" || name == "UNI") return GPIS_UNI;
    if (name == "nee" || name == "NEE") return GPIS_NEE;
    if (name == "mis" || name == "MIS") return GPIS_MIS;
    A::fail("Invalid sparse conv sampling scheme: '" + name + "'");
}
// ProceduralNoise(Vec)::stringToNoiseType, GPFunctions.hpp:644-661
template <class A> int noiseType(const std::string &noise)
{
    if (noise == "bottom_top") return GPIS_RAMP_BOTTOM_TOP;
    if (noise == "left_right") return GPIS_RAMP_LEFT_RIGHT;
    if (noise == "front_back") return GPIS_RAMP_FRONT_BACK;
    if (noise == "bottom_top_left_right") return GPIS_RAMP_BOTTOM_TOP_LEFT_RIGHT;
    if (noise == "sandstone") return GPIS_NOISE_SANDSTONE;
    if (noise == "rust") return GPIS_NOISE_RUST;
    A::fail("Invalid noise typ function: '" + noise + "'");
}
// ProceduralNoise::fromJson / ProceduralNoiseVec::fromJson, GPFunctions.hpp:671-688, 752-769
template <class A> void readRamp(const typename A::Node &v, gpis_ramp &r)
{
    r.enabled = 1;
    std::string noise = "bottom_top";
    A::str(v, "noise", noise);
    r.type = noiseType<A>(noise);
    A::num(v, "min", r.min); A::num(v, "max", r.max); A::num(v, "start", r.start); A::num(v, "end", r.end);
    A::num(v, "min2", r.min2); A::num(v, "max2", r.max2); A::num(v, "start2", r.start2); A::num(v, "end2", r.end2);
}
// MeanFunction subclasses, GPFunctions.hpp:867-1005
template <class A> void readMean(const typename A::Node &m, gpis_mean &dst)
{
    std::string type = "spherical";
    A::str(m, "type", type);
    if (type == "homogeneous") {                     // GPFunctions.hpp:871-874
        dst.type = GPIS_MEAN_HOMOGENEOUS;
        A::num(m, "offset", dst.offset);
    } else if (type == "spherical") {                // GPFunctions.hpp:908-912
        dst.type = GPIS_MEAN_SPHERICAL;
        A::vec3d(m, "center", dst.center);
        A::num(m, "radius", dst.radius);
    } else if (type == "linear") {                   // GPFunctions.hpp:953-962
        dst.type = GPIS_MEAN_LINEAR;
        A::vec3d(m, "reference_point", dst.center);
        A::vec3d(m, "direction", dst.dir);
        A::num(m, "scale", dst.scale);
        A::num(m, "min", dst.min);
    } else {
        A::fail("Unsupported mean function type: '" + type + "'");
    }
}
// The stationary kernels: SquaredExponentialCovariance::fromJson (GPFunctions.cpp:654-679, "localScale" GPFunctions.hpp:1481-1484),
// MaternCovariance (:866-876), GaborAnisotropic / GaborIsotropic (:1086-1096, 1155-1162)
template <class A> void readStationary(const typename A::Node &c, const std::string &type, gpis_params &p)
{
    if (type == "squared_exponential") {
        p.kernel_type = GPIS_KERNEL_SQUARED_EXPONENTIAL;
        A::num(c, "sigma", p.sigma);
        A::num(c, "lengthScale", p.length_scale);
        A::vec3f(c, "aniso", p.aniso);
        A::flag(c, "useAnisoMtx", p.use_aniso_mtx);
        A::mat3f(c, "anisoMtx", p.aniso_mtx);
        A::num(c, "localScale", p.local_scale);
    } else if (type == "matern") {
        p.kernel_type = GPIS_KERNEL_MATERN;
        A::num(c, "sigma", p.sigma);
        A::num(c, "v", p.matern_v);
        A::num(c, "lengthScale", p.length_scale);
        A::vec3f(c, "aniso", p.aniso);
        A::num(c, "localScale", p.local_scale);
    } else if (type == "gabor_aniso" || type == "gabor_iso") {
        p.kernel_type = type == "gabor_aniso" ? GPIS_KERNEL_GABOR_ANISO : GPIS_KERNEL_GABOR_ISO;
        A::num(c, "sigma", p.sigma);
        A::num(c, "a_inv", p.gabor_a_inv);
        A::num(c, "f_inv", p.gabor_f_inv);
        A::vec3f(c, "omega", p.gabor_omega);
        A::num(c, "localScale", p.local_scale);
    } else {
        A::fail("Unsupported covariance type: '" + type + "'");
    }
}
// CovarianceFunction objects of a GaussianProcess, incl. the two non-stationary wrappers
template <class A> void readCovariance(const typename A::Node &c, gpis_params &p)
{
    std::string type = "squared_exponential";
    A::str(c, "type", type);
    typename A::Node inner = c;
    if (type == "proc_nonstationary") {              // ProceduralNonstationaryCovariance::fromJson, GPFunctions.cpp:1590-1606, GPFunctions.hpp:2211-2217
        p.nonstationary = 1;
        A::flag(c, "multiResolutionGrid", p.multi_resolution_grid);
        if (A::child(c, "cov", inner)) {
            std::string it = "squared_exponential";
            A::str(inner, "type", it);
            readStationary<A>(inner, it, p);
        }
        typename A::Node ls = c;
        if (A::child(c, "ls", ls)) {                 // ProceduralNoiseVec, GPFunctions.hpp:759-776
            std::string noise = "bottom_top";
            A::str(ls, "noise", noise);
            p.ls_ramp_type = noiseType<A>(noise);
            A::num(ls, "min", p.ls_min); A::num(ls, "max", p.ls_max); A::num(ls, "start", p.ls_start); A::num(ls, "end", p.ls_end);
            A::num(ls, "min2", p.ls_min2); A::num(ls, "max2", p.ls_max2); A::num(ls, "start2", p.ls_start2); A::num(ls, "end2", p.ls_end2);
        }
        typename A::Node f = c;
        if (A::child(c, "var", f)) readRamp<A>(f, p.var);            // GPFunctions.cpp:1593-1595
        if (A::child(c, "aniso", f)) readRamp<A>(f, p.aniso_field);  // GPFunctions.cpp:1600-1602
    } else if (type == "grid_nonstationary") {       // GridNonstationaryCovariance::fromJson, GPFunctions.cpp:1326-1358 (factory name: GaussianProcessFactory.cpp:34)
        // the "grid" / "variance" VDB itself is handed over by the loader with gpis_set_variance_grid
        p.nonstationary = 1;
        p.grid_nonstationary = 1;
        A::flag(c, "multiResolutionGrid", p.multi_resolution_grid);
        if (A::child(c, "cov", inner)) {
            std::string it = "squared_exponential";
            A::str(inner, "type", it);
            readStationary<A>(inner, it, p);
        }
        A::num(c, "offset", p.grid_offset);
        A::num(c, "scale", p.grid_scale);
        A::flag(c, "surf_vol_amp_separate", p.grid_surf_vol_amp_separate);
        A::num(c, "surf_vol_amp_thresh", p.grid_surf_vol_amp_thresh);
        A::num(c, "surf_amp_scale", p.grid_surf_amp_scale);
        A::num(c, "vol_amp_scale", p.grid_vol_amp_scale);
        A::num(c, "surf_ls_scale", p.grid_surf_ls_scale);
        A::num(c, "vol_ls_scale", p.grid_vol_ls_scale);
    } else {
        readStationary<A>(c, type, p);
    }
}
// GaussianProcess::fromJson (GaussianProcess.cpp:172-190) for an inline object
template <class A> void readGaussianProcess(const typename A::Node &gp, gpis_params &p)
{
    typename A::Node m = gp, f = gp;
    if (A::child(gp, "mean", m)) {
        readMean<A>(m, p.mean);
        if (A::child(m, "color", f)) readRamp<A>(f, p.mean_color);          // MeanFunction::fromJson, GPFunctions.hpp:808-818
        if (A::child(m, "emission", f)) readRamp<A>(f, p.mean_emission);
    }
    if (A::child(gp, "mean_additional", m)) {        // GPSampleNodeCSG's second mean (GaussianProcess.cpp:25-39)
        p.has_mean_additional = 1;
        readMean<A>(m, p.mean_additional);
    }
    if (A::child(gp, "covariance", m))
        readCovariance<A>(m, p);
}
// The medium object itself: GaussianProcessMedium::fromJson (GaussianProcessMedium.cpp:97-126) and
// SparseConvolutionNoiseMedium::fromJson (SparseConvolutionNoiseMedium.cpp:57-73).  "max_bounces" (Medium.cpp:29-38) is read by
// the caller: Tungsten's base class owns it on the integration side.
template <class A> void readMedium(const typename A::Node &v, gpis_params &p)
{
    A::vec3f(v, "sigma_a", p.sigma_a);
    A::vec3f(v, "sigma_s", p.sigma_s);
    A::num(v, "density", p.density);
    std::string ctxt = "goldfish";                   // the reference's (invalid) default: the key is effectively required
    A::str(v, "correlation_context", ctxt);
    p.correlation_context = correlationContext<A>(ctxt);
    typename A::Node gp = v;
    if (A::child(v, "gaussian_process", gp))
        readGaussianProcess<A>(gp, p);
    A::num(v, "step_size", p.step_size);
    A::num(v, "min_step", p.min_step);
    A::num(v, "seed", p.seed);
    A::num(v, "impulse_density", p.impulse_density);
    A::flag(v, "single_realization", p.single_realization);
    A::flag(v, "isotropic_3D_sampling", p.isotropic_3d_sampling);
    A::flag(v, "1D_sampling", p.sampling_1d);
    std::string scheme = "uni";
    A::str(v, "1D_sampling_scheme", scheme);
    p.scheme_1d = samplingScheme1D<A>(scheme);
    A::flag(v, "1D_gradient_correlationXY", p.correlation_xy);
    A::flag(v, "surf_vol_phase_separate", p.surf_vol_phase_separate);
    A::num(v, "surf_vol_phase_amp_thresh", p.surf_vol_phase_amp_thresh);
}

}   // namespace gpis_json
