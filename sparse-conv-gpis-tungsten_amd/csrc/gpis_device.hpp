// gpis_device.hpp — device-side math of the sparse-convolution GPIS path for gfx950.
//
// Everything here is hand-written for CDNA4 wave64 lanes; one lane carries one ray / query.
// Results must equal the reference CPU path, which is built SSE4.2-only (no FMA), so this
// translation unit is compiled with -ffp-contract=off and every fp32 operation rounds
// separately, in the reference's association order:
//   * Tungsten Vec::dot / lengthSq:           (c0 + c1) + c2          (math/Vec.hpp:134-140, 200-206)
//   * Eigen fixed-size 3-term reductions:     c0 + (c1 + c2)          (redux_novec_unroller)
//   * UniformSampler::next3D():               first draw → z, third → x (g++ argument order)
// expf is re-implemented bit-exactly after glibc 2.35's expf (sysdeps/ieee754/flt-32/e_expf.c:
// N=32 table, degree-3 polynomial in double); v_exp_f32 is 1 ulp-class and would flip signs
// at zero crossings.
//
// Reference citations: SCN = src/core/math/SparseConvolutionNoise.cpp,
// GPF = src/core/math/GPFunctions.{cpp,hpp}, SCNM = src/core/media/SparseConvolutionNoiseMedium.cpp,
// GPM = src/core/media/GaussianProcessMedium.cpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpis.h"
#include "gpis_libm.hpp"

#pragma clang fp contract(off)

namespace gpis {

#define GPIS_DEV __device__ __forceinline__
// a kernel defined in a header shared by several translation units is a template on a dummy parameter, so that only
// the TU that launches it (k<0><<<...>>>) instantiates and compiles it (gpis_launch.hpp)
#define GPIS_TU_KERNEL template <int TU_ = 0>

constexpr int kLevelMin = -24;
constexpr int kLevelMax = 24;
constexpr int kLevels = kLevelMax - kLevelMin + 1;

// Host-precomputed, device-resident constants of one medium (uniform → scalar loads).
// ProceduralNoise / ProceduralNoiseVec of type "noise" with a ramp (GPF.cpp:43-138): logs from the host libm
struct DevRamp {
    int32_t enabled, type;
    double scale, offset, log_min2, log_max2;        // first ramp: clamp(coord * scale + offset), lerp of log(min^2), log(max^2)
    double scale2, offset2, log2_min2, log2_max2;    // second ramp (bottom_top_left_right)
    double vmin, vmax;                               // "min", "max" as given: the scalar sandstone / rust noises lerp between them
};
GPIS_DEV double ramp_unit(double coord, double scale, double offset, double la, double lb)
{
    double u = coord * scale + offset;
    u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
    double l = la * (1.0 - u) + lb * u;
    return sqrt(exp_glibc(l));
}
// GridNonstationaryCovariance's voxel grid (gpis_set_variance_grid); vox == nullptr: getUnscaledVariance = 1
struct DevGrid {
    const float *vox;
    int32_t on;                  // the medium is of the grid flavour
    int32_t interpolate;
    int32_t dims[3], origin[3];
    float lo[3], hi[3];          // bounds_min + 2, bounds_max - 3
    float T[12];                 // rows 0..2 of invNaturalTransform
    float offset, scale, thresh, surf_amp, vol_amp, surf_ls, vol_ls;
    int32_t separate;
};
struct DevModel {
    // flags (SCNM.cpp:57-73, SCN.cpp:21-30)
    int32_t single_realization, iso3d, sampling_1d, correlation_xy, ctx;
    int32_t activate_conditioning, scheme_1d_eff, multi_res, nonstationary, multi_resolution_grid;
    int32_t use_aniso_mtx, surf_vol_phase_separate, has_mean_additional, absorption_only, max_bounces;
    int32_t fbm_noise;            // some procedural field is a sandstone / rust noise, or the wrapper is of the grid flavour: only the
                                  // all-features path instance evaluates those
    DevGrid grid;
    float surf_vol_phase_amp_thresh;
    uint32_t seed, n_impulses, min_step;
    float step_size, impulse_density, sigma;
    // SE constants (GPF.cpp:654-679)
    float w2l[9], w2l_T[9], l2w_T[9];
    float invcov_world[9];        // diag(l_aniso_inv^2) or _cov_mtx_inv (before the per-call scalings)
    double cov_det_sqrt_world;    // the double `covDeterminantSqrt` (GPF.cpp:743-748) before the globalScale/localScale powers
    double pi_pow_1_5;            // pow(M_PI, 1.5) from the host libm
    double sqrt_pi;               // sqrt(M_PI)
    float kernel_scale;           // _kernelScale
    float mtx_factor;             // max l_aniso / end_pt max (GPF.cpp:700-707)
    // stationary fast constants (valid when !nonstationary)
    float radius_world, radius_iso;       // splattingKernelRadius(false/true, 1)
    float norm3d_world, norm3d_iso, norm1d;   // sqrt(variance) (SCN.cpp:254, 279, 308, 331)
    float exp_arg_max;                        // max_a(A_a) * R^2 of the grid space in use: bound on ab^T A ab inside the unit ball
    // non-stationary ramp (GPF.cpp:87-95, GPF.hpp:773-774)
    int32_t ls_ramp_type;
    float ls_maxval;
    double ls_min, ls_max, ls_scale, ls_offset, ls_log_min2, ls_log_max2;
    DevRamp ls;                   // the same "ls" field in the general form (bottom_top_left_right needs the second ramp)
    DevRamp var, color, emission; // "var" (GPF.cpp:1638-1641), mean "color" / "emission" (GPF.hpp:849-857)
    DevRamp aniso;                // "aniso" angle field (GPF.cpp:1678-1689)
    float sigma_raw;              // the stationary kernel's sigma (sparseConvNoiseAmplitude() of GPF.cpp:711)
    // Matérn / Gabor kernels (GPF.cpp:866-1214): world-space 3D sampling only; radius_world / norm3d_world hold their constants
    int32_t kernel_type;          // gpis_kernel_type
    float matern_v, k_l;          // "v", "lengthScale"
    float gabor_a, gabor_f;       // 1 / a_inv, 1 / f_inv
    float gabor_omega[3];         // normalised
    // function-space comparison path (gpis_fs.hpp): sample_points, step_size, the covariance's own aniso (GPF.hpp:1602-1605)
    int32_t fs_n;
    float fs_aniso[3];
    double fs_step;
    // multi-resolution level tables (host libm: powf / logf), index level - kLevelMin
    float level_scale[kLevels];           // powf(2.5, level)
    int32_t level_addseed[kLevels];       // (int)floorf(logf(level_scale)/logf(2.5))
    float log_base;                       // logf(2.5f)
    int32_t world_addseed;                // floor(log(worldSamplingSpatialScale)/log(2.5)), SCN.cpp:252
    // medium coefficients (GPM.cpp:152-158)
    float sigma_s_over_t[3];
    // means
    gpis_mean mean[2];
    double lin_dir[2][3];
};

struct Counters {
    unsigned long long n_eval;
    unsigned long long n_seg;
};

// ---------------------------------------------------------------------------------------
// integer layer (bit-exact): MathUtil.hpp:179-224, UniformSampler.hpp:41-75, BitManip.hpp:47-50
// ---------------------------------------------------------------------------------------
constexpr uint32_t P2 = 2246822519U, P3 = 3266489917U, P4 = 668265263U, P5 = 374761393U;

GPIS_DEV uint32_t rotl17(uint32_t h) { return __builtin_rotateleft32(h, 17); }
GPIS_DEV uint32_t xx_final(uint32_t h)
{
    h = P2 * (h ^ (h >> 15));
    h = P3 * (h ^ (h >> 13));
    return h ^ (h >> 16);
}
GPIS_DEV uint32_t xxhash32_1(uint32_t p) { return xx_final(P4 * rotl17(p + P5)); }
GPIS_DEV uint32_t xxhash32_2(uint32_t x, uint32_t y) { return xx_final(P4 * rotl17(y + P5 + x * P3)); }
GPIS_DEV uint32_t xxhash32_3(uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t h = P4 * rotl17(z + P5 + x * P3);
    h += y * P3;
    return xx_final(P4 * rotl17(h));
}
GPIS_DEV uint32_t xxhash32_4(uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    uint32_t h = P4 * rotl17(w + P5 + x * P3);
    h += y * P3;
    h = P4 * rotl17(h);
    h += z * P3;
    return xx_final(P4 * rotl17(h));
}

constexpr uint64_t kPcgMult = 6364136223846793005ULL;

struct Pcg32 {
    uint64_t state;
    GPIS_DEV uint32_t next_i()
    {
        uint64_t old = state;
        state = old * kPcgMult + 1ULL;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return __builtin_rotateright32(xs, rot);
    }
    GPIS_DEV void set_state(uint64_t s)
    {
        // assign, then discard next2D(): two LCG steps folded into one affine map
        // a^2*s + (a + 1)   (mod 2^64)
        state = s * (kPcgMult * kPcgMult) + (kPcgMult + 1ULL);
    }
};
// the output permutation of next_i() for a given (old) state, and the LCG advanced by 1..4 draws in closed form:
// state_{+j} = A_j * state + C_j (mod 2^64).  The four draws of one impulse then hang off ONE state instead of a chain
// of four dependent 64-bit multiply-adds (same values, 4x shorter dependency chain in the lockstep generator loop).
GPIS_DEV uint32_t pcg_output(uint64_t old)
{
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return __builtin_rotateright32(xs, rot);
}
constexpr uint64_t kPcgA1 = kPcgMult, kPcgC1 = 1ULL;
constexpr uint64_t kPcgA2 = kPcgA1 * kPcgMult, kPcgC2 = kPcgC1 * kPcgMult + 1ULL;
constexpr uint64_t kPcgA3 = kPcgA2 * kPcgMult, kPcgC3 = kPcgC2 * kPcgMult + 1ULL;
constexpr uint64_t kPcgA4 = kPcgA3 * kPcgMult, kPcgC4 = kPcgC3 * kPcgMult + 1ULL;
GPIS_DEV float normalized_uint(uint32_t i) { return __uint_as_float((i >> 9u) | 0x3F800000u) - 1.0f; }

// ---------------------------------------------------------------------------------------
// expf, bit-for-bit glibc 2.35 (e_expf.c / exp2f_data.c): exp(x) = 2^(k/N) * 2^(r/N)
// ---------------------------------------------------------------------------------------
__device__ const uint64_t kExp2fTab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
    0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
    0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
    0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
    0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};

GPIS_DEV float expf_glibc(float x)
{
    // special cases of e_expf.c: below -0x1.9fe368p6f the result underflows to +0;
    // [−103.97, −88] runs the main path and rounds into the subnormals.
    if (x < -0x1.9fe368p6f)
        return 0.0f;
    if (x > 0x1.62e42ep6f)
        return __builtin_huge_valf();
    const double InvLn2N = 0x1.71547652b82fep+0 * 32;
    const double Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32;
    double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + Shift;
    uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= Shift;
    double r = z - kd;
    uint64_t t = kExp2fTab[ki & 31];
    t += ki << (52 - 5);
    double s = __longlong_as_double((long long)t);
    // the x86-64 libm that the reference links is built with FMA contraction in this file; both
    // contracted and uncontracted forms round to the same float (verified on 2e8 inputs)
    double zz = __builtin_fma(C0, r, C1);
    double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(zz, r2, y);
    y = y * s;
    return (float)y;
}

// ---------------------------------------------------------------------------------------
// small vectors in the reference's evaluation order
// ---------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
struct V3d { double x, y, z; };
struct V4 { float v, gx, gy, gz; };
struct Mat9 { float m[9]; };   // a 3x3 kernel matrix passed by value (registers) to an out-of-line function

GPIS_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
// ProceduralNoise::operator() / ProceduralNoiseVec::operator() (one component), GPF.cpp:43-103, _const = 1
GPIS_DEV double ramp_eval(const DevRamp &R, V3d p)
{
    if (R.type == GPIS_RAMP_BOTTOM_TOP_LEFT_RIGHT) {
        float bottomTop = (float)ramp_unit(p.y, R.scale, R.offset, R.log_min2, R.log_max2);
        float leftRight = (float)ramp_unit(p.x, R.scale2, R.offset2, R.log2_min2, R.log2_max2);
        return (double)(bottomTop * leftRight) - 1.0 * 1.0;
    }
    double coord = R.type == GPIS_RAMP_BOTTOM_TOP ? p.y : (R.type == GPIS_RAMP_LEFT_RIGHT ? p.x : p.z);
    return ramp_unit(coord, R.scale, R.offset, R.log_min2, R.log_max2) - 1.0;
}
// ---- "sandstone" / "rust": fbm over 3D simplex noise, math/SdfFunctions.cpp:199-296 (oracle: sdf_random3 / sdf_simplex3d /
// sdf_fbm, pinned bit for bit against the reference's SdfFunctions.cpp).  Here `sin` and `sqrt` are ocml's: the hash
// fract(512 * float(4096 sin(.))) absorbs a last-bit difference of the double sine unless it crosses a float rounding
// boundary, so values agree with the CPU except at isolated points (tolerance stated in the tests).  Not inlined: only the
// all-features instance of the path calls them, from cold code.
static __device__ __attribute__((noinline)) float sdf_simplex3d(float px, float py, float pz)
{
    auto dot3 = [](float ax, float ay, float az, float bx, float by, float bz) { float r = ax * bx; r += ay * by; r += az * bz; return r; };
    auto random3 = [&](float cx, float cy, float cz, float &rx, float &ry, float &rz) {        // SdfFunctions.cpp:199-208
        float j = (float)(4096.0 * sin_glibc((double)dot3(cx, cy, cz, 17.0f, 59.4f, 15.0f)));
        double v;
        v = 512.0 * (double)j; rz = (float)(v - floor(v));
        j = (float)((double)j * .125);
        v = 512.0 * (double)j; rx = (float)(v - floor(v));
        j = (float)((double)j * .125);
        v = 512.0 * (double)j; ry = (float)(v - floor(v));
        rx = rx - 0.5f; ry = ry - 0.5f; rz = rz - 0.5f;
    };
    const float F3 = 0.3333333f, G3 = 0.1666667f;
    const float pf = dot3(px, py, pz, F3, F3, F3);
    const float sx = floorf(px + pf), sy = floorf(py + pf), sz = floorf(pz + pf);
    const float sg = dot3(sx, sy, sz, G3, G3, G3);
    const float x = (px - sx) + sg, y = (py - sy) + sg, z = (pz - sz) + sg;
    const float ex = (x - y) < 0.0f ? 0.f : 1.f, ey = (y - z) < 0.0f ? 0.f : 1.f, ez = (z - x) < 0.0f ? 0.f : 1.f;
    const float i1x = ex * (1.0f - ez), i1y = ey * (1.0f - ex), i1z = ez * (1.0f - ey);
    const float i2x = 1.0f - ez * (1.0f - ex), i2y = 1.0f - ex * (1.0f - ey), i2z = 1.0f - ey * (1.0f - ez);
    const float g2 = 2.0f * G3, g3 = 3.0f * G3;
    const float x1 = (x - i1x) + G3, y1 = (y - i1y) + G3, z1 = (z - i1z) + G3;
    const float x2 = (x - i2x) + g2, y2 = (y - i2y) + g2, z2 = (z - i2z) + g2;
    const float x3 = (x - 1.0f) + g3, y3 = (y - 1.0f) + g3, z3 = (z - 1.0f) + g3;
    float w[4] = {dot3(x, y, z, x, y, z), dot3(x1, y1, z1, x1, y1, z1), dot3(x2, y2, z2, x2, y2, z2), dot3(x3, y3, z3, x3, y3, z3)}, d[4];
    for (int i = 0; i < 4; ++i) { const float t = 0.6f - w[i]; w[i] = t < 0.0f ? 0.0f : t; }
    float rx, ry, rz;
    random3(sx, sy, sz, rx, ry, rz); d[0] = dot3(rx, ry, rz, x, y, z);
    random3(sx + i1x, sy + i1y, sz + i1z, rx, ry, rz); d[1] = dot3(rx, ry, rz, x1, y1, z1);
    random3(sx + i2x, sy + i2y, sz + i2z, rx, ry, rz); d[2] = dot3(rx, ry, rz, x2, y2, z2);
    random3(sx + 1.0f, sy + 1.0f, sz + 1.0f, rx, ry, rz); d[3] = dot3(rx, ry, rz, x3, y3, z3);
    for (int i = 0; i < 4; ++i) { w[i] *= w[i]; w[i] *= w[i]; d[i] *= w[i]; }
    float r = d[0] * 52.0f; r += d[1] * 52.0f; r += d[2] * 52.0f; r += d[3] * 52.0f;
    return r;
}
static __device__ __attribute__((noinline)) double sdf_fbm(double ux, double uy, double uz, int octaves)     // SdfFunctions.cpp:276-296
{
    const float gain = 0.65f, lacunarity = 2.1042f;
    float total, frequency = 0.5f, amplitude = gain;
    const float fx = (float)(ux * 5.0), fy = (float)(uy * 5.0), fz = (float)(uz * 5.0);
    total = sdf_simplex3d(fx, fy, fz);
    for (int i = 0; i < octaves; i++) {
        total += sdf_simplex3d(fx * frequency, fy * frequency, fz * frequency) * amplitude;
        frequency *= lacunarity;
        amplitude *= gain;
    }
    total = (float)(((double)total + 2.0) / 4.0);
    return (double)total;
}
GPIS_DEV double sdf_clamp01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
GPIS_DEV double sdf_sandstone_f(V3d p, int oct)                 // the nested fbm of GPF.cpp:72-73, 106-107
{
    const double x = p.x * 0.3, y = p.y * 0.3, z = p.z * 0.3;
    const double f1 = sdf_fbm(x, y, z, oct);
    const double f2 = sdf_fbm(x + f1, y + f1, z + f1, oct);
    return sdf_fbm(x + f2, y + f2, z + f2, oct);
}
GPIS_DEV double sdf_rust_f(V3d p, int oct, double sign)         // GPF.cpp:80-81 (sign -1), 114-115 (sign +1)
{
    const double x = p.x * 2.0, y = p.y * 2.0, z = p.z * 2.0;
    const double a = sdf_fbm(x * .1, y * .1, z * .1, oct) * 0.4;
    const double b = sdf_fbm(x + a, y + a, z + a, oct);
    const double c = sdf_fbm(x * 25., y * 25., z * 25., oct) * 0.1;
    double v = sign < 0.0 ? b - c : b + c;
    v = sdf_clamp01((v - 0.4) / (0.6 - 0.4));                   // smoothStep(0.4, 0.6, .), MathUtil.hpp:107-112
    return v * v * (3.0 - 2.0 * v);
}
// the sandstone / rust branches as ONE out-of-line call per use (the all-features instance inlines the evaluator at many sites)
static __device__ __attribute__((noinline)) double field_scalar_fbm(int type, double vmin, double vmax, double x, double y, double z)
{
    const V3d p{x, y, z};
    const double t = type == GPIS_NOISE_SANDSTONE ? sdf_clamp01(sqrt((sdf_sandstone_f(p, 2) * 1.9) * 1.2) - 0.35) : sdf_clamp01(sdf_rust_f(p, 2, -1.0));
    return vmin * (1.0 - t) + vmax * t;                            // lerp(_min, _max, .)
}
static __device__ __attribute__((noinline)) V3d field_vec_fbm(int type, double x, double y, double z)
{
    const V3d p{x, y, z};
    if (type == GPIS_NOISE_SANDSTONE) {
        const double f = sdf_sandstone_f(p, 10);
        return V3d{sdf_clamp01((sqrt((f * 1.9) * 1.2) - 0.35) * 0.2), sdf_clamp01((sqrt((f * 0.7) * 1.2) - 0.35) * 0.2),
                   sdf_clamp01((sqrt((f * 0.25) * 1.2) - 0.35) * 0.2)};
    }
    const double f = sdf_rust_f(p, 10, 1.0);
    return V3d{0.278 * (1.0 - f) + 1. * f, 0.212 * (1.0 - f) + 1. * f, 0.141 * (1.0 - f) + 1. * f};
}
// ProceduralNoise::operator() (scalar fields "var", "aniso"), GPF.cpp:53-85
GPIS_DEV double field_scalar(const DevRamp &R, V3d p, bool fbm_possible)
{
    if (fbm_possible && R.type >= GPIS_NOISE_SANDSTONE)
        return field_scalar_fbm(R.type, R.vmin, R.vmax, p.x, p.y, p.z);
    return ramp_eval(R, p);
}
// ProceduralNoiseVec::operator() (vector fields "ls", mean "color" / "emission"), GPF.cpp:87-120
GPIS_DEV void field_vec(const DevRamp &R, V3d p, bool fbm_possible, double out[3])
{
    if (fbm_possible && R.type >= GPIS_NOISE_SANDSTONE) {
        const V3d v = field_vec_fbm(R.type, p.x, p.y, p.z);
        out[0] = v.x; out[1] = v.y; out[2] = v.z;
        return;
    }
    out[0] = out[1] = out[2] = ramp_eval(R, p);
}
// ---- GridNonstationaryCovariance, GPF.cpp:1386-1427 over VdbGrid::density, VdbGrid.cpp:405-431 (OpenVDB's PointSampler / BoxSampler
// restated from tools/Interpolation.h — absent dependency, parity unpinned for the lookup; same code as the oracle's).  Cold code of
// the all-features path instance: not inlined.
static __device__ __attribute__((noinline)) float grid_unscaled_variance(const DevGrid &G, double pxd, double pyd, double pzd)
{
    if (!G.vox) return 1.f;
    const float x = (float)pxd, y = (float)pyd, z = (float)pzd;
    float q[3] = {G.T[0] * x + G.T[1] * y + G.T[2] * z + G.T[3], G.T[4] * x + G.T[5] * y + G.T[6] * z + G.T[7], G.T[8] * x + G.T[9] * y + G.T[10] * z + G.T[11]};
    for (int c = 0; c < 3; ++c) {
        const float v = q[c] < G.lo[c] ? G.lo[c] : q[c];
        q[c] = v < G.hi[c] ? v : G.hi[c];
    }
    auto voxel = [&](long i, long j, long k) -> float {
        i -= G.origin[0]; j -= G.origin[1]; k -= G.origin[2];
        if (i < 0 || j < 0 || k < 0 || i >= G.dims[0] || j >= G.dims[1] || k >= G.dims[2]) return 0.f;
        return G.vox[(size_t)i + (size_t)G.dims[0] * ((size_t)j + (size_t)G.dims[1] * (size_t)k)];
    };
    auto lerp = [](float a, float b, double w) -> float { const double temp = (double)(b - a) * w; return a + (float)temp; };
    const double xd = q[0], yd = q[1], zd = q[2];
    float dens;
    if (G.interpolate == 0) {
        dens = voxel((long)floor(xd + 0.5), (long)floor(yd + 0.5), (long)floor(zd + 0.5));
    } else {
        const double fx = floor(xd), fy = floor(yd), fz = floor(zd);
        const long i = (long)fx, j = (long)fy, k = (long)fz;
        const double u = xd - fx, v = yd - fy, w = zd - fz;
        dens = lerp(lerp(lerp(voxel(i, j, k), voxel(i, j, k + 1), w), lerp(voxel(i, j + 1, k), voxel(i, j + 1, k + 1), w), v),
                    lerp(lerp(voxel(i + 1, j, k), voxel(i + 1, j, k + 1), w), lerp(voxel(i + 1, j + 1, k), voxel(i + 1, j + 1, k + 1), w), v), u);
    }
    return (dens + G.offset) * G.scale;
}
// getVariance, GPF.cpp:1393-1403
static __device__ __attribute__((noinline)) double grid_variance(const DevGrid &G, double px, double py, double pz)
{
    const float amplitude = grid_unscaled_variance(G, px, py, pz);
    if (!G.separate) return (double)amplitude;
    return (double)(amplitude < G.thresh ? amplitude * G.surf_amp : amplitude * G.vol_amp);
}
// getKernelScale, GPF.cpp:1409-1420
static __device__ __attribute__((noinline)) float grid_kernel_scale(const DevGrid &G, float px, float py, float pz)
{
    if (!G.separate) return 1.f;
    const float amplitude = grid_unscaled_variance(G, (double)px, (double)py, (double)pz);
    return amplitude < G.thresh ? G.surf_ls : G.vol_ls;
}
// Modified Bessel functions of the second kind K0(x), K1(x), x > 0, for the Matern v = 3/2 splatting kernel (GPF.cpp:1053-1056,
// 1071-1074 call boost::math::cyl_bessel_k, which is neither vendored nor installed: PARITY UNPINNED VS BOOST; the same code as
// the oracle's bessel_k01, which tests/ check against the Wronskian and scipy to 1e-14).  x <= 2: ascending series; x > 2: Steed's
// algorithm on the continued fraction CF2.  Not inlined: cold code of the all-features path instance.
static __device__ __attribute__((noinline)) void bessel_k01(double x, double *k0, double *k1)
{
    const double EULER = 0.57721566490153286061;
    if (x <= 2.0) {
        const double q = 0.25 * x * x, lg = log_glibc(0.5 * x);
        double term0 = 1.0, i0 = 1.0, s0 = 0.0, hk = 0.0;           /* term0 = q^k / (k!)^2 */
        double term1 = 1.0, i1s = 1.0, s1 = 1.0 - 2.0 * EULER;       /* term1 = q^k / (k! (k+1)!), s1 = sum (psi(k+1) + psi(k+2)) term1 */
        double hk1 = 1.0;                                            /* H_{k+1} */
        for (int k = 1; k < 40; ++k) {
            term0 *= q / ((double)k * (double)k);
            hk += 1.0 / (double)k;
            i0 += term0;
            s0 += term0 * hk;
            term1 *= q / ((double)k * (double)(k + 1));
            hk1 += 1.0 / (double)(k + 1);
            i1s += term1;
            s1 += term1 * ((hk - EULER) + (hk1 - EULER));            /* psi(k+1) = H_k - g */
            if (term0 < 1e-18 * i0) break;
        }
        *k0 = -(lg + EULER) * i0 + s0;
        *k1 = 1.0 / x + lg * (0.5 * x * i1s) - 0.25 * x * s1;
        return;
    }
    {
        double b = 2.0 * (1.0 + x), d = 1.0 / b, h = d, delh = d, q1 = 0.0, q2 = 1.0;
        const double a1 = 0.25;
        double qq = a1, c = a1, a = -a1, s = 1.0 + qq * delh;
        for (int i = 2; i <= 500; ++i) {
            a -= 2.0 * (double)(i - 1);
            c = -a * c / (double)i;
            const double qnew = (q1 - b * q2) / a;
            q1 = q2; q2 = qnew;
            qq += c * qnew;
            b += 2.0;
            d = 1.0 / (b + a * d);
            delh = (b * d - 1.0) * delh;
            h += delh;
            const double dels = qq * delh;
            s += dels;
            if (fabs(dels / s) < 1e-17) break;
        }
        h = a1 * h;
        const double rk0 = sqrt(3.14159265358979323846 / (2.0 * x)) * exp_glibc(-x) / s;
        *k0 = rk0;
        *k1 = rk0 * (x + 0.5 - h) / x;
    }
}
GPIS_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
GPIS_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
GPIS_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
GPIS_DEV V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
GPIS_DEV V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
GPIS_DEV float dot(V3 a, V3 b) { float s = a.x * b.x; s += a.y * b.y; s += a.z * b.z; return s; }
GPIS_DEV float length_sq(V3 a) { float r = 0.f; r += a.x * a.x; r += a.y * a.y; r += a.z * a.z; return r; }
GPIS_DEV float length(V3 a) { return sqrtf(length_sq(a)); }
GPIS_DEV V3 normalized(V3 a) { float inv = 1.0f / length(a); return V3{a.x * inv, a.y * inv, a.z * inv}; }
GPIS_DEV V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
GPIS_DEV V3d to_d(V3 a) { return V3d{(double)a.x, (double)a.y, (double)a.z}; }
GPIS_DEV V3 to_f(V3d a) { return V3{(float)a.x, (float)a.y, (float)a.z}; }
GPIS_DEV double length_d(V3d a) { double r = 0.; r += a.x * a.x; r += a.y * a.y; r += a.z * a.z; return sqrt(r); }
GPIS_DEV V3d ray_at(V3d p, V3d rd, double t) { return V3d{p.x + t * rd.x, p.y + t * rd.y, p.z + t * rd.z}; }

GPIS_DEV V4 v4(float v, float gx, float gy, float gz) { return V4{v, gx, gy, gz}; }
GPIS_DEV V4 operator+(V4 a, V4 b) { return V4{a.v + b.v, a.gx + b.gx, a.gy + b.gy, a.gz + b.gz}; }
GPIS_DEV V4 operator*(V4 a, float s) { return V4{a.v * s, a.gx * s, a.gy * s, a.gz * s}; }
GPIS_DEV V4 operator*(float s, V4 a) { return V4{s * a.v, s * a.gx, s * a.gy, s * a.gz}; }
GPIS_DEV V4 operator/(V4 a, float s) { return V4{a.v / s, a.gx / s, a.gy / s, a.gz / s}; }

// Duff et al. ONB, TangentFrame.hpp:74-86
struct Frame { V3 normal, tangent, bitangent; };
GPIS_DEV Frame frame_from_normal(V3 n)
{
    Frame f;
    f.normal = normalized(n);
    float sign = __builtin_copysignf(1.0f, f.normal.z);
    const float a = -1.0f / (sign + f.normal.z);
    const float b = f.normal.x * f.normal.y * a;
    f.tangent = normalized(v3(1.0f + sign * f.normal.x * f.normal.x * a, sign * b, -sign * f.normal.x));
    f.bitangent = normalized(v3(b, sign + f.normal.y * f.normal.y * a, -f.normal.y));
    return f;
}
GPIS_DEV V3 to_local(const Frame &f, V3 p) { return v3(dot(f.tangent, p), dot(f.bitangent, p), dot(f.normal, p)); }
GPIS_DEV V3 to_global(const Frame &f, V3 p) { return (f.tangent * p.x + f.bitangent * p.y) + f.normal * p.z; }

// Eigen fixed-size forms
GPIS_DEV float sum3e(float c0, float c1, float c2) { return c0 + (c1 + c2); }
#define GM(m, r, c) ((m)[3 * (r) + (c)])
GPIS_DEV V3 eig_matvec(const float *m, V3 v)
{
    return v3(sum3e(GM(m, 0, 0) * v.x, GM(m, 0, 1) * v.y, GM(m, 0, 2) * v.z),
              sum3e(GM(m, 1, 0) * v.x, GM(m, 1, 1) * v.y, GM(m, 1, 2) * v.z),
              sum3e(GM(m, 2, 0) * v.x, GM(m, 2, 1) * v.y, GM(m, 2, 2) * v.z));
}
// (ab^T M) — shared by dist2_ab's row vector and the three column dots (same expressions)
GPIS_DEV V3 eig_row_times(V3 ab, const float *m)
{
    return v3(sum3e(ab.x * GM(m, 0, 0), ab.y * GM(m, 1, 0), ab.z * GM(m, 2, 0)),
              sum3e(ab.x * GM(m, 0, 1), ab.y * GM(m, 1, 1), ab.z * GM(m, 2, 1)),
              sum3e(ab.x * GM(m, 0, 2), ab.y * GM(m, 1, 2), ab.z * GM(m, 2, 2)));
}
GPIS_DEV float cofactor3(const float *m, int i, int j)
{
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return GM(m, i1, j1) * GM(m, i2, j2) - GM(m, i1, j2) * GM(m, i2, j1);
}
// Eigen/src/LU/InverseImpl.h:130-179
GPIS_DEV void eig_inverse3(const float *m, float *r)
{
    float c0 = cofactor3(m, 0, 0), c1 = cofactor3(m, 1, 0), c2 = cofactor3(m, 2, 0);
    float det = sum3e(c0 * GM(m, 0, 0), c1 * GM(m, 1, 0), c2 * GM(m, 2, 0));
    float invdet = 1.0f / det;
    float t10 = cofactor3(m, 0, 1) * invdet, t11 = cofactor3(m, 1, 1) * invdet, t20 = cofactor3(m, 0, 2) * invdet;
    float t12 = cofactor3(m, 2, 1) * invdet, t21 = cofactor3(m, 1, 2) * invdet, t22 = cofactor3(m, 2, 2) * invdet;
    GM(r, 1, 0) = t10; GM(r, 1, 1) = t11; GM(r, 2, 0) = t20;
    GM(r, 1, 2) = t12; GM(r, 2, 1) = t21; GM(r, 2, 2) = t22;
    GM(r, 0, 0) = c0 * invdet; GM(r, 0, 1) = c1 * invdet; GM(r, 0, 2) = c2 * invdet;
}
// mult(Matrix3f, Vec), GPF.hpp:131-139
GPIS_DEV V3 mult3(const float *a, V3 b)
{
    return v3(GM(a, 0, 0) * b.x + GM(a, 0, 1) * b.y + GM(a, 0, 2) * b.z,
              GM(a, 1, 0) * b.x + GM(a, 1, 1) * b.y + GM(a, 1, 2) * b.z,
              GM(a, 2, 0) * b.x + GM(a, 2, 1) * b.y + GM(a, 2, 2) * b.z);
}

// Box–Muller in double, Gaussian.cpp:21-34 (PI is the float constant of Angle.hpp:8)
GPIS_DEV void rand_normal_2(Pcg32 &s, double &z1, double &z2)
{
    double u1 = (double)normalized_uint(s.next_i());
    double u2 = (double)normalized_uint(s.next_i());
    double r = sqrt(-2 * log_glibc(1. - u1));
    const float two_pi = 2 * 3.1415926536f;
    double ang = (double)two_pi * u2;
    double sn, cs;                       // rand_normal_2 takes both, which a gcc build turns into one sincos call (gpis_libm.hpp)
    sincos_glibc(ang, &sn, &cs);
    z1 = r * cs;
    z2 = r * sn;
}

// ---------------------------------------------------------------------------------------
// per-path data
// ---------------------------------------------------------------------------------------
struct RayInfo { uint32_t px, py, spp, seg, scene_seed; float t; };   // MediumSample.hpp:14-18

struct Realization {   // SparseConvolutionNoiseRealization's mutable part (SCN.hpp:39-40)
    gpis_cond_coeff c;     // coeff_3D or coeff_1D, whichever the medium uses
    uint32_t n_eval;
};

// ---- means (GPF.hpp:867-1005, GaussianProcess.cpp:379-393) ------------------------------
GPIS_DEV double mean_eval(const DevModel &M, int w, V3d a)
{
    const gpis_mean &mu = M.mean[w];
    if (mu.type == GPIS_MEAN_HOMOGENEOUS)
        return (double)mu.offset;
    if (mu.type == GPIS_MEAN_SPHERICAL) {
        V3d d{a.x - mu.center[0], a.y - mu.center[1], a.z - mu.center[2]};
        return length_d(d) - (double)mu.radius;
    }
    double dx = a.x - mu.center[0], dy = a.y - mu.center[1], dz = a.z - mu.center[2];
    double dt = dx * M.lin_dir[w][0]; dt += dy * M.lin_dir[w][1]; dt += dz * M.lin_dir[w][2];
    double v = dt * (double)mu.scale, mn = (double)mu.min;
    return v > mn ? v : mn;
}
GPIS_DEV V3d mean_grad(const DevModel &M, int w, V3d a)
{
    const gpis_mean &mu = M.mean[w];
    if (mu.type == GPIS_MEAN_HOMOGENEOUS)
        return V3d{0., 0., 0.};
    if (mu.type == GPIS_MEAN_SPHERICAL) {
        V3d d{a.x - mu.center[0], a.y - mu.center[1], a.z - mu.center[2]};
        double inv = 1.0 / length_d(d);
        return V3d{d.x * inv, d.y * inv, d.z * inv};
    }
    double dx = a.x - mu.center[0], dy = a.y - mu.center[1], dz = a.z - mu.center[2];
    double dt = dx * M.lin_dir[w][0]; dt += dy * M.lin_dir[w][1]; dt += dz * M.lin_dir[w][2];
    if (dt * (double)mu.scale < (double)mu.min)
        return V3d{0., 0., 0.};
    return V3d{M.lin_dir[w][0] * (double)mu.scale, M.lin_dir[w][1] * (double)mu.scale, M.lin_dir[w][2] * (double)mu.scale};
}
GPIS_DEV void mean_weight_space(const DevModel &M, V3d p, double &mean, int &id)
{
    mean = mean_eval(M, 0, p);
    id = 0;
    if (M.has_mean_additional) {
        double add = mean_eval(M, 1, p);
        if (add < mean) { mean = add; id = 1; }
    }
}


// ---------------------------------------------------------------------------------------
// wave-level primitives shared by the cooperative kernels (gpis_fast.hpp, gpis_guide.hpp) and the persistent
// march (gpis_persist.inc)
// ---------------------------------------------------------------------------------------
// ---- LCG jump-ahead tables: state after n further draws = A_n * s + C_n -------------------
struct JumpTable {
    uint64_t A[64];
    uint64_t C[64];
};
constexpr JumpTable make_jump_table()
{
    JumpTable t{};
    uint64_t a = 1, c = 0;
    for (int n = 0; n < 256; ++n) {
        if ((n & 3) == 0) { t.A[n >> 2] = a; t.C[n >> 2] = c; }
        a = a * kPcgMult;
        c = c * kPcgMult + 1ULL;
    }
    return t;
}
__device__ const JumpTable kJump4 = make_jump_table();   // entry k: jump by 4k draws

// one wave per cell: lane k produces impulse k exactly as the sequential generator (SCN.cpp:376-385)
GPIS_DEV void gen_impulse(uint32_t ci, uint32_t cj, uint32_t ck, uint32_t seed, uint64_t jA, uint64_t jC,
                          float &px, float &py, float &pz, float &pw)
{
    const uint32_t h = xxhash32_4(ck, cj, ci, seed) + 1u;
    const uint64_t s0 = (uint64_t)h * (kPcgMult * kPcgMult) + (kPcgMult + 1ULL);   // set_state
    Pcg32 g;
    g.state = jA * s0 + jC;
    pz = normalized_uint(g.next_i());
    py = normalized_uint(g.next_i());
    px = normalized_uint(g.next_i());
    pw = (g.next_i() >> 31) ? 1.f : -1.f;   // Bernoulli(next1D(), -1, 1, 0.5)
}
typedef float float2v __attribute__((ext_vector_type(2)));    // two fp32 lanes of a v_pk_* instruction
#ifndef GPIS_LDS_BCAST
#define GPIS_LDS_BCAST 1                // 1: candidate impulses are broadcast through LDS, 0: with v_readlane
#endif
struct FastLds {
    uint64_t exptab[32];                // glibc's exp2f table, staged once per wave (ds_read_b64 per use)
#if GPIS_LDS_BCAST
    // the current cell's impulses, read back at a wave-uniform address (= broadcast), laid out in the
    // pairs the candidate body consumes with packed fp32 instructions
    float4 xyr[64];                     // (x, y, kernelRadius * x, kernelRadius * y)
    float4 zw[64];                      // (z, kernelRadius * z, w = +-1, unused)
#endif
    // staging area of the sideways evaluators' ordered sums (ChainSum below): groups of four fp32 values, read back as broadcasts
    float4 chain[64];
    // lane tables of the two-way candidate split (coop_noise3d): the r-th lane with a query / the r-th lane without one
    uint32_t owner_tab[64], idle_tab[64];
    // k-way candidate split (coop_noise3d_ksplit): the values of one pass, group-major (group g, helper j at g*k + j)
    float4 exch[16];
};
constexpr int kChainGroups = 64;
template <class LDS>
GPIS_DEV void fast_lds_init(LDS &lds)
{
    const int lane = (int)(threadIdx.x & 63);
    if (lane < 32)
        lds.exptab[lane] = kExp2fTab[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// wave-uniform value → SGPR
GPIS_DEV int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
GPIS_DEV float uni_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// lane k's value, k wave-uniform (v_readlane_b32: no LDS round trip)
GPIS_DEV float lane_f(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }

// ---- ordered sums of the sideways evaluators (lane = impulse) ------------------------------------------------
// The reference adds the passing impulses of a cell in ascending k, starting from 0, and then adds the cell's sum to the
// running sum, cells in dx,dy,dz order (SCN.cpp:368-371, 380-392).  With lane = impulse that is a serial chain over
// values held by different lanes.  Round 1 walked the ballot with one v_readlane + v_add_f32 per value (9 issue
// cycles; 11 % of the guided sampleDistance kernel's issue cycles).  Here the passing lanes' values are compacted
// into LDS (slot = popcount of the ballot below the lane), padded with +0 to a multiple of four, and every lane sums
// them from wave-uniform ds_read_b128 reads (a broadcast on the LDS pipe): 2 issue cycles per value.  The padding is
// exact: x + (+0) = x for every x except -0, and a partial sum is never -0 (it starts from +0, and (+0) + (-0) = +0).
struct ChainSum {
    float sum = 0.f;                  // cells finished so far, in order
    int used = 0;                     // groups staged (wave-uniform)
    unsigned long long ends = 0ULL;   // bit g: group g closes its cell
};
template <class LDS>
GPIS_DEV void chain_flush(LDS &lds, ChainSum &cs)
{
    // one wave per workgroup and in-order LDS: only the compiler has to be told that lanes talk to each other here
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // scalar control flow throughout: `ends` and the group index live in SGPRs
    int g = 0;
    for (unsigned long long e = cs.ends; e; e &= e - 1ULL) {
        const int last = __builtin_ctzll(e);
        float cell = 0.f;
        for (; g <= last; ++g) {
            const float4 v = lds.chain[g];
            cell = (((cell + v.x) + v.y) + v.z) + v.w;
        }
        cs.sum = cs.sum + cell;
    }
    cs.used = 0;
    cs.ends = 0ULL;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// stages one cell: m = ballot of the lanes whose impulse passes (m != 0, wave-uniform), pass = this lane's bit, c = its value
template <class LDS>
GPIS_DEV void chain_push(LDS &lds, ChainSum &cs, unsigned long long m, bool pass, float c)
{
    const int cnt = __popcll(m), groups = (cnt + 3) >> 2;
    if (__builtin_expect(cs.used + groups > kChainGroups, 0))
        chain_flush(lds, cs);
    const int lane = (int)(threadIdx.x & 63);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    // passing lanes take slots 0 .. cnt-1 in lane order, the others follow (a permutation of 0 .. 63): they supply the +0 padding
    const int slot = pass ? rank : cnt + (lane - rank);
    if (slot < 4 * groups)
        reinterpret_cast<float *>(lds.chain)[4 * cs.used + slot] = pass ? c : 0.f;
    cs.used += groups;
    cs.ends |= 1ULL << (cs.used - 1);
}

// expf_glibc with the table in LDS and the range checks folded into selects (same results).
// NONPOS: the caller guarantees x <= 0 (or NaN), so the overflow check cannot fire and is left out.
// NOUNDER: the caller guarantees that every result it USES has x >= -100 (results of smaller x are
// discarded), so the underflow select is left out as well.
// (LDS = any struct with the staged table as `uint64_t exptab[32]`)
template <bool NONPOS = false, bool NOUNDER = false, class LDS = FastLds>
GPIS_DEV float expf_glibc_lds(const LDS &lds, float x)
{
    const double InvLn2N = 0x1.71547652b82fep+0 * 32;
    const double Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32;
    double z = InvLn2N * (double)x;
    double kd = z + Shift;
    uint64_t ki = (uint64_t)__double_as_longlong(kd);
    uint64_t t = lds.exptab[ki & 31];
    kd -= Shift;
    double r = z - kd;
    t += ki << (52 - 5);
    double zz = __builtin_fma(C0, r, C1);
    double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(zz, r2, y);
    y = y * __longlong_as_double((long long)t);
    float res = (float)y;
    if (!NOUNDER)
        res = x < -0x1.9fe368p6f ? 0.0f : res;
    if (!NONPOS)
        res = x > 0x1.62e42ep6f ? __builtin_huge_valf() : res;
    return res;
}


// ---- persistent march (gpis_persist.inc) ------------------------------------------------------------------
template <int Q>
struct PersistQueue {
    uint64_t exptab[32];                           // glibc's exp2f table (expf_glibc_lds)
    uint32_t e0[Q][64], e1[Q][64], e2[Q][64];      // (x | w sign, y, z) of the impulses that passed the unit-ball test, slot-major
    // ring of unmasked ray indices waiting for a free lane (refill, gpis_persist.inc).  56 entries, not 128: LDS is handed out in
    // 1280-byte granules on this chip, 12 one-wave workgroups per CU leave 12 800 B each, and the queues above take 12 544 — with a
    // 128-entry ring (13 072 B) only 11 workgroups fit although the occupancy query still says 12, and every persistent launch
    // lost 5-8 % (C3 480x270x8 sampleDistance 631 -> 670 ms) whichever refill scheme it ran.
    static constexpr int kReady = 56;
    uint32_t ready[kReady];
    uint32_t rq[4];                                // the ring's head, count, "batch exhausted" flag: kept here, not in (S)GPRs across the march
};
enum PersistPhase : int {
    PP_IDLE = 0,      // no ray (batch drained)
    PP_COND = 1,      // unconditioned evaluation at the segment start (conditioning, SCN.cpp:431-595)
    PP_F0 = 2,        // f(nearT)                                   SCNM.cpp:125-128
    PP_MARCH = 3,     // f(t), fixed steps                           SCNM.cpp:132-141, 172-173
    PP_REFINE = 4,    // secant-then-shrink refinement               SCNM.cpp:142-162
    PP_FINAL = 5,     // lastVal = f(farT) on exit                   SCNM.cpp:176-181
    PP_GRAD = 6,      // sampleGradient                              SCNM.cpp:93-100, GPM.cpp:283 / 319
    PP_DONE = 7       // result complete: write it, take the next ray
};
struct PersistArgs {
    size_t n;
    const gpis_ray_in *rays;
    gpis_seg_out *out;            // sampleDistance
    gpis_cond_coeff *coeff;       // sampleDistance, optional
    uint8_t *visible;             // transmittance
    const uint8_t *mask;          // optional
    Counters *cnt;
    unsigned int *next;           // the global ray counter of this launch (zeroed by the host)
    int solo_max;                 // sideways evaluation when at most this many lanes have a job
};

// ---- plain data shared by every specialisation of the path section ------------------------
struct LevelInfo { float lo, hi, ratio_lo, ratio_hi; int add_lo, add_hi; };
struct NeeShared {
    V3 grad_constraint_3d, grad_3d_iso_ray, plane_normal_world;
    float plane_stretch_jacobian, isect_dist;
};

// ---- medium level state (GPM.cpp:221-393) ------------------------------------------------
struct MediumState {
    bool first_scatter;
    int bounce, last_gp_id, scheme;
    V3d last_aniso;
    float last_val;
    RayInfo info;
};
GPIS_DEV void state_from_ray(const gpis_ray_in &r, MediumState &s)
{
    s.first_scatter = r.first_scatter != 0;
    s.bounce = r.bounce;
    s.last_gp_id = r.last_gp_id;
    s.scheme = GPIS_UNI;
    s.last_aniso = V3d{r.last_aniso[0], r.last_aniso[1], r.last_aniso[2]};
    s.last_val = r.last_val;
    s.info = RayInfo{r.pixel[0], r.pixel[1], r.spp, r.segment, r.scene_seed, r.info_t};
}
GPIS_DEV double lerp_d(double a, double b, double ratio) { return a * (1.0 - ratio) + b * ratio; }

}   // namespace gpis

// ---------------------------------------------------------------------------------------------
// The path section (covariance layer … sampleDistance) is compiled once per SPECIALISATION: the
// all-features instance needs 256 VGPRs and spills 2 000+ SGPRs (1 wave/SIMD), so the march kernels
// are instantiated with the configuration's structural flags as compile-time constants and the
// launcher picks the instance (gpis_hip.hip: sample_distance_impl).  GPIS_FLAG_x(M) is either the
// runtime field or a literal.
// ---------------------------------------------------------------------------------------------
// generic: every flag read at run time.  Its names are visible in gpis through a using-directive
// (not an inline namespace: ADL must not see them from inside the specialised namespaces).
#define GPIS_PATH_NS generic
#define GPIS_PATH_OUTLINE_NOISE3D 1
#define GPIS_FLAG_fbm_noise(M) ((M).fbm_noise != 0)
#define GPIS_FLAG_other_kernels(M) ((M).kernel_type != GPIS_KERNEL_SQUARED_EXPONENTIAL)
#define GPIS_FLAG_aniso_field(M) ((M).aniso.enabled)
#define GPIS_FLAG_sampling_1d(M) ((M).sampling_1d)
#define GPIS_FLAG_multi_res(M) ((M).multi_res)
#define GPIS_FLAG_nonstationary(M) ((M).nonstationary)
#define GPIS_FLAG_multi_resolution_grid(M) ((M).multi_resolution_grid)
#define GPIS_FLAG_activate_conditioning(M) ((M).activate_conditioning)
#include "gpis_path.inc"
// (no using-directive: every user names its instance — fast / guided / wavefront kernels spec_3d, the single-query entries generic)

// 1D sampling along the ray (config C2)
#define GPIS_PATH_NS spec_1d
#define GPIS_FLAG_fbm_noise(M) 0
#define GPIS_FLAG_other_kernels(M) 0
#define GPIS_FLAG_aniso_field(M) 0
#define GPIS_FLAG_sampling_1d(M) 1
#define GPIS_FLAG_multi_res(M) ((M).multi_res)
#define GPIS_FLAG_nonstationary(M) ((M).nonstationary)
#define GPIS_FLAG_multi_resolution_grid(M) ((M).multi_resolution_grid)
#define GPIS_FLAG_activate_conditioning(M) ((M).activate_conditioning)
#include "gpis_path.inc"

// 3D sampling, stationary kernel (config C0 and its per-path variants, C1 without the fast path)
#define GPIS_PATH_NS spec_3d
#define GPIS_FLAG_fbm_noise(M) 0
#define GPIS_FLAG_other_kernels(M) 0
#define GPIS_FLAG_aniso_field(M) 0
#define GPIS_FLAG_sampling_1d(M) 0
#define GPIS_FLAG_multi_res(M) 0
#define GPIS_FLAG_nonstationary(M) 0
#define GPIS_FLAG_multi_resolution_grid(M) 0
#define GPIS_FLAG_activate_conditioning(M) ((M).activate_conditioning)
#include "gpis_path.inc"

// 3D sampling, non-stationary length scale on the multi-resolution grid (config C3)
#define GPIS_PATH_NS spec_3d_multires
#define GPIS_FLAG_fbm_noise(M) 0
#define GPIS_FLAG_other_kernels(M) 0
#define GPIS_FLAG_aniso_field(M) 0
#define GPIS_FLAG_sampling_1d(M) 0
#define GPIS_FLAG_multi_res(M) 1
#define GPIS_FLAG_nonstationary(M) 1
#define GPIS_FLAG_multi_resolution_grid(M) 1
#define GPIS_FLAG_activate_conditioning(M) ((M).activate_conditioning)
#include "gpis_path.inc"
