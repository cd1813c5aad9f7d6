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

#pragma clang fp contract(off)

namespace gpis {

#define GPIS_DEV __device__ __forceinline__

constexpr int kLevelMin = -24;
constexpr int kLevelMax = 24;
constexpr int kLevels = kLevelMax - kLevelMin + 1;

// Host-precomputed, device-resident constants of one medium (uniform → scalar loads).
struct DevModel {
    // flags (SCNM.cpp:57-73, SCN.cpp:21-30)
    int32_t single_realization, iso3d, sampling_1d, correlation_xy, ctx;
    int32_t activate_conditioning, scheme_1d_eff, multi_res, nonstationary, multi_resolution_grid;
    int32_t use_aniso_mtx, surf_vol_phase_separate, has_mean_additional, absorption_only, max_bounces;
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
    // non-stationary ramp (GPF.cpp:87-95, GPF.hpp:773-774)
    int32_t ls_ramp_type;
    float ls_maxval;
    double ls_min, ls_max, ls_scale, ls_offset, ls_log_min2, ls_log_max2;
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

GPIS_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
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
    double r = sqrt(-2 * log(1. - u1));
    const float two_pi = 2 * 3.1415926536f;
    double ang = (double)two_pi * u2;
    z1 = r * cos(ang);
    z2 = r * sin(ang);
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

// ---- covariance layer --------------------------------------------------------------------
GPIS_DEV double ls_ramp(const DevModel &M, V3d p)   // ProceduralNoiseVec, GPF.cpp:87-95
{
    double coord = M.ls_ramp_type == GPIS_RAMP_BOTTOM_TOP ? p.y : (M.ls_ramp_type == GPIS_RAMP_LEFT_RIGHT ? p.x : p.z);
    double u = coord * M.ls_scale + M.ls_offset;
    u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
    double l = M.ls_log_min2 * (1.0 - u) + M.ls_log_max2 * u;
    return sqrt(exp(l)) - 1.0;
}
GPIS_DEV float cov_lateral_scale(const DevModel &M, V3 p) { return M.nonstationary ? (float)ls_ramp(M, to_d(p)) : 1.0f; }
GPIS_DEV float cov_ns_scale(const DevModel &M, V3 p)
{
    if (!M.nonstationary || M.multi_resolution_grid)
        return 1.f;
    return cov_lateral_scale(M, p) / M.ls_maxval;
}
GPIS_DEV float cov_world_sampling_scale(const DevModel &M) { return M.nonstationary ? M.ls_maxval : 1.f; }
GPIS_DEV float cov_kernel_radius(const DevModel &M, bool isIdentity, float localScale)
{
    if (!M.nonstationary)
        return isIdentity ? M.radius_iso : (localScale == 1.0f ? M.radius_world : M.kernel_scale * localScale * M.mtx_factor);
    localScale = (float)((double)localScale * (M.multi_resolution_grid ? 1.0 : (double)M.ls_maxval));
    localScale *= 1.f;
    if (isIdentity)
        return M.kernel_scale;
    return M.kernel_scale * localScale * M.mtx_factor;
}
GPIS_DEV float se_variance3d(const DevModel &M, float dens, float R, bool isIdentity, float globalScale, float localScale)
{
    double idua = (double)(dens / (R * R * R));
    double cds = 1.0;
    if (!isIdentity) {
        cds = M.cov_det_sqrt_world;
        cds *= pow((double)globalScale, 3.0);
    }
    cds *= pow((double)localScale, 3.0);
    double iks = M.pi_pow_1_5 * cds;
    return (float)(idua * iks);
}
GPIS_DEV float se_variance1d(const DevModel &M, float dens, float R, float localScale)
{
    double idua = (double)(dens / R);
    double iks = M.sqrt_pi * (double)localScale;
    return (float)(idua * iks);
}
GPIS_DEV float cov_norm3d(const DevModel &M, V3 p, float dens, float R, bool isIdentity, float globalScale)
{
    if (!M.nonstationary)
        return isIdentity ? M.norm3d_iso : M.norm3d_world;
    float localScale = M.multi_resolution_grid ? 1.0f : cov_ns_scale(M, p);
    return sqrtf(se_variance3d(M, dens, R, isIdentity, globalScale, localScale));
}
GPIS_DEV float cov_norm1d(const DevModel &M, V3 p, float dens, float R)
{
    if (!M.nonstationary)
        return M.norm1d;
    return sqrtf(se_variance1d(M, dens, R, cov_ns_scale(M, p)));
}
GPIS_DEV float cov_xf_scale(const DevModel &M, float globalScale)
{
    if (!M.nonstationary)
        return globalScale;
    return M.multi_resolution_grid ? globalScale : M.ls_maxval;
}
// x / 1.0f and x * 1.0f are exact, so the (wave-uniform) unit scale of stationary kernels skips the IEEE divide
GPIS_DEV V3 cov_pos_w2l(const DevModel &M, V3 v, float s) { float c = cov_xf_scale(M, s); V3 r = eig_matvec(M.w2l, v); return c == 1.0f ? r : r / c; }
GPIS_DEV V3 cov_grad_w2l(const DevModel &M, V3 v, float s) { float c = cov_xf_scale(M, s); V3 r = eig_matvec(M.l2w_T, v); return c == 1.0f ? r : r * c; }
GPIS_DEV V3 cov_grad_l2w(const DevModel &M, V3 v, float s) { float c = cov_xf_scale(M, s); V3 r = eig_matvec(M.w2l_T, v); return c == 1.0f ? r : r / c; }
GPIS_DEV float cov_splat_cov_1d(const DevModel &M, V3 pq, V3 pc)
{
    if (!M.nonstationary)
        return 1.0f;
    float a = cov_ns_scale(M, pq), b = cov_ns_scale(M, pc);
    float scale = (float)(0.5 * (double)(a * a + b * b));
    return sqrtf(scale);
}

// SE::getInvCovMtx, GPF.cpp:774-802
GPIS_DEV void se_inv_cov_mtx(const DevModel &M, bool isCov, bool isIso, float globalScale, float localScale, float *A)
{
    if (isIso) {
        for (int i = 0; i < 9; ++i) A[i] = 0.f;
        A[0] = A[4] = A[8] = 1.f;
    } else {
        float g2 = globalScale * globalScale;
        for (int i = 0; i < 9; ++i) A[i] = M.invcov_world[i] / g2;
    }
    if (isCov)
        for (int i = 0; i < 9; ++i) A[i] *= 0.5f;
    float l2 = localScale * localScale;
    for (int i = 0; i < 9; ++i) A[i] /= l2;
    for (int i = 0; i < 9; ++i) A[i] *= 0.5f;
}
// splattingKernel3D (value, gradient) for a given matrix, GPF.cpp:552-562, 804-817
GPIS_DEV V4 se_splat3d(V3 ab, const float *A)
{
    V3 t = eig_row_times(ab, A);
    float absq = sum3e(t.x * ab.x, t.y * ab.y, t.z * ab.z);
    float f = expf_glibc(-absq);
    return v4(f, -2.f * t.x * f, -2.f * t.y * f, -2.f * t.z * f);
}
// splattingKernel3DGrad, GPF.cpp:564-577, 819-833
GPIS_DEV V4 se_splat3d_grad(V3 ab, const float *A, V3 coeff)
{
    V3 t = eig_row_times(ab, A);
    float absq = sum3e(t.x * ab.x, t.y * ab.y, t.z * ab.z);
    float f = expf_glibc(-absq);
    float dx = -2.f * t.x, dy = -2.f * t.y, dz = -2.f * t.z;
    float gx = dx * f, gy = dy * f, gz = dz * f;
    float h00 = (dx * dx - 2.f * GM(A, 0, 0)) * f, h11 = (dy * dy - 2.f * GM(A, 1, 1)) * f, h22 = (dz * dz - 2.f * GM(A, 2, 2)) * f;
    float h01 = (dx * dy - 2.f * GM(A, 0, 1)) * f, h02 = (dx * dz - 2.f * GM(A, 0, 2)) * f, h12 = (dy * dz - 2.f * GM(A, 1, 2)) * f;
    V4 vx = v4(gx, h00, h01, h02), vy = v4(gy, h01, h11, h12), vz = v4(gz, h02, h12, h22);
    return (vx * coeff.x + vy * coeff.y) + vz * coeff.z;
}
// SE 1D family, GPF.cpp:835-865
GPIS_DEV float se_splat1d_val(float ab, float ls)
{
    float den = (float)(2.0 * (double)(ls * ls));
    return expf_glibc(-(ab * ab / den));
}
GPIS_DEV float sqrt2_scaled(float ls) { return (float)((double)ls * 1.4142135623730951); }   // ls * sqrt(2) in double
GPIS_DEV float se_splat1d_grad(float ab, float ls)
{
    float f = se_splat1d_val(ab, ls);
    float den = (float)(2.0 * (double)(ls * ls));
    return (float)(-2.0 * (double)ab / (double)den * (double)f);
}
GPIS_DEV float se_cov1d_val(float ab, float ls) { return se_splat1d_val(ab, sqrt2_scaled(ls)); }
GPIS_DEV float se_cov1d_grad(float ab, float ls) { return se_splat1d_grad(ab, sqrt2_scaled(ls)); }
GPIS_DEV float se_cov2d_2nd(float ab, float ls)
{
    float f = se_cov1d_val(ab, ls);
    float den = (float)(4.0 * (double)(ls * ls));
    float g1 = (float)(-2.0 * (double)ab / (double)den);
    return (float)(((double)(g1 * g1) - 2.0 / (double)den) * (double)f);
}
GPIS_DEV float se_cov2d_2nd_normal(float ab, float ls)
{
    float f = se_cov1d_val(ab, ls);
    float den = (float)(4.0 * (double)(ls * ls));
    return (float)(-2.0 / (double)den * (double)f);
}
// sparseConvNoiseOneOverSecondDerivative, GPF.cpp:713-739
GPIS_DEV void cov_one_over_second_derivative(const DevModel &M, bool isIso, float *out)
{
    float S[9];
    for (int i = 0; i < 9; ++i) {
        float a = isIso ? ((i == 0 || i == 4 || i == 8) ? 1.f : 0.f) : M.invcov_world[i];
        a *= 0.25f;
        S[i] = -2.f * a;
    }
    eig_inverse3(S, out);
}

// ---- noise evaluator (SCN.cpp) -----------------------------------------------------------
GPIS_DEV uint32_t compute_seed(const DevModel &M, const RayInfo &info)   // SCN.cpp:40-49
{
    uint32_t seed = M.seed;
    if (!M.single_realization) {
        uint32_t w = M.ctx == GPIS_CTX_GLOBAL ? 0u : info.seg;
        seed += xxhash32_4(info.px, info.py, info.spp, w) + xxhash32_1(info.scene_seed);
    }
    return seed;
}

struct LevelInfo { float lo, hi, ratio_lo, ratio_hi; int add_lo, add_hi; };
// kernelScaleLevelRatio, SCN.cpp:60-70 (+ the additional_seed of SCN.cpp:252 from the host tables)
GPIS_DEV LevelInfo kernel_scale_level_ratio(const DevModel &M, V3 p)
{
    LevelInfo L;
    float ls = cov_lateral_scale(M, p);
    float level_low = floorf(logf(ls) / M.log_base);
    int li = (int)level_low;
    li = li < kLevelMin ? kLevelMin : (li > kLevelMax - 1 ? kLevelMax - 1 : li);
    L.lo = M.level_scale[li - kLevelMin];
    L.hi = M.level_scale[li + 1 - kLevelMin];
    L.add_lo = M.level_addseed[li - kLevelMin];
    L.add_hi = M.level_addseed[li + 1 - kLevelMin];
    float frac = (ls - L.lo) / (L.hi - L.lo);
    double fd = (double)frac;
    double den = sqrt(1.0 - 2.0 * fd + 2.0 * fd * fd);
    L.ratio_lo = (float)((1.0 - fd) / den);
    L.ratio_hi = (float)(fd / den);
    return L;
}

// cell3D, SCN.cpp:375-395.  A = getInvCovMtx for this evaluation (constant over the cell).
GPIS_DEV V4 cell3d(const DevModel &M, uint32_t ci, uint32_t cj, uint32_t ck, V3 p, uint32_t seed, float R, const float *A, bool diag)
{
    Pcg32 s;
    s.set_state((uint64_t)(uint32_t)(xxhash32_4(ck, cj, ci, seed) + 1u));
    V4 sum = v4(0.f, 0.f, 0.f, 0.f);
    const V3 Rp = R * p;
    const uint32_t n = M.n_impulses;
    for (uint32_t k = 0u; k < n; ++k) {
        float pz = normalized_uint(s.next_i());
        float py = normalized_uint(s.next_i());
        float px = normalized_uint(s.next_i());
        uint32_t wi = s.next_i();
        V3 to_point = p - v3(px, py, pz);
        if (length_sq(to_point) < 1.0f) {
            V3 ab = Rp - R * v3(px, py, pz);
            V3 t = diag ? v3(ab.x * A[0], ab.y * A[4], ab.z * A[8]) : eig_row_times(ab, A);
            float absq = sum3e(t.x * ab.x, t.y * ab.y, t.z * ab.z);
            float f = expf_glibc(-absq);
            V4 k4 = v4(f, -2.f * t.x * f, -2.f * t.y * f, -2.f * t.z * f);
            // Bernoulli(next1D(), -1, 1, 0.5): rv < 0.5  ⇔  top mantissa bit clear
            float w_i = (wi >> 31) ? 1.f : -1.f;
            sum = sum + w_i * k4;
        }
    }
    return sum;
}
// noise3D, SCN.cpp:362-373
GPIS_DEV V4 noise3d(const DevModel &M, V3 p, uint32_t seed, float R, const float *A, bool diag)
{
    V3 pg = p / R;
    V3 fl = v3(floorf(pg.x), floorf(pg.y), floorf(pg.z));
    V3 frac = pg - fl;
    int i = (int)fl.x, j = (int)fl.y, k = (int)fl.z;
    V4 sum = v4(0.f, 0.f, 0.f, 0.f);
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz)
                sum = sum + cell3d(M, (uint32_t)(i + dx), (uint32_t)(j + dy), (uint32_t)(k + dz),
                                   frac - v3((float)dx, (float)dy, (float)dz), seed, R, A, diag);
    return sum;
}
// cell1D / noise1D, SCN.cpp:397-428
GPIS_DEV void noise1d(const DevModel &M, V3 p_world, float t, uint32_t seed, float R, float &ov, float &og)
{
    float tg = t / R;
    float fl = floorf(tg);
    float frac = tg - fl;
    int i = (int)fl;
    float ls = 1.0f;
    if (M.nonstationary) {
        ls = cov_ns_scale(M, p_world);
        ls *= 1.0f;
    }
    float sv = 0.f, sg = 0.f;
    for (int dx = -1; dx <= 1; ++dx) {
        Pcg32 s;
        s.set_state((uint64_t)(uint32_t)(xxhash32_2((uint32_t)(i + dx), seed) + 1u));
        float tc = frac - (float)dx;
        float cv = 0.f, cg = 0.f;
        for (uint32_t k = 0u; k < M.n_impulses; ++k) {
            float t_i = normalized_uint(s.next_i());
            uint32_t wi = s.next_i();
            float to_point = tc - t_i;
            if (to_point * to_point < 1.0f) {
                float ab = R * tc - R * t_i;
                float w_i = (wi >> 31) ? 1.f : -1.f;
                cv += w_i * se_splat1d_val(ab, ls);
                cg += w_i * se_splat1d_grad(ab, ls);
            }
        }
        sv += cv; sg += cg;
    }
    ov = sv; og = sg;
}

GPIS_DEV V3 coeff_origin(const gpis_cond_coeff &c) { return v3(c.ray_origin[0], c.ray_origin[1], c.ray_origin[2]); }
GPIS_DEV V3 coeff_gs(const gpis_cond_coeff &c) { return v3(c.gradient_scale[0], c.gradient_scale[1], c.gradient_scale[2]); }

// evaluateNoise3DNormalized (world space), SCN.cpp:251-260
GPIS_DEV V4 eval_noise3d_world(const DevModel &M, const Realization &r, V3 p, uint32_t seed, int addseed, float R, float kss, bool conditioning)
{
    float A[9];
    se_inv_cov_mtx(M, false, false, kss, cov_ns_scale(M, p), A);
    V4 noise = noise3d(M, p, seed + (uint32_t)addseed, R, A, !M.use_aniso_mtx);
    noise = noise / cov_norm3d(M, p, M.impulse_density, R, false, kss);
    if (M.activate_conditioning && conditioning) {
        float C[9];
        se_inv_cov_mtx(M, true, false, kss, cov_ns_scale(M, p), C);
        V3 ab = p - coeff_origin(r.c);
        noise = noise + (se_splat3d(ab, C) * r.c.value_scale + se_splat3d_grad(ab, C, coeff_gs(r.c)));
    }
    return noise;
}
// evaluateNoise3DIsotropicRayNormalized, SCN.cpp:291-320
GPIS_DEV V4 eval_noise3d_iso_ray(const DevModel &M, const Realization &r, V3 p, V3 rayDir, uint32_t seed, int addseed, float R, float kss, bool conditioning)
{
    V3 ray_dir_iso = normalized(cov_pos_w2l(M, rayDir, 1.0f));
    Frame coord = frame_from_normal(ray_dir_iso);
    V3 p_iso_ray = to_local(coord, cov_pos_w2l(M, p, kss));
    float A[9];
    se_inv_cov_mtx(M, false, true, 1.0f, cov_ns_scale(M, p), A);
    V4 n = noise3d(M, p_iso_ray, seed + (uint32_t)addseed, R, A, true);
    V3 gw = cov_grad_l2w(M, to_global(coord, v3(n.gx, n.gy, n.gz)), kss);
    V4 nw = v4(n.v, gw.x, gw.y, gw.z) / cov_norm3d(M, p, M.impulse_density, R, true, 1.0f);
    if (M.activate_conditioning && conditioning) {
        V3 origin_iso_ray = to_local(coord, cov_pos_w2l(M, coeff_origin(r.c), kss));
        float C[9];
        se_inv_cov_mtx(M, true, true, 1.0f, cov_ns_scale(M, p), C);
        V3 ab = p_iso_ray - origin_iso_ray;
        V4 d = se_splat3d(ab, C) * r.c.value_scale + se_splat3d_grad(ab, C, coeff_gs(r.c));
        V3 gdw = cov_grad_l2w(M, to_global(coord, v3(d.gx, d.gy, d.gz)), kss);
        nw = nw + v4(d.v, gdw.x, gdw.y, gdw.z);
    }
    return nw;
}
// evaluateNoise3D, SCN.cpp:101-124
GPIS_DEV V4 eval_noise3d(const DevModel &M, Realization &r, V3 p, V3 rayDir, uint32_t seed, bool conditioning)
{
    r.n_eval++;
    if (!M.iso3d) {
        if (!M.multi_res) {
            float kss = cov_world_sampling_scale(M);
            return eval_noise3d_world(M, r, p, seed, M.world_addseed, cov_kernel_radius(M, false, 1.0f), kss, conditioning);
        }
        LevelInfo L = kernel_scale_level_ratio(M, p);
        V4 lo = eval_noise3d_world(M, r, p, seed, L.add_lo, cov_kernel_radius(M, false, L.lo), L.lo, conditioning);
        V4 hi = eval_noise3d_world(M, r, p, seed, L.add_hi, cov_kernel_radius(M, false, L.hi), L.hi, conditioning);
        return L.ratio_lo * lo + L.ratio_hi * hi;
    }
    if (!M.multi_res)
        return eval_noise3d_iso_ray(M, r, p, rayDir, seed, 0, cov_kernel_radius(M, true, 1.0f), 1.0f, conditioning);
    LevelInfo L = kernel_scale_level_ratio(M, p);
    V4 lo = eval_noise3d_iso_ray(M, r, p, rayDir, seed, L.add_lo, cov_kernel_radius(M, true, 1.0f), L.lo, conditioning);
    V4 hi = eval_noise3d_iso_ray(M, r, p, rayDir, seed, L.add_hi, cov_kernel_radius(M, true, 1.0f), L.hi, conditioning);
    return L.ratio_lo * lo + L.ratio_hi * hi;
}

// evaluateNoise1DNormalized, SCN.cpp:322-360 → (value, gx_cond, gy_cond, gz)
GPIS_DEV V4 eval_noise1d_normalized(const DevModel &M, const Realization &r, V3 p, V3 rayDir, uint32_t seed, int addseed, float R, float kss,
                                    bool conditioning, float origin_ratio)
{
    V3 ray_dir_iso = normalized(cov_pos_w2l(M, rayDir, 1.0f));
    Frame coord = frame_from_normal(ray_dir_iso);
    V3 p_iso_ray = to_local(coord, cov_pos_w2l(M, p, kss));
    float nv, ng;
    noise1d(M, p, p_iso_ray.z, seed + (uint32_t)addseed, R, nv, ng);
    float nf = cov_norm1d(M, p, M.impulse_density, R);
    nv /= nf; ng /= nf;
    float gsx = 0.f, gsy = 0.f;
    if (M.activate_conditioning && conditioning) {
        V3 o = coeff_origin(r.c);
        float osf = M.multi_res ? origin_ratio : 1.0f;
        V3 origin_iso_ray = to_local(coord, cov_pos_w2l(M, o, kss));
        float d = p_iso_ray.z - origin_iso_ray.z;
        float cs = M.nonstationary ? cov_splat_cov_1d(M, p, o) : 1.0f;
        float kv = se_cov1d_val(d, cs), kg = se_cov1d_grad(d, cs), k2 = se_cov2d_2nd(d, cs);
        float vs_v = r.c.value_scale * kv, vs_g = r.c.value_scale * kg;
        float kz = kss * r.c.gradient_scale[2];
        float gz_v = kz * kg, gz_g = kz * k2;
        nv += osf * (vs_v + gz_v);
        ng += osf * (vs_g + gz_g);
        if (M.correlation_xy) {
            float kn = se_cov2d_2nd_normal(d, cs);
            gsx = osf * kss * r.c.gradient_scale[0] * kn;
            gsy = osf * kss * r.c.gradient_scale[1] * kn;
        }
    }
    return v4(nv, gsx, gsy, ng);
}
// origin_scale_factor of SCN.cpp:338-342 for both levels
GPIS_DEV void origin_ratios(const DevModel &M, const Realization &r, bool conditioning, float &lo, float &hi)
{
    lo = hi = 1.0f;
    if (M.multi_res && M.activate_conditioning && conditioning) {
        LevelInfo O = kernel_scale_level_ratio(M, coeff_origin(r.c));
        lo = O.ratio_lo; hi = O.ratio_hi;
    }
}
// evaluateValueNoise1D, SCN.cpp:126-135
GPIS_DEV float eval_value_noise1d(const DevModel &M, Realization &r, V3 p, V3 rayDir, uint32_t seed, bool conditioning)
{
    r.n_eval++;
    float R = cov_kernel_radius(M, true, 1.0f);
    if (!M.multi_res)
        return eval_noise1d_normalized(M, r, p, rayDir, seed, 0, R, 1.0f, conditioning, 1.0f).v;
    LevelInfo L = kernel_scale_level_ratio(M, p);
    float olo, ohi;
    origin_ratios(M, r, conditioning, olo, ohi);
    float lo = eval_noise1d_normalized(M, r, p, rayDir, seed, L.add_lo, R, L.lo, conditioning, olo).v;
    float hi = eval_noise1d_normalized(M, r, p, rayDir, seed, L.add_hi, R, L.hi, conditioning, ohi).v;
    return L.ratio_lo * lo + L.ratio_hi * hi;
}
GPIS_DEV uint64_t xy_reseed(const RayInfo &info)
{
    return (uint64_t)(uint32_t)(xxhash32_4(info.px, info.py, info.spp, info.seg) + xxhash32_1(info.scene_seed) + 1u);
}
// evaluateGradientNoise1D, SCN.cpp:137-239
GPIS_DEV V3 eval_gradient_noise1d(const DevModel &M, Realization &r, V3 p, float tSegment, V3 rayDir, const RayInfo &info, bool conditioning)
{
    uint32_t seed = compute_seed(M, info);
    r.n_eval++;
    float R = cov_kernel_radius(M, true, 1.0f);
    float scaleX = cov_splat_cov_1d(M, p, p), scaleY = cov_splat_cov_1d(M, p, p);
    double xs0 = 1.0 / (double)scaleX, xs1 = 1.0 / (double)scaleY;
    const double div = (double)sqrtf(2.f);
    bool plus_xy = (M.ctx == GPIS_CTX_RENEWAL_PLUS && M.correlation_xy);
    Pcg32 s;
    if (!M.multi_res) {
        V4 base = eval_noise1d_normalized(M, r, p, rayDir, seed, 0, R, 1.0f, conditioning, 1.0f);
        V3 gbase = v3(base.gx, base.gy, base.gz);
        s.set_state(xy_reseed(info));
        double a, b;
        rand_normal_2(s, a, b);
        double x1 = a / div, y1 = b / div;
        V3 g;
        if (tSegment == 0) {
            g = v3((float)(x1 * xs0), (float)(y1 * xs1), 0.f) + gbase;
        } else if (!plus_xy) {
            double c, d;
            rand_normal_2(s, c, d);
            g = v3((float)(c / div * xs0), (float)(d / div * xs1), 0.f) + gbase;
        } else {
            float factor = (float)(exp((double)(-tSegment * tSegment) / 4.0) * (0.5 - (double)(tSegment * tSegment) * 0.25));
            double mux = (double)factor * 2.0 * x1, muy = (double)factor * 2.0 * y1;
            float cov = (float)(0.5 - (double)(factor * factor * 2));
            double c, d;
            rand_normal_2(s, c, d);
            double sc = (double)sqrtf(cov);
            g = v3((float)(c * sc + mux), (float)(d * sc + muy), 0.f) + gbase;
        }
        V3 ray_dir_iso = normalized(cov_pos_w2l(M, rayDir, 1.0f));
        Frame coord = frame_from_normal(ray_dir_iso);
        return cov_grad_l2w(M, to_global(coord, g), 1.0f);
    }
    LevelInfo L = kernel_scale_level_ratio(M, p);
    float olo, ohi;
    origin_ratios(M, r, conditioning, olo, ohi);
    V4 bl = eval_noise1d_normalized(M, r, p, rayDir, seed, L.add_lo, R, L.lo, conditioning, olo);
    V4 bh = eval_noise1d_normalized(M, r, p, rayDir, seed, L.add_hi, R, L.hi, conditioning, ohi);
    V3 glb = v3(bl.gx, bl.gy, bl.gz), ghb = v3(bh.gx, bh.gy, bh.gz);
    s.set_state(xy_reseed(info));
    double al, bl2, ah, bh2;
    rand_normal_2(s, al, bl2);
    rand_normal_2(s, ah, bh2);
    double x1l = al / div, y1l = bl2 / div, x1h = ah / div, y1h = bh2 / div;
    V3 gl, gh;
    if (tSegment == 0) {
        gl = v3((float)(x1l * xs0), (float)(y1l * xs1), 0.f) + glb;
        gh = v3((float)(x1h * xs0), (float)(y1h * xs1), 0.f) + ghb;
    } else if (!plus_xy) {
        double cl, dl, ch, dh;
        rand_normal_2(s, cl, dl);
        rand_normal_2(s, ch, dh);
        gl = v3((float)(cl / div * xs0), (float)(dl / div * xs1), 0.f) + glb;
        gh = v3((float)(ch / div * xs0), (float)(dh / div * xs1), 0.f) + ghb;
    } else {
        float factor = (float)(exp((double)(-tSegment * tSegment) / 4.0) * (0.5 - (double)(tSegment * tSegment) * 0.25));
        float cov = (float)(0.5 - (double)(factor * factor * 2));
        double sc = (double)sqrtf(cov);
        double cl, dl, ch, dh;
        rand_normal_2(s, cl, dl);
        rand_normal_2(s, ch, dh);
        gl = v3((float)(cl * sc + (double)factor * 2.0 * x1l), (float)(dl * sc + (double)factor * 2.0 * y1l), 0.f) + glb;
        gh = v3((float)(ch * sc + (double)factor * 2.0 * x1h), (float)(dh * sc + (double)factor * 2.0 * y1h), 0.f) + ghb;
    }
    Frame cl_ = frame_from_normal(normalized(cov_pos_w2l(M, rayDir, L.lo)));
    Frame ch_ = frame_from_normal(normalized(cov_pos_w2l(M, rayDir, L.hi)));
    V3 gwl = cov_grad_l2w(M, to_global(cl_, gl), L.lo);
    V3 gwh = cov_grad_l2w(M, to_global(ch_, gh), L.hi);
    return gwl * L.ratio_lo + gwh * L.ratio_hi;
}

GPIS_DEV float cov_amplitude(const DevModel &M) { return M.sigma; }   // GPF.cpp:611-613, 711, 1235-1237

// evaluateValue, SCN.cpp:73-89
GPIS_DEV float evaluate_value(const DevModel &M, Realization &r, V3 p, V3 rayDir, const RayInfo &info, int &gp_id)
{
    uint32_t seed = compute_seed(M, info);
    float noise_val = M.sampling_1d ? eval_value_noise1d(M, r, p, rayDir, seed, true) : eval_noise3d(M, r, p, rayDir, seed, true).v;
    double mean;
    int id;
    mean_weight_space(M, to_d(p), mean, id);
    gp_id = id;
    if (M.surf_vol_phase_separate)
        gp_id = (1.f < M.surf_vol_phase_amp_thresh) ? 0 : 1;
    return (float)((double)(cov_amplitude(M) * noise_val) + mean);
}
// evaluateGradient, SCN.cpp:92-99
GPIS_DEV V3 evaluate_gradient(const DevModel &M, Realization &r, V3 p, float t, V3 rayDir, const RayInfo &info)
{
    uint32_t seed = compute_seed(M, info);
    V3 ng;
    if (M.sampling_1d) {
        ng = eval_gradient_noise1d(M, r, p, t, rayDir, info, true);
    } else {
        V4 n = eval_noise3d(M, r, p, rayDir, seed, true);
        ng = v3(n.gx, n.gy, n.gz);
    }
    double mean;
    int id;
    mean_weight_space(M, to_d(p), mean, id);
    V3 mg = to_f(mean_grad(M, id, to_d(p)));
    return cov_amplitude(M) * ng + mg;
}

// conditioning, SCN.cpp:431-595 (the sanity re-evaluations are executed, as in the reference)
GPIS_DEV void conditioning(const DevModel &M, Realization &r, V3 p, V3 rayDir, float targetVal, V3 targetGrad, const RayInfo &info)
{
    if (!M.activate_conditioning)
        return;
    r.c.value_scale = 0.f;
    r.c.gradient_scale[0] = r.c.gradient_scale[1] = r.c.gradient_scale[2] = 0.f;
    r.c.ray_origin[0] = p.x; r.c.ray_origin[1] = p.y; r.c.ray_origin[2] = p.z;
    LevelInfo K{};
    if (M.multi_res)
        K = kernel_scale_level_ratio(M, p);
    uint32_t seed = compute_seed(M, info);
    float amplitude = cov_amplitude(M);
    if (amplitude == 0)
        return;
    double mean_d;
    int id;
    mean_weight_space(M, to_d(p), mean_d, id);
    float mean = (float)mean_d;
    const bool plus = M.ctx == GPIS_CTX_RENEWAL_PLUS;
    if (M.sampling_1d) {
        float cur = eval_value_noise1d(M, r, p, rayDir, seed, false);
        r.c.value_scale = (targetVal - mean) / amplitude - cur;
        if (M.multi_res)
            r.c.value_scale /= K.ratio_lo * K.ratio_lo + K.ratio_hi * K.ratio_hi;
        (void)eval_value_noise1d(M, r, p, rayDir, seed, true);
        if (plus) {
            V3 mg = to_f(mean_grad(M, id, to_d(p)));
            V3 cgw = eval_gradient_noise1d(M, r, p, 0.f, rayDir, info, false);
            V3 delta = (targetGrad - mg) / amplitude - cgw;
            Frame coord = frame_from_normal(normalized(cov_pos_w2l(M, rayDir, 1.0f)));
            V3 gi = to_local(coord, cov_grad_w2l(M, delta, 1.0f));
            float sX = cov_splat_cov_1d(M, p, p), sY = sX, sZ = sX;
            V3 g = v3((-2.f * gi.x) * (sX * sX), (-2.f * gi.y) * (sY * sY), (-2.f * gi.z) * (sZ * sZ));
            if (M.multi_res)
                g = g / (K.ratio_lo * K.ratio_lo + K.ratio_hi * K.ratio_hi);
            if (!M.correlation_xy) { g.x = 0; g.y = 0; }
            r.c.gradient_scale[0] = g.x; r.c.gradient_scale[1] = g.y; r.c.gradient_scale[2] = g.z;
            (void)eval_gradient_noise1d(M, r, p, 0.f, rayDir, info, true);
        }
        return;
    }
    float cur = eval_noise3d(M, r, p, rayDir, seed, false).v;
    r.c.value_scale = (targetVal - mean) / amplitude - cur;
    if (M.multi_res)
        r.c.value_scale /= K.ratio_lo + K.ratio_hi;
    (void)eval_noise3d(M, r, p, rayDir, seed, true);
    if (plus) {
        V3 mg = to_f(mean_grad(M, id, to_d(p)));
        V4 c4 = eval_noise3d(M, r, p, rayDir, seed, false);
        V3 delta = (targetGrad - mg) / amplitude - v3(c4.gx, c4.gy, c4.gz);
        float S[9];
        V3 g;
        if (M.iso3d) {
            Frame coord = frame_from_normal(normalized(cov_pos_w2l(M, rayDir, 1.0f)));
            V3 gi = to_local(coord, cov_grad_w2l(M, delta, 1.0f));
            cov_one_over_second_derivative(M, true, S);
            g = eig_matvec(S, gi);
            if (M.multi_res) {
                g = g / (K.ratio_lo / K.lo + K.ratio_hi / K.hi);
            } else {
                float q = cov_ns_scale(M, p);
                g = g * (q * q);
            }
        } else {
            cov_one_over_second_derivative(M, false, S);
            g = eig_matvec(S, delta);
            if (M.multi_res) {
                g = g / (K.ratio_lo / (K.lo * K.lo) + K.ratio_hi / (K.hi * K.hi));
            } else {
                float q = cov_lateral_scale(M, p);
                g = g * (q * q);
            }
        }
        r.c.gradient_scale[0] = g.x; r.c.gradient_scale[1] = g.y; r.c.gradient_scale[2] = g.z;
        (void)eval_noise3d(M, r, p, rayDir, seed, true);
    }
}

// neeShared / neeGrad / neePDF, SCN.cpp:601-743
struct NeeShared {
    V3 grad_constraint_3d, grad_3d_iso_ray, plane_normal_world;
    float plane_stretch_jacobian, isect_dist;
};
GPIS_DEV void nee_shared(const DevModel &M, Realization &r, V3 rayDir, V3 normal, V3 p, const RayInfo &info, NeeShared &o)
{
    uint32_t seed = compute_seed(M, info);
    float amplitude = cov_amplitude(M);
    double mean_d;
    int id;
    mean_weight_space(M, to_d(p), mean_d, id);
    V3 mg = to_f(mean_grad(M, id, to_d(p)));
    Frame coord = frame_from_normal(normalized(cov_pos_w2l(M, rayDir, 1.0f)));
    float rc[9] = {coord.tangent.x, coord.bitangent.x, coord.normal.x,
                   coord.tangent.y, coord.bitangent.y, coord.normal.y,
                   coord.tangent.z, coord.bitangent.z, coord.normal.z};
    float scale = 1.f;
    if (M.nonstationary)
        scale = (float)(M.multi_resolution_grid ? 1.0 : (double)M.ls_maxval);
    float sa[9], mtx[9], inv[9];
    for (int i = 0; i < 9; ++i) {
        float w = M.nonstationary ? M.w2l_T[i] / scale : M.w2l_T[i];
        sa[i] = amplitude * w;
    }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            GM(mtx, a, b) = sum3e(GM(sa, a, 0) * GM(rc, 0, b), GM(sa, a, 1) * GM(rc, 1, b), GM(sa, a, 2) * GM(rc, 2, b));
    eig_inverse3(mtx, inv);
    r.n_eval++;
    float R = cov_kernel_radius(M, true, 1.0f);
    if (!M.multi_res) {
        V4 n = eval_noise1d_normalized(M, r, p, rayDir, seed, 0, R, 1.0f, true, 1.0f);
        o.grad_constraint_3d = v3(n.gx, n.gy, n.gz);
    } else {
        LevelInfo L = kernel_scale_level_ratio(M, p);
        float olo, ohi;
        origin_ratios(M, r, true, olo, ohi);
        V4 lo = eval_noise1d_normalized(M, r, p, rayDir, seed, L.add_lo, R, L.lo, true, olo);
        V4 hi = eval_noise1d_normalized(M, r, p, rayDir, seed, L.add_hi, R, L.hi, true, ohi);
        o.grad_constraint_3d = (v3(lo.gx, lo.gy, lo.gz) * L.ratio_lo) / L.lo + (v3(hi.gx, hi.gy, hi.gz) * L.ratio_hi) / L.hi;
    }
    V3 plane_pt = mult3(mtx, v3(0.f, 0.f, o.grad_constraint_3d.z)) + mg;
    V3 cp = cross(mult3(mtx, v3(1.f, 0.f, 0.f)), mult3(mtx, v3(0.f, 1.f, 0.f)));
    o.plane_normal_world = normalized(cp);
    o.plane_stretch_jacobian = length(cp);
    o.isect_dist = dot(plane_pt, o.plane_normal_world) / dot(normal, o.plane_normal_world);
    o.grad_3d_iso_ray = mult3(inv, o.isect_dist * normal - mg);
}
GPIS_DEV V3 nee_grad(const DevModel &M, Realization &r, V3 rayDir, V3 normal, V3 p, const RayInfo &info)
{
    NeeShared s;
    nee_shared(M, r, rayDir, normal, p, info, s);
    return s.isect_dist * normal;
}
GPIS_DEV float nee_pdf(const DevModel &M, Realization &r, V3 rayDir, V3 normal, V3 p, float tSegment, const RayInfo &info)
{
    NeeShared s;
    nee_shared(M, r, rayDir, normal, p, info, s);
    if ((double)s.isect_dist < 0.0)
        return 0.0f;
    float scaleX = cov_splat_cov_1d(M, p, p), scaleY = scaleX;
    float sample_x, sample_y, pdf_area;
    bool plus_xy = (M.ctx == GPIS_CTX_RENEWAL_PLUS && M.correlation_xy);
    const double div = (double)sqrtf(2.f);
    const double kSqrt2 = 1.4142135623730951, kPi = 3.14159265358979323846;
    float dxg = s.grad_3d_iso_ray.x - s.grad_constraint_3d.x, dyg = s.grad_3d_iso_ray.y - s.grad_constraint_3d.y;
    Pcg32 ps;
    if (!M.multi_res) {
        if (!plus_xy) {
            sample_x = (float)((double)dxg * kSqrt2 * (double)scaleX);
            sample_y = (float)((double)dyg * kSqrt2 * (double)scaleY);
            pdf_area = (float)(exp((double)(-sample_x * sample_x) / 2.0) * exp((double)(-sample_y * sample_y) / 2.0) / (2.0 * kPi) * 2.0 * (double)scaleX * (double)scaleY);
        } else {
            ps.set_state(xy_reseed(info));
            double a, b;
            rand_normal_2(ps, a, b);
            double x1 = a / div, y1 = b / div;
            float factor = (float)(exp((double)(-tSegment * tSegment) / 4.0) * (0.5 - (double)(tSegment * tSegment) * 0.25));
            double mux = (double)factor * 2.0 * x1, muy = (double)factor * 2.0 * y1;
            float cov = (float)(0.5 - (double)(factor * factor * 2));
            double sc = (double)sqrtf(cov);
            sample_x = (float)(((double)dxg - mux) / sc);
            sample_y = (float)(((double)dyg - muy) / sc);
            pdf_area = (float)(exp((double)(-sample_x * sample_x) / 2.0) * exp((double)(-sample_y * sample_y) / 2.0) / (2.0 * kPi) / (double)cov);
        }
    } else {
        LevelInfo L = kernel_scale_level_ratio(M, p);
        float q0 = L.ratio_lo / L.lo, q1 = L.ratio_hi / L.hi;
        float nss = (float)(1.0 / (double)sqrtf(q0 * q0 + q1 * q1));
        scaleX *= nss; scaleY *= nss;
        if (!plus_xy) {
            sample_x = (float)((double)dxg * kSqrt2 * (double)scaleX);
            sample_y = (float)((double)dyg * kSqrt2 * (double)scaleY);
            pdf_area = (float)(exp((double)(-sample_x * sample_x) / 2.0) * exp((double)(-sample_y * sample_y) / 2.0) / (2.0 * kPi) * 2.0 * (double)scaleX * (double)scaleY);
        } else {
            ps.set_state(xy_reseed(info));
            double a, b, c, d;
            rand_normal_2(ps, a, b);
            rand_normal_2(ps, c, d);
            double x1 = a / div, y1 = b / div, x2 = c / div, y2 = d / div;
            float factor = (float)(exp((double)(-tSegment * tSegment) / 4.0) * (0.5 - (double)(tSegment * tSegment) * 0.25));
            double mux = (double)factor * 2.0 * (x1 * (double)L.ratio_lo / (double)L.lo + x2 * (double)L.ratio_hi / (double)L.hi);
            double muy = (double)factor * 2.0 * (y1 * (double)L.ratio_lo / (double)L.lo + y2 * (double)L.ratio_hi / (double)L.hi);
            float cov = (float)(0.5 - (double)(factor * factor * 2));
            double sc = (double)sqrtf(cov);
            sample_x = (float)(((double)dxg - mux) / sc * (double)nss);
            sample_y = (float)(((double)dyg - muy) / sc * (double)nss);
            pdf_area = (float)(exp((double)(-sample_x * sample_x) / 2.0) * exp((double)(-sample_y * sample_y) / 2.0) / (2.0 * kPi * (double)cov) * (double)(nss * nss));
        }
    }
    float cosl = fabsf(dot(normal, s.plane_normal_world));
    float pdf_normal = pdf_area * (s.isect_dist * s.isect_dist) / cosl / s.plane_stretch_jacobian;
    return (float)((double)pdf_normal / (4.0 * (double)dot(normal, v3(-rayDir.x, -rayDir.y, -rayDir.z))));
}

// ---- medium level (SCNM.cpp:102-183, GPM.cpp:221-393) -------------------------------------
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

// intersectGP, SCNM.cpp:102-183
GPIS_DEV bool intersect_gp(const DevModel &M, Realization &noise, V3 pos, V3 dir, float nearT, float farT, float u, MediumState &st, double &t_out)
{
    V3d rd = to_d(dir), p = to_d(pos);
    float step_size = (farT - nearT) / (float)M.min_step;
    if (M.step_size < step_size)
        step_size = M.step_size;
    int gp = 0;
    double t = (double)nearT;
    if (!st.first_scatter)
        conditioning(M, noise, to_f(p), dir, (float)(double)st.last_val, to_f(st.last_aniso), st.info);
    double f0 = (double)evaluate_value(M, noise, to_f(ray_at(p, rd, t)), dir, st.info, gp);
    int sign0 = f0 < 0 ? -1 : 1;
    double pf = f0;
    t = (double)(nearT + step_size * u);
    int step = 0;
    while (t < (double)farT) {
        step++;
        double f_c = (double)evaluate_value(M, noise, to_f(ray_at(p, rd, t)), dir, st.info, gp);
        int signc = f_c < 0 ? -1 : 1;
        if (!st.first_scatter && step == 1) {
            sign0 = signc;
        } else if (signc != sign0) {
            double intp = pf / (pf - f_c);
            double a = t - (double)step_size;
            double t_test, t_prev = lerp_d(a, t, intp);
            for (;;) {
                t_test = lerp_d(a, t, intp);
                double f_test = (double)evaluate_value(M, noise, to_f(ray_at(p, rd, t_test)), dir, st.info, gp);
                int sign_test = f_test < 0 ? -1 : 1;
                if (sign_test == sign0)
                    break;
                intp *= 0.9;
                if (intp <= 0.01) {
                    t_prev = t_test = 0;
                    break;
                }
                t_prev = t_test;
            }
            t = t_prev;
            st.last_gp_id = gp;
            st.scheme = cov_amplitude(M) < 1e-6f ? GPIS_UNI : M.scheme_1d_eff;   // samplingScheme, SCN.cpp:51-58
            st.last_val = 0.0f;
            t_out = t;
            return true;
        }
        pf = f_c;
        t += (double)step_size;
    }
    t = (double)farT;
    st.scheme = GPIS_UNI;
    st.last_val = evaluate_value(M, noise, to_f(ray_at(p, rd, t)), dir, st.info, gp);
    t_out = t;
    return false;
}

// GaussianProcessMedium::transmittance, GPM.cpp:343-393
GPIS_DEV bool transmittance_one(const DevModel &M, Realization &noise, const gpis_ray_in &ray, MediumState &st)
{
    V3 pos = v3(ray.pos[0], ray.pos[1], ray.pos[2]), dir = v3(ray.dir[0], ray.dir[1], ray.dir[2]);
    double startT = (double)ray.near_t;
    float farT = ray.far_t;
    if (!__builtin_isfinite(farT))
        farT = (float)(startT + 2000);
    float maxT = farT;
    double t = (double)maxT;
    V3d ro = to_d(pos), rd = to_d(dir);
    { double inv = 1.0 / length_d(rd); rd.x *= inv; rd.y *= inv; rd.z *= inv; }
    bool exited;
    do {
        exited = !intersect_gp(M, noise, pos, dir, (float)startT, farT, ray.u_jitter, st, t);
        if (t < (double)maxT) {
            V3 g = evaluate_gradient(M, noise, to_f(ray_at(ro, rd, t)), (float)t, dir, st.info);
            st.last_aniso = to_d(g);
            st.first_scatter = false;
            double avg = ((double)g.x + (double)g.y + (double)g.z) / 3.0;
            if (!__builtin_isfinite(avg))
                return false;
        }
        startT = t;
    } while (t < (double)maxT && exited);
    return exited;
}

// GaussianProcessMedium::sampleDistance, GPM.cpp:221-341
GPIS_DEV void sample_distance_one(const DevModel &M, Realization &noise, const gpis_ray_in &ray, gpis_seg_out &out)
{
    MediumState st;
    state_from_ray(ray, st);
    V3 pos = v3(ray.pos[0], ray.pos[1], ray.pos[2]), dir = v3(ray.dir[0], ray.dir[1], ray.dir[2]);
    double startT = (double)ray.near_t;
    float farT = ray.far_t;
    if (!__builtin_isfinite(farT))
        farT = (float)(startT + 2000);
    float maxT = farT;
    out.t = 0.;
    out.sample_t = 0.f; out.continued_t = 0.f;
    for (int c = 0; c < 3; ++c) { out.weight[c] = 0.f; out.continued_weight[c] = 0.f; out.p[c] = 0.f; }
    out.exited = 0; out.ok = 0; out.scheme = GPIS_UNI;
    out.gp_id = st.last_gp_id;
    out.last_val = st.last_val;
    out.aniso[0] = st.last_aniso.x; out.aniso[1] = st.last_aniso.y; out.aniso[2] = st.last_aniso.z;
    if (st.bounce >= M.max_bounces)
        return;
    if (maxT == 0.f) {
        out.weight[0] = out.weight[1] = out.weight[2] = 1.f;
        out.exited = 1;
        V3 pp = pos + dir * out.sample_t;
        out.p[0] = pp.x; out.p[1] = pp.y; out.p[2] = pp.z;
        out.ok = 1;
        return;
    }
    V3d aniso = st.last_aniso;
    if (M.absorption_only) {
        if (maxT == __builtin_huge_valf())
            return;
        out.sample_t = maxT;
        bool vis = transmittance_one(M, noise, ray, st);
        out.weight[0] = out.weight[1] = out.weight[2] = vis ? 1.f : 0.f;
        out.exited = 1;
        aniso = st.last_aniso;
    } else {
        double t = (double)maxT;
        V3d ro = to_d(pos), rd = to_d(dir);
        { double inv = 1.0 / length_d(rd); rd.x *= inv; rd.y *= inv; rd.z *= inv; }
        bool exited;
        do {
            exited = !intersect_gp(M, noise, pos, dir, (float)startT, farT, ray.u_jitter, st, t);
            if (t < (double)maxT) {
                V3 g = evaluate_gradient(M, noise, to_f(ray_at(ro, rd, t)), (float)t, dir, st.info);
                aniso = to_d(g);
                st.last_aniso = aniso;
                st.first_scatter = false;
                double avg = (aniso.x + aniso.y + aniso.z) / 3.0;
                if (!__builtin_isfinite(avg)) {
                    out.aniso[0] = 1.; out.aniso[1] = 0.; out.aniso[2] = 0.;
                    out.t = t; out.exited = exited ? 1 : 0;
                    out.last_val = st.last_val; out.gp_id = st.last_gp_id;
                    return;
                }
            }
            startT = t;
        } while (t < (double)maxT && exited);
        out.t = t;
        out.exited = exited ? 1 : 0;
        if (!exited) {
            double d = aniso.x * (double)dir.x; d += aniso.y * (double)dir.y; d += aniso.z * (double)dir.z;
            double l2 = 0.; l2 += aniso.x * aniso.x; l2 += aniso.y * aniso.y; l2 += aniso.z * aniso.z;
            bool wrong_side = d > 0;
            bool zero_grad = !wrong_side && l2 < (double)0.0000001f;
            if (wrong_side || zero_grad) {
                if (zero_grad) aniso = V3d{1., 0., 0.};
                out.aniso[0] = aniso.x; out.aniso[1] = aniso.y; out.aniso[2] = aniso.z;
                out.last_val = st.last_val; out.gp_id = st.last_gp_id;
                return;
            }
        } else {
            aniso = to_d(evaluate_gradient(M, noise, to_f(ray_at(ro, rd, t)), (float)t, dir, st.info));
        }
        float ft = (float)t;
        out.sample_t = ft < maxT ? ft : maxT;
        out.continued_t = ft;
        for (int c = 0; c < 3; ++c) {
            out.weight[c] = 1.f * M.sigma_s_over_t[c];
            out.continued_weight[c] = 1.f * M.sigma_s_over_t[c];
        }
        st.last_aniso = aniso;
        out.scheme = st.scheme;
    }
    V3 pp = pos + dir * out.sample_t;
    out.p[0] = pp.x; out.p[1] = pp.y; out.p[2] = pp.z;
    out.aniso[0] = aniso.x; out.aniso[1] = aniso.y; out.aniso[2] = aniso.z;
    out.gp_id = st.last_gp_id;
    out.last_val = st.last_val;
    out.ok = 1;
}

}   // namespace gpis
