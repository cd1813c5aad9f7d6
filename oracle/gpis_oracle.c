/*
 * gpis_oracle.c — TEST INFRASTRUCTURE (not product code).  See gpis_oracle.h for the
 * pinning status.  Plain-C restatement of the reference's sparse-convolution GPIS path:
 *
 *   SCN.cpp  = src/core/math/SparseConvolutionNoise.cpp
 *   GPF.cpp  = src/core/math/GPFunctions.cpp        GPF.hpp = its header
 *   SCNM.cpp = src/core/media/SparseConvolutionNoiseMedium.cpp
 *   GPM.cpp  = src/core/media/GaussianProcessMedium.cpp
 *
 * Arithmetic model.  The reference is built SSE4.2-only (CMakeLists.txt:17-20: no AVX, hence
 * no FMA) so every fp32 operation rounds separately; this file is compiled with
 * -ffp-contract=off -mno-fma to match.  Which libm overload each call resolves to was
 * determined by compiling the reference's own include set (minus Boost/FFTW) with g++ 11.4:
 * inside namespace Tungsten `exp/log/pow(float,float)/floor/fabs` of a float resolve to the
 * FLOAT overloads (expf, logf, powf), `sqrt` to std::sqrt of the argument type,
 * `pow(float,int)` to double pow.  Eigen's fixed-size 3-term reductions associate as
 * c0 + (c1 + c2) (redux_novec_unroller), Tungsten's own Vec::dot / lengthSq as
 * (c0 + c1) + c2 (Vec.hpp:134-140, 200-206).  UniformSampler::next3D() is
 * `Vec3f(next1D(), next1D(), next1D())` (UniformSampler.hpp:59-61); g++ evaluates those
 * arguments right to left, so the FIRST draw lands in z and the third in x.  All of these are
 * pinned against the compiled reference by tests/test_oracle_vs_ref.py.
 */
#define _GNU_SOURCE
#include "gpis_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================
 * Integer layer (bit-exact)
 * ==================================================================================== */

#define PRIME32_2 2246822519U
#define PRIME32_3 3266489917U
#define PRIME32_4 668265263U
#define PRIME32_5 374761393U

static inline uint32_t rotl17(uint32_t h) { return (h << 17) | (h >> (32 - 17)); }
static inline uint32_t xx_final(uint32_t h32)
{
    h32 = PRIME32_2 * (h32 ^ (h32 >> 15));
    h32 = PRIME32_3 * (h32 ^ (h32 >> 13));
    return h32 ^ (h32 >> 16);
}
/* MathUtil.hpp:179-187 */
static inline uint32_t xxhash32_1(uint32_t p)
{
    uint32_t h32 = p + PRIME32_5;
    h32 = PRIME32_4 * rotl17(h32);
    return xx_final(h32);
}
/* MathUtil.hpp:190-198 */
static inline uint32_t xxhash32_2(uint32_t x, uint32_t y)
{
    uint32_t h32 = y + PRIME32_5 + x * PRIME32_3;
    h32 = PRIME32_4 * rotl17(h32);
    return xx_final(h32);
}
/* MathUtil.hpp:200-210 */
static inline uint32_t xxhash32_3(uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t h32 = z + PRIME32_5 + x * PRIME32_3;
    h32 = PRIME32_4 * rotl17(h32);
    h32 += y * PRIME32_3;
    h32 = PRIME32_4 * rotl17(h32);
    return xx_final(h32);
}
/* MathUtil.hpp:212-224 */
static inline uint32_t xxhash32_4(uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    uint32_t h32 = w + PRIME32_5 + x * PRIME32_3;
    h32 = PRIME32_4 * rotl17(h32);
    h32 += y * PRIME32_3;
    h32 = PRIME32_4 * rotl17(h32);
    h32 += z * PRIME32_3;
    h32 = PRIME32_4 * rotl17(h32);
    return xx_final(h32);
}

/* UniformSampler.hpp:41-75: PCG-XSH-RR, sequence 0 → increment (0|1) = 1 */
typedef struct { uint64_t state; } pcg32;

static inline uint32_t pcg_next_i(pcg32 *s)
{
    uint64_t old = s->state;
    s->state = old * 6364136223846793005ULL + 1ULL;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
}
/* BitManip.hpp:47-50 */
static inline float normalized_uint(uint32_t i)
{
    union { uint32_t u; float f; } c;
    c.u = (i >> 9u) | 0x3F800000u;
    return c.f - 1.0f;
}
static inline float pcg_next_1d(pcg32 *s) { return normalized_uint(pcg_next_i(s)); }
/* set_state assigns then discards next2D() (UniformSampler.hpp:69-73) */
static inline void pcg_set_state(pcg32 *s, uint64_t state)
{
    s->state = state;
    (void)pcg_next_i(s);
    (void)pcg_next_i(s);
}
/* MathUtil.hpp:226-228 with (a,b,p) = (-1, 1, 0.5) */
static inline float bernoulli_pm1(float rv) { return rv < 0.5f ? -1.f : 1.f; }

/* ======================================================================================
 * Small float / double vector helpers in the reference's evaluation order
 * ==================================================================================== */

typedef struct { float x, y, z; } v3f;
typedef struct { double x, y, z; } v3d;
typedef struct { float v, gx, gy, gz; } v4f; /* (value, gradient) */

static inline v3f v3(float x, float y, float z) { v3f r = {x, y, z}; return r; }
static inline v3f v3_add(v3f a, v3f b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3f v3_sub(v3f a, v3f b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3f v3_scale(v3f a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline v3f v3_div(v3f a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
/* Vec.hpp:200-206: sum = v0*o0; sum += v1*o1; sum += v2*o2 */
static inline float v3_dot(v3f a, v3f b) { float s = a.x * b.x; s += a.y * b.y; s += a.z * b.z; return s; }
/* Vec.hpp:134-140: res = 0; res += v*v ... */
static inline float v3_length_sq(v3f a) { float r = 0.f; r += a.x * a.x; r += a.y * a.y; r += a.z * a.z; return r; }
static inline float v3_length(v3f a) { return sqrtf(v3_length_sq(a)); }
/* Vec.hpp:191-198 */
static inline v3f v3_normalized(v3f a) { float inv = 1.0f / v3_length(a); return v3(a.x * inv, a.y * inv, a.z * inv); }
static inline v3f v3_cross(v3f a, v3f b)
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3d v3d_of(v3f a) { v3d r = {a.x, a.y, a.z}; return r; }
static inline v3f v3f_of(v3d a) { return v3((float)a.x, (float)a.y, (float)a.z); }
static inline double v3d_length(v3d a) { double r = 0.; r += a.x * a.x; r += a.y * a.y; r += a.z * a.z; return sqrt(r); }

static inline v4f v4(float v, float gx, float gy, float gz) { v4f r = {v, gx, gy, gz}; return r; }
static inline v4f v4_add(v4f a, v4f b) { return v4(a.v + b.v, a.gx + b.gx, a.gy + b.gy, a.gz + b.gz); }
static inline v4f v4_scale(v4f a, float s) { return v4(a.v * s, a.gx * s, a.gy * s, a.gz * s); }
static inline v4f v4_scale_l(float s, v4f a) { return v4(s * a.v, s * a.gx, s * a.gy, s * a.gz); }
static inline v4f v4_div(v4f a, float s) { return v4(a.v / s, a.gx / s, a.gy / s, a.gz / s); }

/* TangentFrame(n), TangentFrame.hpp:74-86 */
typedef struct { v3f normal, tangent, bitangent; } frame;
static inline frame frame_from_normal(v3f n)
{
    frame f;
    f.normal = v3_normalized(n);
    float sign = copysignf(1.0f, f.normal.z);
    const float a = -1.0f / (sign + f.normal.z);
    const float b = f.normal.x * f.normal.y * a;
    f.tangent = v3(1.0f + sign * f.normal.x * f.normal.x * a, sign * b, -sign * f.normal.x);
    f.bitangent = v3(b, sign + f.normal.y * f.normal.y * a, -f.normal.y);
    f.tangent = v3_normalized(f.tangent);     /* normalize(): same invLen scheme, Vec.hpp:184-189 */
    f.bitangent = v3_normalized(f.bitangent);
    return f;
}
/* TangentFrame.hpp:88-100 */
static inline v3f frame_to_local(const frame *f, v3f p) { return v3(v3_dot(f->tangent, p), v3_dot(f->bitangent, p), v3_dot(f->normal, p)); }
static inline v3f frame_to_global(const frame *f, v3f p)
{
    /* tangent*p.x() + bitangent*p.y() + normal*p.z(), left to right */
    return v3_add(v3_add(v3_scale(f->tangent, p.x), v3_scale(f->bitangent, p.y)), v3_scale(f->normal, p.z));
}

/* ---- Eigen fixed-size forms: 3-term reductions associate as c0 + (c1 + c2) ---------- */
#define M(m, r, c) ((m)[3 * (r) + (c)])
static inline float sum3_eig(float c0, float c1, float c2) { return c0 + (c1 + c2); }

static inline v3f eig_matvec(const float *m, v3f v)
{
    return v3(sum3_eig(M(m, 0, 0) * v.x, M(m, 0, 1) * v.y, M(m, 0, 2) * v.z),
              sum3_eig(M(m, 1, 0) * v.x, M(m, 1, 1) * v.y, M(m, 1, 2) * v.z),
              sum3_eig(M(m, 2, 0) * v.x, M(m, 2, 1) * v.y, M(m, 2, 2) * v.z));
}
/* dist2_ab, GPF.hpp:75-78: (ab^T * M) * ab */
static inline float eig_dist2_ab(v3f ab, const float *m)
{
    float t0 = sum3_eig(ab.x * M(m, 0, 0), ab.y * M(m, 1, 0), ab.z * M(m, 2, 0));
    float t1 = sum3_eig(ab.x * M(m, 0, 1), ab.y * M(m, 1, 1), ab.z * M(m, 2, 1));
    float t2 = sum3_eig(ab.x * M(m, 0, 2), ab.y * M(m, 1, 2), ab.z * M(m, 2, 2));
    return sum3_eig(t0 * ab.x, t1 * ab.y, t2 * ab.z);
}
static inline float eig_dot_col(v3f ab, const float *m, int c)
{
    return sum3_eig(ab.x * M(m, 0, c), ab.y * M(m, 1, c), ab.z * M(m, 2, c));
}
static inline void mat3_transpose(const float *m, float *out)
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M(out, r, c) = M(m, c, r);
}
static inline void mat3_matmul_eig(const float *a, const float *b, float *out)
{
    float tmp[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
        M(tmp, r, c) = sum3_eig(M(a, r, 0) * M(b, 0, c), M(a, r, 1) * M(b, 1, c), M(a, r, 2) * M(b, 2, c));
    memcpy(out, tmp, sizeof tmp);
}
/* Eigen/src/LU/InverseImpl.h:130-179 (compute_inverse<…,3>) */
static inline float cofactor3(const float *m, int i, int j)
{
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return M(m, i1, j1) * M(m, i2, j2) - M(m, i1, j2) * M(m, i2, j1);
}
static inline void eig_inverse3(const float *m, float *out)
{
    float c0 = cofactor3(m, 0, 0), c1 = cofactor3(m, 1, 0), c2 = cofactor3(m, 2, 0);
    float det = sum3_eig(c0 * M(m, 0, 0), c1 * M(m, 1, 0), c2 * M(m, 2, 0));
    float invdet = 1.0f / det;
    float r[9];
    M(r, 1, 0) = cofactor3(m, 0, 1) * invdet;
    M(r, 1, 1) = cofactor3(m, 1, 1) * invdet;
    M(r, 2, 0) = cofactor3(m, 0, 2) * invdet;
    M(r, 1, 2) = cofactor3(m, 2, 1) * invdet;
    M(r, 2, 1) = cofactor3(m, 1, 2) * invdet;
    M(r, 2, 2) = cofactor3(m, 2, 2) * invdet;
    M(r, 0, 0) = c0 * invdet;
    M(r, 0, 1) = c1 * invdet;
    M(r, 0, 2) = c2 * invdet;
    memcpy(out, r, sizeof r);
}
/* Eigen determinant_impl<…,3>: bruteforce_det3_helper */
static inline float det3_helper(const float *m, int a, int b, int c)
{
    return M(m, 0, a) * (M(m, 1, b) * M(m, 2, c) - M(m, 1, c) * M(m, 2, b));
}
static inline float eig_det3(const float *m)
{
    return det3_helper(m, 0, 1, 2) - det3_helper(m, 1, 0, 2) + det3_helper(m, 2, 0, 1);
}
/* mult(Matrix3f, Vec) GPF.hpp:131-139: plain left-to-right row dot */
static inline v3f mult3(const float *a, v3f b)
{
    return v3(M(a, 0, 0) * b.x + M(a, 0, 1) * b.y + M(a, 0, 2) * b.z,
              M(a, 1, 0) * b.x + M(a, 1, 1) * b.y + M(a, 1, 2) * b.z,
              M(a, 2, 0) * b.x + M(a, 2, 1) * b.y + M(a, 2, 2) * b.z);
}

/* Which libm entry a trigonometric call reaches is part of the result's last bit: glibc 2.35's sin() and cos() resolve to FMA
 * variants on an x86-64 CPU with FMA, its sincos() has one SSE2 body, and gcc turns a sin and a cos of the same argument inside
 * one function into one sincos() call.  The reference's functions are separate (virtual) functions, so what a gcc build of it calls
 * is decided per function: rand_normal_2 and the two Gabor gradients take both (sincos), the two Gabor values take one (cos, sin).
 * These helpers are not inlined, which pins the same choice here whatever the optimiser would do across the restatement's larger
 * functions; csrc/gpis_libm.hpp restates the three entries for the device. */
static __attribute__((noinline)) double libm_sin_alone(double x) { return sin(x); }
static __attribute__((noinline)) double libm_cos_alone(double x) { return cos(x); }
static __attribute__((noinline)) void libm_sin_and_cos(double x, double *sn, double *cs) { *sn = sin(x); *cs = cos(x); }

/* Box–Muller, Gaussian.cpp:21-34; PI is the float constant of Angle.hpp:8 */
static const float PI_F = 3.1415926536f;
static inline void rand_normal_2(pcg32 *s, double *z1, double *z2)
{
    double u1 = pcg_next_1d(s);
    double u2 = pcg_next_1d(s);
    double r = sqrt(-2 * log(1. - u1));
    double x, y;                     /* 2*PI is float*int→float, then * double */
    libm_sin_and_cos(2 * PI_F * u2, &y, &x);
    *z1 = r * x;
    *z2 = r * y;
}

/* ======================================================================================
 * The model: what fromJson/prepareForRender precompute
 * ==================================================================================== */

typedef struct {
    uint64_t n_eval, n_seg;
    char pad[48];
} oracle_counters;

#define ORACLE_MAX_THREADS 256
#define LEVEL_MIN (-24)
#define LEVEL_MAX 24

struct oracle_medium {
    gpis_params P;
    /* SquaredExponentialCovariance::fromJson, GPF.cpp:654-679 */
    float l_conv;
    float l_aniso[3], l_aniso_inv[3];
    float l2w[9], w2l[9], l2w_T[9], w2l_T[9];
    float cov_mtx_inv[9];
    float cov_mtx_inv_det;
    float kernel_scale;   /* _kernelScale ("localScale") */
    uint32_t n_impulses;  /* uint(impulseDensity), SCN.cpp:380 */
    int activate_conditioning, scheme_1d_eff, multi_res; /* SCN.cpp:21-30 */
    float base;           /* SCN.cpp:32 */
    /* ProceduralNoiseVec::fromJson, GPF.hpp:765-776 */
    double ls_scale, ls_offset;
    float gabor_a, gabor_f, gabor_omega[3];   /* _a = 1/a_inv, _f = 1/f_inv, normalised omega (GPF.cpp:1088-1096) */
    float ls_maxval;      /* maxVal(), GPF.cpp:124-138 */
    /* GaussianProcessMedium::prepareForRender, GPM.cpp:152-158 */
    float sigma_a[3], sigma_s[3], sigma_t[3];
    int absorption_only;
    double lin_dir[2][3]; /* LinearMean direction after normalize() */
    int threads;
    /* GridNonstationaryCovariance: the voxel grid (oracle_set_variance_grid); grid_vox == NULL: getUnscaledVariance = 1 */
    gpis_variance_grid grid;
    float *grid_vox;
    oracle_counters counters[ORACLE_MAX_THREADS];
};

typedef struct {
    const oracle_medium *m;
    gpis_cond_coeff c3;   /* coeff_3D, SCN.hpp:7-13 */
    gpis_cond_coeff c1;   /* coeff_1D, SCN.hpp:15-21 */
    oracle_counters *cnt;
} realization;

/* RayInfo, MediumSample.hpp:14-18 */
typedef struct { uint32_t px, py, spp, seg, scene_seed; float t; } ray_info;

static __thread char g_err[256];

/* ---- means ------------------------------------------------------------------------- */

/* MeanFunction::mean, GPF.hpp:887-889, 933-935, 992-994 */
static double mean_eval(const oracle_medium *m, int which, v3d a)
{
    const gpis_mean *mu = which ? &m->P.mean_additional : &m->P.mean;
    switch (mu->type) {
    case GPIS_MEAN_HOMOGENEOUS:
        return mu->offset;
    case GPIS_MEAN_SPHERICAL: {
        v3d d = {a.x - mu->center[0], a.y - mu->center[1], a.z - mu->center[2]};
        return v3d_length(d) - mu->radius;
    }
    default: { /* linear: max((a - ref).dot(dir) * scale, (double)min) */
        const double *dir = m->lin_dir[which];
        double dx = a.x - mu->center[0], dy = a.y - mu->center[1], dz = a.z - mu->center[2];
        double dot = dx * dir[0]; dot += dy * dir[1]; dot += dz * dir[2];
        double v = dot * mu->scale;
        double mn = (double)mu->min;
        return v > mn ? v : mn;   /* Tungsten max(a,b) = a > b ? a : b (MathUtil.hpp) */
    }
    }
}
/* MeanFunction::dmean_da, GPF.hpp:891-893, 937-939, 996-1003 */
static v3d mean_grad(const oracle_medium *m, int which, v3d a)
{
    const gpis_mean *mu = which ? &m->P.mean_additional : &m->P.mean;
    v3d r = {0., 0., 0.};
    switch (mu->type) {
    case GPIS_MEAN_HOMOGENEOUS:
        return r;
    case GPIS_MEAN_SPHERICAL: {
        v3d d = {a.x - mu->center[0], a.y - mu->center[1], a.z - mu->center[2]};
        double inv = 1.0 / v3d_length(d);
        r.x = d.x * inv; r.y = d.y * inv; r.z = d.z * inv;
        return r;
    }
    default: {
        const double *dir = m->lin_dir[which];
        double dx = a.x - mu->center[0], dy = a.y - mu->center[1], dz = a.z - mu->center[2];
        double dot = dx * dir[0]; dot += dy * dir[1]; dot += dz * dir[2];
        if (dot * mu->scale < mu->min)
            return r;
        r.x = dir[0] * mu->scale; r.y = dir[1] * mu->scale; r.z = dir[2] * mu->scale;
        return r;
    }
    }
}
/* GaussianProcess::mean_weight_space, GaussianProcess.cpp:379-393 (_id = 0, _id_additional = 1) */
static void mean_weight_space(const oracle_medium *m, v3d p, double *mean, double *id)
{
    *mean = mean_eval(m, 0, p);
    *id = 0;
    if (m->P.has_mean_additional) {
        double add = mean_eval(m, 1, p);
        if (add < *mean) {
            *mean = add;
            *id = 1;
        }
    }
}

/* ---- covariance: SE kernel + (optional) procedural non-stationary wrapper ------------ */

/* ProceduralNoise::operator() / one component of ProceduralNoiseVec::operator() (all three are equal for these noise
 * types), GPF.cpp:43-103, with _const = 1 and _scale = 1/(_end - _start), _offset = -_start * _scale (GPF.hpp:684-687) */
static double ramp_unit(double coord, double start, double end, double mn, double mx)
{
    const double c = 1.;
    double scale = 1.0 / (end - start), offset = -start * scale;
    double lo = mn + c, hi = mx + c;
    double u = coord * scale + offset;
    u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);                 /* clamp */
    double a = log(lo * lo), b = log(hi * hi);
    double l = a * (1.0 - u) + b * u;                        /* lerp, MathUtil.hpp:90-94 */
    return sqrt(exp(l));
}
static double ramp_eval(int type, double mn, double mx, double start, double end, double mn2, double mx2, double start2, double end2, v3d p)
{
    const double c = 1.;
    if (type == GPIS_RAMP_BOTTOM_TOP_LEFT_RIGHT) {           /* GPF.cpp:96-103: the two factors are narrowed to float */
        float bottomTop = (float)ramp_unit(p.y, start, end, mn, mx);
        float leftRight = (float)ramp_unit(p.x, start2, end2, mn2, mx2);
        return (double)(bottomTop * leftRight) - c * c;
    }
    double coord = type == GPIS_RAMP_BOTTOM_TOP ? p.y : (type == GPIS_RAMP_LEFT_RIGHT ? p.x : p.z);
    return ramp_unit(coord, start, end, mn, mx) - c;
}
/* ---- "sandstone" / "rust": fbm over 3D simplex noise, math/SdfFunctions.cpp:199-296.  Pinned bit for bit against the
 * reference's own SdfFunctions.cpp compiled in place (oracle/_ref: ref_random3 / ref_simplex3d / ref_fbm,
 * tests/test_oracle_vs_ref.py).  `sin` of the float dot product resolves to the double function there. */
static inline float sdf_dot3(v3f a, v3f b) { float r = a.x * b.x; r += a.y * b.y; r += a.z * b.z; return r; }
static v3f sdf_random3(v3f c)                                  /* SdfFunctions.cpp:199-208 */
{
    const v3f k = {17.0f, 59.4f, 15.0f};
    float j = (float)(4096.0 * libm_sin_alone((double)sdf_dot3(c, k)));
    v3f r; double v;
    v = 512.0 * (double)j; r.z = (float)(v - floor(v));
    j = (float)((double)j * .125);
    v = 512.0 * (double)j; r.x = (float)(v - floor(v));
    j = (float)((double)j * .125);
    v = 512.0 * (double)j; r.y = (float)(v - floor(v));
    r.x = r.x - 0.5f; r.y = r.y - 0.5f; r.z = r.z - 0.5f;
    return r;
}
static float sdf_simplex3d(v3f p)                              /* SdfFunctions.cpp:228-273 */
{
    const float F3 = 0.3333333f, G3 = 0.1666667f;              /* :211-212 */
    const v3f f3 = {F3, F3, F3}, g3 = {G3, G3, G3};
    const float pf = sdf_dot3(p, f3);
    v3f s = {floorf(p.x + pf), floorf(p.y + pf), floorf(p.z + pf)};
    const float sg = sdf_dot3(s, g3);
    const v3f x = {(p.x - s.x) + sg, (p.y - s.y) + sg, (p.z - s.z) + sg};
    const v3f e = {(x.x - x.y) < 0.0f ? 0.f : 1.f, (x.y - x.z) < 0.0f ? 0.f : 1.f, (x.z - x.x) < 0.0f ? 0.f : 1.f};   /* step(0, x - x.yzx) */
    const v3f ez = {e.z, e.x, e.y};
    const v3f i1 = {e.x * (1.0f - ez.x), e.y * (1.0f - ez.y), e.z * (1.0f - ez.z)};
    const v3f i2 = {1.0f - ez.x * (1.0f - e.x), 1.0f - ez.y * (1.0f - e.y), 1.0f - ez.z * (1.0f - e.z)};
    const float g2 = 2.0f * G3, g3s = 3.0f * G3;
    const v3f x1 = {(x.x - i1.x) + G3, (x.y - i1.y) + G3, (x.z - i1.z) + G3};
    const v3f x2 = {(x.x - i2.x) + g2, (x.y - i2.y) + g2, (x.z - i2.z) + g2};
    const v3f x3 = {(x.x - 1.0f) + g3s, (x.y - 1.0f) + g3s, (x.z - 1.0f) + g3s};
    float w[4] = {sdf_dot3(x, x), sdf_dot3(x1, x1), sdf_dot3(x2, x2), sdf_dot3(x3, x3)}, d[4];
    for (int i = 0; i < 4; ++i) { const float t = 0.6f - w[i]; w[i] = t < 0.0f ? 0.0f : t; }     /* max(0.6 - w, 0) */
    const v3f s1 = {s.x + i1.x, s.y + i1.y, s.z + i1.z}, s2 = {s.x + i2.x, s.y + i2.y, s.z + i2.z}, s3 = {s.x + 1.0f, s.y + 1.0f, s.z + 1.0f};
    d[0] = sdf_dot3(sdf_random3(s), x);
    d[1] = sdf_dot3(sdf_random3(s1), x1);
    d[2] = sdf_dot3(sdf_random3(s2), x2);
    d[3] = sdf_dot3(sdf_random3(s3), x3);
    for (int i = 0; i < 4; ++i) { w[i] *= w[i]; w[i] *= w[i]; d[i] *= w[i]; }
    float r = d[0] * 52.0f; r += d[1] * 52.0f; r += d[2] * 52.0f; r += d[3] * 52.0f;
    return r;
}
static double sdf_fbm(v3d uv, int octaves)                     /* SdfFunctions.cpp:276-296 */
{
    const float gain = 0.65f, lacunarity = 2.1042f;
    float total, frequency = 0.5f, amplitude = gain;
    const v3f u = {(float)(uv.x * 5.0), (float)(uv.y * 5.0), (float)(uv.z * 5.0)};
    total = sdf_simplex3d(u);
    for (int i = 0; i < octaves; i++) {
        const v3f q = {u.x * frequency, u.y * frequency, u.z * frequency};
        total += sdf_simplex3d(q) * amplitude;
        frequency *= lacunarity;
        amplitude *= gain;
    }
    total = (float)(((double)total + 2.0) / 4.0);
    return (double)total;
}
static inline v3d v3d_adds(v3d a, double s) { v3d r = {a.x + s, a.y + s, a.z + s}; return r; }
static inline v3d v3d_muls(v3d a, double s) { v3d r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline double clamp01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }     /* MathUtil clamp: min(max(v, lo), hi) */
static inline double smooth_step(double e0, double e1, double x) { x = clamp01((x - e0) / (e1 - e0)); return x * x * (3.0 - 2.0 * x); }   /* MathUtil.hpp:107-112 */
/* the fbm cascades shared by ProceduralNoise (octaves 2) and ProceduralNoiseVec (octaves 10), GPF.cpp:70-83, 104-117 */
static double noise_sandstone_f(v3d p, int oct)
{
    const v3d ps = v3d_muls(p, 0.3);
    return sdf_fbm(v3d_adds(ps, sdf_fbm(v3d_adds(ps, sdf_fbm(ps, oct)), oct)), oct);
}
/* ProceduralNoise::operator() (the SCALAR fields "var" and "aniso"), GPF.cpp:53-85 */
static double ramp_of(const gpis_ramp *r, v3d p)
{
    if (r->type == GPIS_NOISE_SANDSTONE) {
        const double f = noise_sandstone_f(p, 2);
        const double colx = sqrt((f * 1.9) * 1.2) - 0.35;
        const double t = clamp01(colx);
        return r->min * (1.0 - t) + r->max * t;                                   /* lerp(_min, _max, .) */
    }
    if (r->type == GPIS_NOISE_RUST) {
        const v3d ps = v3d_muls(p, 2.0);
        const double f = smooth_step(0.4, 0.6, sdf_fbm(v3d_adds(ps, sdf_fbm(v3d_muls(ps, .1), 2) * 0.4), 2) - sdf_fbm(v3d_muls(ps, 25.), 2) * 0.1);
        const double t = clamp01(f);
        return r->min * (1.0 - t) + r->max * t;
    }
    return ramp_eval(r->type, r->min, r->max, r->start, r->end, r->min2, r->max2, r->start2, r->end2, p);
}
/* ProceduralNoiseVec::operator() (the VECTOR fields "ls", mean "color" / "emission"), GPF.cpp:87-120: the ramp noises have
 * three equal components, sandstone / rust do not (and use 10 octaves, and rust ADDS its fine-scale term) */
static void noise_vec(int type, double mn, double mx, double start, double end, double mn2, double mx2, double start2, double end2, v3d p, double out[3])
{
    if (type == GPIS_NOISE_SANDSTONE) {
        const double f = noise_sandstone_f(p, 10);
        const double col[3] = {sqrt((f * 1.9) * 1.2) - 0.35, sqrt((f * 0.7) * 1.2) - 0.35, sqrt((f * 0.25) * 1.2) - 0.35};
        for (int c = 0; c < 3; ++c) out[c] = clamp01(col[c] * 0.2);
        return;
    }
    if (type == GPIS_NOISE_RUST) {
        const v3d ps = v3d_muls(p, 2.0);
        const double f = smooth_step(0.4, 0.6, sdf_fbm(v3d_adds(ps, sdf_fbm(v3d_muls(ps, .1), 10) * 0.4), 10) + sdf_fbm(v3d_muls(ps, 25.), 10) * 0.1);
        const double a[3] = {0.278, 0.212, 0.141};
        for (int c = 0; c < 3; ++c) out[c] = a[c] * (1.0 - f) + 1. * f;            /* Vec lerp: a*(1 - ratio) + b*ratio, MathUtil.hpp:101-105 */
        return;
    }
    out[0] = out[1] = out[2] = ramp_eval(type, mn, mx, start, end, mn2, mx2, start2, end2, p);
}
static void ramp_vec_of(const gpis_ramp *r, v3d p, double out[3])
{
    noise_vec(r->type, r->min, r->max, r->start, r->end, r->min2, r->max2, r->start2, r->end2, p, out);
}
/* ---- GridNonstationaryCovariance, GPF.cpp:1386-1427 over VdbGrid::density, VdbGrid.cpp:405-431 ---------------------------
 * OpenVDB's samplers restated from tools/Interpolation.h (absent dependency: parity unpinned for this lookup):
 *   PointSampler: the voxel at floor(p + 0.5);
 *   BoxSampler: the 8 voxels at floor(p) + {0,1}^3, weights uvw = p - floor(p) in double, interpolated along z, then y, then x
 *   with  lerp(a, b, w) = float(a + float((b - a) * w)). */
static float grid_voxel(const oracle_medium *m, long i, long j, long k)
{
    const gpis_variance_grid *g = &m->grid;
    i -= g->origin[0]; j -= g->origin[1]; k -= g->origin[2];
    if (i < 0 || j < 0 || k < 0 || i >= g->dims[0] || j >= g->dims[1] || k >= g->dims[2]) return 0.f;
    return m->grid_vox[(size_t)i + (size_t)g->dims[0] * ((size_t)j + (size_t)g->dims[1] * (size_t)k)];
}
static float grid_lerp(float a, float b, double w) { double temp = (double)(b - a) * w; return a + (float)temp; }
static float grid_density(const oracle_medium *m, v3f p)
{
    const gpis_variance_grid *g = &m->grid;
    float q[3] = {p.x, p.y, p.z};
    for (int c = 0; c < 3; ++c) {                                   /* clamp(p, bounds.min + 2, bounds.max - 3): min(max(p, lo), hi) */
        float lo = g->bounds_min[c] + 2, hi = g->bounds_max[c] - 3;
        float v = q[c] < lo ? lo : q[c];
        q[c] = v < hi ? v : hi;
    }
    double x = q[0], y = q[1], z = q[2];
    if (g->interpolate == 0)
        return grid_voxel(m, (long)floor(x + 0.5), (long)floor(y + 0.5), (long)floor(z + 0.5));
    double fx = floor(x), fy = floor(y), fz = floor(z);
    long i = (long)fx, j = (long)fy, k = (long)fz;
    double u = x - fx, v = y - fy, w = z - fz;
    float d[2][2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int c = 0; c < 2; ++c) d[a][b][c] = grid_voxel(m, i + a, j + b, k + c);
    return grid_lerp(grid_lerp(grid_lerp(d[0][0][0], d[0][0][1], w), grid_lerp(d[0][1][0], d[0][1][1], w), v),
                     grid_lerp(grid_lerp(d[1][0][0], d[1][0][1], w), grid_lerp(d[1][1][0], d[1][1][1], w), v), u);
}
/* getUnscaledVariance, GPF.cpp:1386-1391: (density(invNaturalTransform * Vec3f(p)) + offset) * scale; Mat4f * Vec3f as Mat4f.hpp:329-336 */
static float grid_unscaled_variance(const oracle_medium *m, v3d pd)
{
    if (!m->grid_vox) return 1.f;
    const float *T = m->grid.inv_natural_transform;
    float x = (float)pd.x, y = (float)pd.y, z = (float)pd.z;
    v3f q = v3(T[0] * x + T[1] * y + T[2] * z + T[3], T[4] * x + T[5] * y + T[6] * z + T[7], T[8] * x + T[9] * y + T[10] * z + T[11]);
    return (grid_density(m, q) + m->P.grid_offset) * m->P.grid_scale;
}
/* getVariance, GPF.cpp:1393-1403 */
static double grid_variance(const oracle_medium *m, v3d p)
{
    float amplitude = grid_unscaled_variance(m, p);
    if (!m->P.grid_surf_vol_amp_separate) return amplitude;
    if (amplitude < m->P.grid_surf_vol_amp_thresh) return amplitude * m->P.grid_surf_amp_scale;
    return amplitude * m->P.grid_vol_amp_scale;
}
/* getKernelScale, GPF.cpp:1409-1420 */
static float grid_kernel_scale(const oracle_medium *m, v3f p)
{
    if (m->P.grid_surf_vol_amp_separate) {
        float amplitude = grid_unscaled_variance(m, v3d_of(p));
        return amplitude < m->P.grid_surf_vol_amp_thresh ? m->P.grid_surf_ls_scale : m->P.grid_vol_ls_scale;
    }
    return 1.f;
}

/* sparseConvNoiseLateralScale: GPF.cpp:607-609 / 1219-1221 → getKernelScale GPF.cpp:1729-1735 */
static float cov_lateral_scale(const oracle_medium *m, v3f p)
{
    if (!m->P.nonstationary)
        return 1.0f;
    if (m->P.grid_nonstationary)
        return grid_kernel_scale(m, p);
    const gpis_params *P = &m->P;
    double ls[3];
    noise_vec(P->ls_ramp_type, P->ls_min, P->ls_max, P->ls_start, P->ls_end, P->ls_min2, P->ls_max2, P->ls_start2, P->ls_end2, v3d_of(p), ls);
    const float lx = (float)ls[0], ly = (float)ls[1], lz = (float)ls[2];
    const float mxy = lx < ly ? ly : lx;                                          /* max(a, b): a < b ? b : a */
    return mxy < lz ? lz : mxy;
}
/* sparseConvNoiseMaxLateralScale, GPF.cpp:1737-1741 */
static float cov_max_lateral_scale(const oracle_medium *m) { return m->P.nonstationary ? m->ls_maxval : 1.f; }
/* nonStationarySplattingKernelScale: GPF.hpp:1493 / GPF.cpp:1223-1229 */
static float cov_ns_scale(const oracle_medium *m, v3f p)
{
    if (!m->P.nonstationary || m->P.multi_resolution_grid)
        return 1.f;
    return cov_lateral_scale(m, p) / cov_max_lateral_scale(m);
}
/* worldSamplingSpatialScale: GPF.hpp:1494 / GPF.cpp:1231-1233 */
static float cov_world_sampling_scale(const oracle_medium *m) { return m->P.nonstationary ? cov_max_lateral_scale(m) : 1.f; }
/* sparseConvNoiseAmplitude: GPF.cpp:611-613,711 (stationary) / 1235-1237: getVariance(Vec3d(p)) * sigma, with
 * getVariance = the "var" field or 1 (GPF.cpp:1638-1641) */
static float cov_amplitude(const oracle_medium *m, v3f p)
{
    if (!m->P.nonstationary)
        return m->P.sigma;
    if (m->P.grid_nonstationary)
        return (float)(grid_variance(m, v3d_of(p)) * m->P.sigma);
    double var = m->P.var.enabled ? ramp_of(&m->P.var, v3d_of(p)) : 1.0;
    return (float)(var * m->P.sigma);
}
/* ---- Matérn (v = 0.5, 1.5, 2.5) and Gabor kernels: GPF.cpp:866-1214 ---------------------------------------------
 * The reference evaluates these in double (exp / pow / cos / sin of libm) and narrows to float. */
/* Modified Bessel functions of the second kind K0(x), K1(x), x > 0, for the Matern v = 3/2 splatting kernel (GPF.cpp:1053-1056,
 * 1071-1074 call boost::math::cyl_bessel_k, which is neither vendored nor installed: PARITY UNPINNED VS BOOST; these are checked
 * against the Wronskian I0 K1 + I1 K0 = 1/x and against scipy to 1e-14 in tests/).  x <= 2: the ascending series
 *   K0 = -(ln(x/2) + g) I0 + sum_k (x^2/4)^k / (k!)^2 H_k,   K1 = 1/x + ln(x/2) I1 - (x/4) sum_k (psi(k+1) + psi(k+2)) (x^2/4)^k / (k! (k+1)!);
 * x > 2: Steed's algorithm on the continued fraction CF2 (Temme 1975), which converges in < 40 terms there. */
static void bessel_k01(double x, double *k0, double *k1)
{
    const double EULER = 0.57721566490153286061;
    if (x <= 2.0) {
        const double q = 0.25 * x * x, lg = log(0.5 * x);
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
        const double rk0 = sqrt(3.14159265358979323846 / (2.0 * x)) * exp(-x) / s;
        *k0 = rk0;
        *k1 = rk0 * (x + 0.5 - h) / x;
    }
}
static float other_kernel_radius(const oracle_medium *m, int isIdentity, float localScale)
{
    float scale_factor = m->kernel_scale;
    if (m->P.kernel_type == GPIS_KERNEL_MATERN) {                        /* GPF.cpp:1020-1025 */
        if (isIdentity)
            return scale_factor;
        float la[3];
        for (int i = 0; i < 3; ++i) {                                    /* fromJson, GPF.cpp:873-875 + filterWithZero */
            la[i] = m->P.length_scale / sqrtf(m->P.aniso[i]);
            if (isinf(la[i]) || isnan(la[i])) la[i] = 0;
        }
        float mx = la[0] > la[1] ? la[0] : la[1];
        mx = mx > la[2] ? mx : la[2];
        return (float)(scale_factor * localScale * sqrt(2) / 2 * mx);
    }
    if (m->P.kernel_type == GPIS_KERNEL_GABOR_ANISO)                     /* GPF.cpp:1127-1130 */
        return (float)(scale_factor * sqrt(2) / 2 * 1.0 / m->gabor_a);
    return (float)(scale_factor * sqrt(2) / 4 * 1.0 / m->gabor_a);       /* GPF.cpp:1192-1195 */
}
static float other_variance3d(const oracle_medium *m, float impulseDensity, float kernelRadius)
{
    double impulseDensityUnitArea = impulseDensity / (kernelRadius * kernelRadius * kernelRadius);
    double integralKernelSquared;
    const float l = m->P.length_scale, a = m->gabor_a, f = m->gabor_f;
    if (m->P.kernel_type == GPIS_KERNEL_MATERN) {                        /* GPF.cpp:1029-1046 */
        if (m->P.matern_v == 0.5) integralKernelSquared = 2.0 * M_PI * l;
        else if (m->P.matern_v == 1.5) integralKernelSquared = pow(M_PI * l, 3) / (24 * sqrt(3));
        else integralKernelSquared = M_PI * pow(l, 3) / (5 * sqrt(5));
    } else if (m->P.kernel_type == GPIS_KERNEL_GABOR_ANISO) {            /* GPF.cpp:1134-1138 */
        float q = f / a;
        integralKernelSquared = pow(1.0 / a, 3) * (1 + exp(-2.0 * M_PI * (q * q))) / (4 * sqrt(2));
    } else {                                                             /* GPF.cpp:1199-1203 */
        integralKernelSquared = 2 * sqrt(2) * M_PI * (f * f) / a * (1 - exp(-2 * M_PI * f / (a * a)));
    }
    return (float)(impulseDensityUnitArea * integralKernelSquared);
}
static v4f other_splat3d(const oracle_medium *m, v3f ab)
{
    const float l = m->P.length_scale, a = m->gabor_a, f = m->gabor_f;
    if (m->P.kernel_type == GPIS_KERNEL_MATERN) {                        /* GPF.cpp:1048-1082 */
        double abLen = sqrtf(v3_length_sq(ab));                          /* Vec3f::length() */
        float val;
        float gs;
        if (m->P.matern_v == 0.5) {
            val = (float)(exp(-abLen / l) / abLen);
            gs = -(float)(exp(-abLen / l) * (1 / pow(abLen, 3) - 1 / (pow(abLen, 2) * l)));
        } else if (m->P.matern_v == 1.5) {                                /* GPF.cpp:1053-1056, 1071-1074 */
            double r_scl = sqrt(3.) * abLen / l, k0, k1;
            bessel_k01(r_scl, &k0, &k1);
            val = (float)k0;
            gs = -(float)(k1 * sqrt(3.) / l / abLen);
        } else {
            val = (float)exp(-sqrt(5.) * abLen / l);
            gs = -(float)(exp(-sqrt(5.) * abLen / l) * sqrt(5.) / l / abLen);
        }
        return v4(val, gs * ab.x, gs * ab.y, gs * ab.z);
    }
    if (m->P.kernel_type == GPIS_KERNEL_GABOR_ANISO) {                   /* GPF.cpp:1140-1150 */
        v3f om = v3(m->gabor_omega[0], m->gabor_omega[1], m->gabor_omega[2]);
        float od = v3_dot(om, ab);
        double sph, cph;                                                 /* value: GPF.cpp:1140-1142 (cos alone); gradient: 1144-1150 (both) */
        float val = (float)(exp(-M_PI * (a * a) * v3_length_sq(ab)) * libm_cos_alone(2.f * M_PI * f * od));
        libm_sin_and_cos(2.f * M_PI * f * od, &sph, &cph);
        float A = (float)exp(-M_PI * (a * a) * v3_length_sq(ab));
        float B = (float)cph;
        float c1 = -(float)(A * sph * 2.f * M_PI * f);
        float c2 = (float)(B * A * 2.f * M_PI * (a * a));
        v3f g = v3_sub(v3_scale(om, c1), v3_scale(ab, c2));
        return v4(val, g.x, g.y, g.z);
    }
    {                                                                    /* GPF.cpp:1205-1214 */
        float r = sqrtf(v3_length_sq(ab));
        float ar = a * r;
        double s2, c2;                                                   /* value: sin alone; gradient: both */
        float val = (float)(exp(-M_PI * (ar * ar)) * 2 * f / r * libm_sin_alone(2 * M_PI * f * r));
        libm_sin_and_cos(2 * M_PI * f * r, &s2, &c2);
        float gs = (float)(2 * f * exp(-M_PI * (a * a) * v3_length_sq(ab)) *
                           (-s2 / pow(r, 3) - 2 * M_PI * (a * a) * s2 / r + 2 * M_PI * f * c2 / (r * r)));
        return v4(val, gs * ab.x, gs * ab.y, gs * ab.z);
    }
}

/* SquaredExponentialCovariance::splattingKernelRadius, GPF.cpp:696-709 */
static float se_kernel_radius(const oracle_medium *m, int isIdentity, float localScale)
{
    if (m->P.kernel_type != GPIS_KERNEL_SQUARED_EXPONENTIAL)
        return other_kernel_radius(m, isIdentity, localScale);
    float scale_factor = m->kernel_scale;
    if (isIdentity)
        return scale_factor;
    float mtx_factor;
    if (!m->P.use_aniso_mtx) {
        float a = m->l_aniso[0] > m->l_aniso[1] ? m->l_aniso[0] : m->l_aniso[1];
        mtx_factor = a > m->l_aniso[2] ? a : m->l_aniso[2];
    } else {
        /* end_pt = col0 + col1 + col2 (Eigen cwise sum, left to right) */
        float e0 = (M(m->l2w, 0, 0) + M(m->l2w, 0, 1)) + M(m->l2w, 0, 2);
        float e1 = (M(m->l2w, 1, 0) + M(m->l2w, 1, 1)) + M(m->l2w, 1, 2);
        float e2 = (M(m->l2w, 2, 0) + M(m->l2w, 2, 1)) + M(m->l2w, 2, 2);
        float a = e0 > e1 ? e0 : e1;
        mtx_factor = a > e2 ? a : e2;
    }
    return scale_factor * localScale * mtx_factor;
}
/* splattingKernelRadius: SE directly, or NonstationaryCovariance GPF.cpp:1245-1249 */
static float cov_kernel_radius(const oracle_medium *m, int isIdentity, float localScale)
{
    if (m->P.nonstationary) {
        localScale = (float)(localScale * (m->P.multi_resolution_grid ? 1.0 : (double)cov_max_lateral_scale(m)));
        localScale *= m->P.aniso_field.enabled ? 1.5f : 1.f; /* sparseConvNoiseMaxAnisotropyScale(), GPF.cpp:1743-1747 */
    }
    return se_kernel_radius(m, isIdentity, localScale);
}
/* SE::sparseConvNoiseVariance3D, GPF.cpp:741-754 */
static float se_variance3d(const oracle_medium *m, float impulseDensity, float kernelRadius, int isIdentity,
                           float globalScale, float localScale)
{
    if (m->P.kernel_type != GPIS_KERNEL_SQUARED_EXPONENTIAL)
        return other_variance3d(m, impulseDensity, kernelRadius);
    double impulseDensityUnitArea = impulseDensity / (kernelRadius * kernelRadius * kernelRadius);
    double covDeterminantSqrt = 1.0;
    if (!isIdentity) {
        if (!m->P.use_aniso_mtx)
            covDeterminantSqrt = m->l_aniso[0] * m->l_aniso[1] * m->l_aniso[2];
        else
            covDeterminantSqrt = 1.0 / sqrtf(m->cov_mtx_inv_det);  /* sqrt(float) → float, 1.0/float → double */
        covDeterminantSqrt *= pow(globalScale, 3);
    }
    covDeterminantSqrt *= pow(localScale, 3);
    double integralKernelSquared = pow(M_PI, 1.5) * covDeterminantSqrt;
    return (float)(impulseDensityUnitArea * integralKernelSquared);
}
/* SE::sparseConvNoiseVariance1D, GPF.cpp:756-760 */
static float se_variance1d(float impulseDensity, float kernelRadius, float localScale)
{
    double impulseDensityUnitArea = impulseDensity / kernelRadius;
    double integralKernelSquared = sqrt(M_PI) * localScale;
    return (float)(impulseDensityUnitArea * integralKernelSquared);
}
/* sparseConvNoiseVariance3D(p, …): GPF.cpp:615-617 / 1251-1254 */
static float cov_variance3d(const oracle_medium *m, v3f p, float dens, float R, int isIdentity, float globalScale)
{
    if (!m->P.nonstationary)
        return se_variance3d(m, dens, R, isIdentity, globalScale, 1.0f);
    float localScale = m->P.multi_resolution_grid ? 1.0f : cov_ns_scale(m, p);
    return se_variance3d(m, dens, R, isIdentity, globalScale, localScale);
}
/* sparseConvNoiseVariance1D(p, …): GPF.cpp:619-621 / 1256-1259 */
static float cov_variance1d(const oracle_medium *m, v3f p, float dens, float R)
{
    if (!m->P.nonstationary)
        return se_variance1d(dens, R, 1.0f);
    return se_variance1d(dens, R, cov_ns_scale(m, p));
}
/* transform*: GPF.cpp:631-641 (stationary) / 1261-1279 (wrapper picks the scale) */
static float cov_xf_scale(const oracle_medium *m, float globalScale)
{
    if (!m->P.nonstationary)
        return globalScale;
    return m->P.multi_resolution_grid ? globalScale : cov_max_lateral_scale(m);
}
static v3f cov_pos_world_to_local(const oracle_medium *m, v3f v, float s) { return v3_div(eig_matvec(m->w2l, v), cov_xf_scale(m, s)); }
static v3f cov_grad_world_to_local(const oracle_medium *m, v3f v, float s) { return v3_scale(eig_matvec(m->l2w_T, v), cov_xf_scale(m, s)); }
static v3f cov_grad_local_to_world(const oracle_medium *m, v3f v, float s) { return v3_div(eig_matvec(m->w2l_T, v), cov_xf_scale(m, s)); }
/* localToWorldInvTransposeMatrix(p): GPF.cpp:629,649 / 1281-1284 */
static void cov_l2w_inv_transpose(const oracle_medium *m, float *out)
{
    float scale = 1.f;
    if (m->P.nonstationary)
        scale = (float)(m->P.multi_resolution_grid ? 1.0 : (double)cov_max_lateral_scale(m));
    for (int i = 0; i < 9; ++i)
        out[i] = m->P.nonstationary ? m->w2l_T[i] / scale : m->w2l_T[i];
}
/* getNonstationaryCovSplatCov1D: GPF.hpp:1213 / GPF.cpp:1715-1727 (no aniso field) */
static float cov_splat_cov_1d(const oracle_medium *m, v3f pq, v3f pc)
{
    if (!m->P.nonstationary)
        return 1.0f;
    float a = cov_ns_scale(m, pq), b = cov_ns_scale(m, pc);
    float scale = (float)(0.5 * (a * a + b * b));
    return sqrtf(scale);
}

/* ProceduralNonstationaryCovariance::getNonstationaryAniso3D, GPF.cpp:1678-1689: the "aniso" field gives an angle in units
 * of pi/2; the matrix is compute_ansio_full(angle, (1.5, 1/1.5, 1)) = (S V)^T (S V) with V the rotation block, GPF.cpp:27-40
 * (_anisotropyOnAxis = 1.5, GPF.hpp:2372).  Returns 0 without a field (the reference's null pointer). */
static int cov_ns_aniso3d(const oracle_medium *m, v3f p_world, float *out)
{
    if (!m->P.aniso_field.enabled)
        return 0;
    const float PI_HALF = PI_F * 0.5f;                               /* Angle.hpp:9 */
    float angle = (float)(ramp_of(&m->P.aniso_field, v3d_of(p_world)) * PI_HALF);
    float V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, S[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, SV[9], SVt[9];
    M(V, 0, 0) = cosf(angle);
    M(V, 0, 1) = sinf(angle);
    M(V, 1, 0) = -sinf(angle);
    M(V, 1, 1) = cosf(angle);
    M(S, 0, 0) = 1.5f; M(S, 1, 1) = 1.f / 1.5f; M(S, 2, 2) = 1.f;
    mat3_matmul_eig(S, V, SV);
    mat3_transpose(SV, SVt);
    mat3_matmul_eig(SVt, SV, out);
    return 1;
}
/* invSigma^T * aniso_inv * invSigma of SE::getInvCovMtx / sparseConvNoiseOneOverSecondDerivative (GPF.cpp:716-726, 776-786) */
static void se_sandwich(const oracle_medium *m, int isIsotropic, int divideGlobal, float globalScale, const float *aniso_inv, float *out)
{
    float iS[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, iSt[9], T[9];
    if (!isIsotropic) {
        if (!m->P.use_aniso_mtx) {
            iS[0] = m->l_aniso_inv[0]; iS[4] = m->l_aniso_inv[1]; iS[8] = m->l_aniso_inv[2];
        } else {
            memcpy(iS, m->w2l, sizeof iS);
        }
        if (divideGlobal)
            for (int i = 0; i < 9; ++i) iS[i] /= globalScale;
    }
    mat3_transpose(iS, iSt);
    mat3_matmul_eig(iSt, aniso_inv, T);
    mat3_matmul_eig(T, iS, out);
}
/* SE::getInvCovMtx, GPF.cpp:774-802 (aniso_inv null without an aniso field) */
static void se_inv_cov_mtx(const oracle_medium *m, int isCov, int isIsotropic, float globalScale, float localScale, const float *aniso_inv, float *out)
{
    float A[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (aniso_inv) {
        se_sandwich(m, isIsotropic, 1, globalScale, aniso_inv, A);
    } else if (!isIsotropic) {
        if (!m->P.use_aniso_mtx) {
            A[0] = m->l_aniso_inv[0] * m->l_aniso_inv[0];
            A[4] = m->l_aniso_inv[1] * m->l_aniso_inv[1];
            A[8] = m->l_aniso_inv[2] * m->l_aniso_inv[2];
        } else {
            memcpy(A, m->cov_mtx_inv, sizeof A);
        }
        float g2 = globalScale * globalScale;
        for (int i = 0; i < 9; ++i) A[i] /= g2;
    }
    if (isCov)
        for (int i = 0; i < 9; ++i) A[i] *= 0.5f;
    float l2 = localScale * localScale;
    for (int i = 0; i < 9; ++i) A[i] /= l2;
    for (int i = 0; i < 9; ++i) A[i] *= 0.5f;
    memcpy(out, A, sizeof A);
}
/* SE::splattingKernel3DVal / 1stGrad / 2ndGrad, GPF.cpp:804-833 */
static float se_splat3d_val(v3f ab, const float *A) { return expf(-eig_dist2_ab(ab, A)); }
static v3f se_splat3d_grad(v3f ab, const float *A)
{
    float f = se_splat3d_val(ab, A);
    return v3(-2.f * eig_dot_col(ab, A, 0) * f, -2.f * eig_dot_col(ab, A, 1) * f, -2.f * eig_dot_col(ab, A, 2) * f);
}
static void se_splat3d_hess(v3f ab, const float *A, float *H)
{
    float f = se_splat3d_val(ab, A);
    float dx = -2.f * eig_dot_col(ab, A, 0), dy = -2.f * eig_dot_col(ab, A, 1), dz = -2.f * eig_dot_col(ab, A, 2);
    float G[9];
    M(G, 0, 0) = dx * dx - 2.f * M(A, 0, 0);
    M(G, 1, 1) = dy * dy - 2.f * M(A, 1, 1);
    M(G, 2, 2) = dz * dz - 2.f * M(A, 2, 2);
    M(G, 0, 1) = M(G, 1, 0) = dx * dy - 2.f * M(A, 0, 1);
    M(G, 0, 2) = M(G, 2, 0) = dx * dz - 2.f * M(A, 0, 2);
    M(G, 1, 2) = M(G, 2, 1) = dy * dz - 2.f * M(A, 1, 2);
    for (int i = 0; i < 9; ++i) H[i] = G[i] * f;
}
/* CovarianceFunction::splattingKernel3D, GPF.cpp:552-562 */
static v4f cov_splat3d(const oracle_medium *m, v3f pa, v3f pb, int isCov, int isIso, float globalScale, v3f p_world)
{
    if (m->P.kernel_type != GPIS_KERNEL_SQUARED_EXPONENTIAL)
        return other_splat3d(m, v3_sub(pa, pb));      /* StationaryCovariance forwards a - b, GPF.hpp:1607-1612 */
    float localScale = cov_ns_scale(m, p_world);
    float A[9], an[9], an_inv[9];
    int has = cov_ns_aniso3d(m, p_world, an);
    if (has) eig_inverse3(an, an_inv);
    se_inv_cov_mtx(m, isCov, isIso, globalScale, localScale, has ? an_inv : NULL, A);
    v3f ab = v3_sub(pa, pb);
    float val = se_splat3d_val(ab, A);
    v3f g = se_splat3d_grad(ab, A);
    return v4(val, g.x, g.y, g.z);
}
/* CovarianceFunction::splattingKernel3DGrad, GPF.cpp:564-577 */
static v4f cov_splat3d_grad(const oracle_medium *m, v3f pa, v3f pb, v3f coeff, int isCov, int isIso, float globalScale, v3f p_world)
{
    if (m->P.kernel_type != GPIS_KERNEL_SQUARED_EXPONENTIAL) {
        /* these kernels do not override splattingKernel3D2ndGrad: the base returns a zero matrix (GPF.hpp:1518-1521) */
        v4f k = other_splat3d(m, v3_sub(pa, pb));
        v4f vx = v4(k.gx, 0.f, 0.f, 0.f), vy = v4(k.gy, 0.f, 0.f, 0.f), vz = v4(k.gz, 0.f, 0.f, 0.f);
        return v4_add(v4_add(v4_scale(vx, coeff.x), v4_scale(vy, coeff.y)), v4_scale(vz, coeff.z));
    }
    float localScale = cov_ns_scale(m, p_world);
    float A[9], H[9], an[9], an_inv[9];
    int has = cov_ns_aniso3d(m, p_world, an);
    if (has) eig_inverse3(an, an_inv);
    se_inv_cov_mtx(m, isCov, isIso, globalScale, localScale, has ? an_inv : NULL, A);
    v3f ab = v3_sub(pa, pb);
    v3f g = se_splat3d_grad(ab, A);
    se_splat3d_hess(ab, A, H);
    v4f vx = v4(g.x, M(H, 0, 0), M(H, 0, 1), M(H, 0, 2));
    v4f vy = v4(g.y, M(H, 1, 0), M(H, 1, 1), M(H, 1, 2));
    v4f vz = v4(g.z, M(H, 2, 0), M(H, 2, 1), M(H, 2, 2));
    return v4_add(v4_add(v4_scale(vx, coeff.x), v4_scale(vy, coeff.y)), v4_scale(vz, coeff.z));
}

/* SE 1D family, GPF.cpp:835-865 */
static float se_splat1d_val(float ab, float localScale)
{
    float denominator = (float)(2.0 * (localScale * localScale));
    return expf(-(ab * ab / denominator));
}
static float se_cov1d_val(float ab, float localScale) { return se_splat1d_val(ab, (float)(localScale * sqrt(2.0))); }
static float se_splat1d_grad(float ab, float localScale)
{
    float f = se_splat1d_val(ab, localScale);
    float denominator = (float)(2.0 * (localScale * localScale));
    return (float)(-2.0 * ab / denominator * f);
}
static float se_cov1d_grad(float ab, float localScale) { return se_splat1d_grad(ab, (float)(localScale * sqrt(2.0))); }
static float se_cov2d_2nd(float ab, float localScale)
{
    float f = se_cov1d_val(ab, localScale);
    float denominator = (float)(4.0 * (localScale * localScale));
    float grad1st = (float)(-2.0 * ab / denominator);
    return (float)((grad1st * grad1st - 2.0 / denominator) * f);
}
static float se_cov2d_2nd_normal(float ab, float localScale)
{
    float f = se_cov1d_val(ab, localScale);
    float denominator = (float)(4.0 * (localScale * localScale));
    return (float)(-2.0 / denominator * f);
}
/* CovarianceFunction::splattingKernel1D, GPF.cpp:579-585; the stationary overrides drop
 * localScale (GPF.hpp:1618-1628), the wrapper forwards it (GPF.cpp:1298-1312). */
static void cov_splat1d(const oracle_medium *m, float pQuery, float pCenter, v3f pCenterWorld, float *val, float *grad)
{
    float localScale = cov_ns_scale(m, pCenterWorld);
    localScale *= sqrtf(1.0f); /* getNonstationaryAniso1D = 1 */
    float ls = m->P.nonstationary ? localScale : 1.0f;
    *val = se_splat1d_val(pQuery - pCenter, ls);
    *grad = se_splat1d_grad(pQuery - pCenter, ls);
}
/* covarianceKernel1D*, GPF.cpp:587-602, GPF.hpp:1622-1640, GPF.cpp:1302-1331 */
static float cov_cov1d_scale(const oracle_medium *m, v3f pq, v3f pc) { return m->P.nonstationary ? cov_splat_cov_1d(m, pq, pc) : 1.0f; }

/* SE::sparseConvNoiseOneOverSecondDerivative, GPF.cpp:713-739, reached through NonstationaryCovariance (GPF.cpp:1239-1243),
 * which hands over the aniso matrix at p_world (inverted inside) */
static void cov_one_over_second_derivative(const oracle_medium *m, v3f p_world, int isIsotropic, float *out)
{
    float A[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float an[9], an_inv[9];
    if (cov_ns_aniso3d(m, p_world, an)) {
        eig_inverse3(an, an_inv);
        se_sandwich(m, isIsotropic, 0, 1.f, an_inv, A);
    } else if (!isIsotropic) {
        if (!m->P.use_aniso_mtx) {
            A[0] = m->l_aniso_inv[0] * m->l_aniso_inv[0];
            A[4] = m->l_aniso_inv[1] * m->l_aniso_inv[1];
            A[8] = m->l_aniso_inv[2] * m->l_aniso_inv[2];
        } else {
            memcpy(A, m->cov_mtx_inv, sizeof A);
        }
    }
    float S[9];
    for (int i = 0; i < 9; ++i) A[i] *= 0.25f;
    for (int i = 0; i < 9; ++i) S[i] = -2.f * A[i];
    eig_inverse3(S, out);
}

/* ======================================================================================
 * Noise evaluator (SCN.cpp)
 * ==================================================================================== */

/* computeSeed, SCN.cpp:40-49 */
static uint32_t compute_seed(const oracle_medium *m, ray_info info)
{
    uint32_t seed = m->P.seed;
    if (!m->P.single_realization) {
        uint32_t w = info.seg;
        if (m->P.correlation_context == GPIS_CTX_GLOBAL)
            w = 0;
        seed += xxhash32_4(info.px, info.py, info.spp, w) + xxhash32_1(info.scene_seed);
    }
    return seed;
}

/* kernelScaleLevelRatio, SCN.cpp:60-70 → (lateral_scale_low, lateral_scale_high, ratio_low, ratio_high) */
static void kernel_scale_level_ratio(const oracle_medium *m, v3f p, float out[4])
{
    float lateral_scale = cov_lateral_scale(m, p);
    float level_low = floorf(logf(lateral_scale) / logf(m->base));
    float level_high = level_low + 1;
    float lateral_scale_low = powf(m->base, level_low);
    float lateral_scale_high = powf(m->base, level_high);
    float frac = (lateral_scale - lateral_scale_low) / (lateral_scale_high - lateral_scale_low);
    float ratio_low = (float)((1.0 - frac) / sqrt(1.0 - 2.0 * frac + 2.0 * frac * frac));
    float ratio_high = (float)(frac / sqrt(1.0 - 2.0 * frac + 2.0 * frac * frac));
    out[0] = lateral_scale_low; out[1] = lateral_scale_high; out[2] = ratio_low; out[3] = ratio_high;
}
/* `int additional_seed = floor(log(kernelSpatialScale) / log(_base))`, SCN.cpp:252,272,300,328 */
static int additional_seed_of(const oracle_medium *m, float kernelSpatialScale)
{
    return (int)floorf(logf(kernelSpatialScale) / logf(m->base));
}

/* cell3D, SCN.cpp:375-395.  `p` is the query relative to the cell origin in cell units. */
static v4f cell3d(const realization *r, v3f p_world, uint32_t ci, uint32_t cj, uint32_t ck, v3f p, uint32_t seed,
                  float kernelRadius, float kernelSpatialScale)
{
    const oracle_medium *m = r->m;
    pcg32 s;
    pcg_set_state(&s, (uint64_t)(uint32_t)(xxhash32_4(ck, cj, ci, seed) + 1u));
    v4f sum = v4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t k = 0u; k < m->n_impulses; ++k) {
        /* next3D(): g++ evaluates Vec3f(next1D(), next1D(), next1D()) right to left */
        float pz = pcg_next_1d(&s);
        float py = pcg_next_1d(&s);
        float px = pcg_next_1d(&s);
        v3f p_i = v3(px, py, pz);
        float w_i = bernoulli_pm1(pcg_next_1d(&s));
        v3f to_point = v3_sub(p, p_i);
        if (v3_length_sq(to_point) < 1.0f) {
            v4f k4 = cov_splat3d(m, v3_scale(p, kernelRadius), v3_scale(p_i, kernelRadius), 0, m->P.isotropic_3d_sampling,
                                 kernelSpatialScale, p_world);
            sum = v4_add(sum, v4_scale_l(w_i, k4));
        }
    }
    return sum;
}
/* noise3D, SCN.cpp:362-373 */
static v4f noise3d(const realization *r, v3f p_world, v3f p, uint32_t seed, float kernelRadius, float kernelSpatialScale)
{
    v3f p_grid = v3_div(p, kernelRadius);
    v3f fl = v3(floorf(p_grid.x), floorf(p_grid.y), floorf(p_grid.z));
    v3f frac = v3_sub(p_grid, fl);
    int i = (int)fl.x, j = (int)fl.y, k = (int)fl.z;
    v4f sum = v4(0.f, 0.f, 0.f, 0.f);
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz)
                sum = v4_add(sum, cell3d(r, p_world, (uint32_t)(i + dx), (uint32_t)(j + dy), (uint32_t)(k + dz),
                                         v3_sub(frac, v3((float)dx, (float)dy, (float)dz)), seed, kernelRadius, kernelSpatialScale));
    return sum;
}
/* cell1D / noise1D, SCN.cpp:397-428 → (value, d/dz) */
static void cell1d(const realization *r, v3f p_world, uint32_t i, float t, uint32_t seed, float kernelRadius, float *sv, float *sg)
{
    const oracle_medium *m = r->m;
    pcg32 s;
    pcg_set_state(&s, (uint64_t)(uint32_t)(xxhash32_2(i, seed) + 1u));
    float sum_v = 0.f, sum_g = 0.f;
    for (uint32_t k = 0u; k < m->n_impulses; ++k) {
        float t_i = pcg_next_1d(&s);
        float w_i = bernoulli_pm1(pcg_next_1d(&s));
        float to_point = t - t_i;
        if (to_point * to_point < 1.0f) {
            float kv, kg;
            cov_splat1d(m, kernelRadius * t, kernelRadius * t_i, p_world, &kv, &kg);
            sum_v += w_i * kv;
            sum_g += w_i * kg;
        }
    }
    *sv = sum_v; *sg = sum_g;
}
static void noise1d(const realization *r, v3f p_world, float t, uint32_t seed, float kernelRadius, float *ov, float *og)
{
    float t_grid = t / kernelRadius;
    float fl = floorf(t_grid);
    float frac = t_grid - fl;
    int i = (int)fl;
    float sv = 0.f, sg = 0.f;
    for (int dx = -1; dx <= 1; ++dx) {
        float cv, cg;
        cell1d(r, p_world, (uint32_t)(i + dx), frac - dx, seed, kernelRadius, &cv, &cg);
        sv += cv; sg += cg;
    }
    *ov = sv; *og = sg;
}

/* evaluateNoise3DNormalized (world space), SCN.cpp:251-260 */
static v4f eval_noise3d_world(const realization *r, v3f p, uint32_t seed, float dens, float kernelRadius, float kss, int conditioning)
{
    const oracle_medium *m = r->m;
    int additional_seed = additional_seed_of(m, kss);
    v4f noise = noise3d(r, p, p, seed + (uint32_t)additional_seed, kernelRadius, kss);
    float nf = sqrtf(cov_variance3d(m, p, dens, kernelRadius, 0, kss));
    noise = v4_div(noise, nf);
    if (m->activate_conditioning && conditioning) {
        v3f o = v3(r->c3.ray_origin[0], r->c3.ray_origin[1], r->c3.ray_origin[2]);
        v3f gs = v3(r->c3.gradient_scale[0], r->c3.gradient_scale[1], r->c3.gradient_scale[2]);
        v4f a = v4_scale(cov_splat3d(m, p, o, 1, 0, kss, p), r->c3.value_scale);
        v4f b = cov_splat3d_grad(m, p, o, gs, 1, 0, kss, p);
        noise = v4_add(noise, v4_add(a, b));
    }
    return noise;
}
/* evaluateNoise3DIsotropicRayNormalized, SCN.cpp:291-320 (selected by SCN.cpp:17, 262-267) */
static v4f eval_noise3d_iso_ray(const realization *r, v3f p, v3f rayDir, uint32_t seed, float dens, float kernelRadius, float kss, int conditioning)
{
    const oracle_medium *m = r->m;
    v3f ray_dir_iso = v3_normalized(cov_pos_world_to_local(m, rayDir, 1.0f));
    frame coord = frame_from_normal(ray_dir_iso);
    v3f p_iso = cov_pos_world_to_local(m, p, kss);
    v3f p_iso_ray = frame_to_local(&coord, p_iso);
    int additional_seed = additional_seed_of(m, kss);
    v4f n = noise3d(r, p, p_iso_ray, seed + (uint32_t)additional_seed, kernelRadius, 1.0f);
    v3f grad_iso = frame_to_global(&coord, v3(n.gx, n.gy, n.gz));
    v3f grad_world = cov_grad_local_to_world(m, grad_iso, kss);
    v4f noise_world = v4(n.v, grad_world.x, grad_world.y, grad_world.z);
    float nf = sqrtf(cov_variance3d(m, p, dens, kernelRadius, 1, 1.0f));
    noise_world = v4_div(noise_world, nf);
    if (m->activate_conditioning && conditioning) {
        v3f o = v3(r->c3.ray_origin[0], r->c3.ray_origin[1], r->c3.ray_origin[2]);
        v3f gs = v3(r->c3.gradient_scale[0], r->c3.gradient_scale[1], r->c3.gradient_scale[2]);
        v3f origin_iso = cov_pos_world_to_local(m, o, kss);
        v3f origin_iso_ray = frame_to_local(&coord, origin_iso);
        v4f d = v4_add(v4_scale(cov_splat3d(m, p_iso_ray, origin_iso_ray, 1, 1, 1.0f, p), r->c3.value_scale),
                       cov_splat3d_grad(m, p_iso_ray, origin_iso_ray, gs, 1, 1, 1.0f, p));
        v3f gd_iso = frame_to_global(&coord, v3(d.gx, d.gy, d.gz));
        v3f gd_world = cov_grad_local_to_world(m, gd_iso, kss);
        noise_world = v4_add(noise_world, v4(d.v, gd_world.x, gd_world.y, gd_world.z));
    }
    return noise_world;
}
/* evaluateNoise3D, SCN.cpp:101-124 */
static v4f eval_noise3d(realization *r, v3f p, v3f rayDir, uint32_t seed, int conditioning)
{
    const oracle_medium *m = r->m;
    float dens = m->P.impulse_density;
    r->cnt->n_eval++;
    if (!m->P.isotropic_3d_sampling) {
        if (!m->multi_res) {
            float kss = cov_world_sampling_scale(m);
            return eval_noise3d_world(r, p, seed, dens, cov_kernel_radius(m, 0, 1.0f), kss, conditioning);
        } else {
            float info[4];
            kernel_scale_level_ratio(m, p, info);
            v4f lo = eval_noise3d_world(r, p, seed, dens, cov_kernel_radius(m, 0, info[0]), info[0], conditioning);
            v4f hi = eval_noise3d_world(r, p, seed, dens, cov_kernel_radius(m, 0, info[1]), info[1], conditioning);
            return v4_add(v4_scale_l(info[2], lo), v4_scale_l(info[3], hi));
        }
    } else {
        if (!m->multi_res)
            return eval_noise3d_iso_ray(r, p, rayDir, seed, dens, cov_kernel_radius(m, 1, 1.0f), 1.0f, conditioning);
        float info[4];
        kernel_scale_level_ratio(m, p, info);
        v4f lo = eval_noise3d_iso_ray(r, p, rayDir, seed, dens, cov_kernel_radius(m, 1, 1.0f), info[0], conditioning);
        v4f hi = eval_noise3d_iso_ray(r, p, rayDir, seed, dens, cov_kernel_radius(m, 1, 1.0f), info[1], conditioning);
        return v4_add(v4_scale_l(info[2], lo), v4_scale_l(info[3], hi));
    }
}

/* evaluateNoise1DNormalized, SCN.cpp:322-360 → (value, gx_cond, gy_cond, gz) */
static v4f eval_noise1d_normalized(const realization *r, v3f p, v3f rayDir, uint32_t seed, float dens, float kernelRadius,
                                   float kss, int conditioning, int multiResLowLevel)
{
    const oracle_medium *m = r->m;
    v3f ray_dir_iso = v3_normalized(cov_pos_world_to_local(m, rayDir, 1.0f));
    frame coord = frame_from_normal(ray_dir_iso);
    v3f p_iso = cov_pos_world_to_local(m, p, kss);
    v3f p_iso_ray = frame_to_local(&coord, p_iso);
    int additional_seed = additional_seed_of(m, kss);
    float nv, ng;
    noise1d(r, p, p_iso_ray.z, seed + (uint32_t)additional_seed, kernelRadius, &nv, &ng);
    float nf = sqrtf(cov_variance1d(m, p, dens, kernelRadius));
    nv /= nf; ng /= nf;
    float gsx = 0.f, gsy = 0.f;
    if (m->activate_conditioning && conditioning) {
        v3f o = v3(r->c1.ray_origin[0], r->c1.ray_origin[1], r->c1.ray_origin[2]);
        float origin_scale_factor = 1.0f;
        if (m->multi_res) {
            float oi[4];
            kernel_scale_level_ratio(m, o, oi);
            origin_scale_factor = multiResLowLevel ? oi[2] : oi[3];
        }
        v3f origin_iso = cov_pos_world_to_local(m, o, kss);
        v3f origin_iso_ray = frame_to_local(&coord, origin_iso);
        float d = p_iso_ray.z - origin_iso_ray.z;
        float cs = cov_cov1d_scale(m, p, o);
        /* covarianceKernel1D = (val, 1stGrad); covarianceKernel1DGrad = (1stGrad, 2ndGrad) */
        float kv = se_cov1d_val(d, cs), kg = se_cov1d_grad(d, cs), k2 = se_cov2d_2nd(d, cs);
        float vs_v = r->c1.value_scale * kv, vs_g = r->c1.value_scale * kg;
        float kz = kss * r->c1.gradient_scale[2];
        float gz_v = kz * kg, gz_g = kz * k2;
        nv += origin_scale_factor * (vs_v + gz_v);
        ng += origin_scale_factor * (vs_g + gz_g);
        if (m->P.correlation_xy) {
            float kn = se_cov2d_2nd_normal(d, cs);
            gsx = origin_scale_factor * kss * r->c1.gradient_scale[0] * kn;
            gsy = origin_scale_factor * kss * r->c1.gradient_scale[1] * kn;
        }
    }
    return v4(nv, gsx, gsy, ng);
}
/* evaluateValueNoise1D, SCN.cpp:126-135 */
static float eval_value_noise1d(realization *r, v3f p, v3f rayDir, uint32_t seed, int conditioning)
{
    const oracle_medium *m = r->m;
    float dens = m->P.impulse_density;
    r->cnt->n_eval++;
    if (!m->multi_res)
        return eval_noise1d_normalized(r, p, rayDir, seed, dens, cov_kernel_radius(m, 1, 1.0f), 1.0f, conditioning, 0).v;
    float info[4];
    kernel_scale_level_ratio(m, p, info);
    float lo = eval_noise1d_normalized(r, p, rayDir, seed, dens, cov_kernel_radius(m, 1, 1.0f), info[0], conditioning, 1).v;
    float hi = eval_noise1d_normalized(r, p, rayDir, seed, dens, cov_kernel_radius(m, 1, 1.0f), info[1], conditioning, 0).v;
    return info[2] * lo + info[3] * hi;
}
/* the reseed shared by evaluateGradientNoise1D and neePDF, SCN.cpp:151,193,692,720 */
static uint64_t xy_reseed(ray_info info)
{
    return (uint64_t)(uint32_t)(xxhash32_4(info.px, info.py, info.spp, info.seg) + xxhash32_1(info.scene_seed) + 1u);
}
/* evaluateGradientNoise1D, SCN.cpp:137-239 */
static v3f eval_gradient_noise1d(realization *r, v3f p, float tSegment, v3f rayDir, ray_info info, int conditioning)
{
    const oracle_medium *m = r->m;
    float dens = m->P.impulse_density;
    uint32_t seed = compute_seed(m, info);
    r->cnt->n_eval++;
    v3f ray_dir_iso = v3_normalized(cov_pos_world_to_local(m, rayDir, 1.0f));
    frame coord = frame_from_normal(ray_dir_iso);
    float scaleX = cov_splat_cov_1d(m, p, p);
    float scaleY = cov_splat_cov_1d(m, p, p);
    double xs0 = 1.0 / scaleX, xs1 = 1.0 / scaleY;
    pcg32 s;
    const double inv_div = (double)sqrtf(2.f);   /* `/ sqrt(2.f)` on a VectorXd */
    int plus_xy = (m->P.correlation_context == GPIS_CTX_RENEWAL_PLUS && m->P.correlation_xy);
    if (!m->multi_res) {
        v4f base = eval_noise1d_normalized(r, p, rayDir, seed, dens, cov_kernel_radius(m, 1, 1.0f), 1.0f, conditioning, 0);
        v3f gbase = v3(base.gx, base.gy, base.gz);
        pcg_set_state(&s, xy_reseed(info));
        double a, b;
        rand_normal_2(&s, &a, &b);
        double x1 = a / inv_div, y1 = b / inv_div;
        v3f g;
        if (tSegment == 0) {
            x1 = x1 * xs0; y1 = y1 * xs1;
            g = v3_add(v3((float)x1, (float)y1, 0.f), gbase);
        } else if (!plus_xy) {
            double c, d;
            rand_normal_2(&s, &c, &d);
            double x2 = c / inv_div, y2 = d / inv_div;
            x2 = x2 * xs0; y2 = y2 * xs1;
            g = v3_add(v3((float)x2, (float)y2, 0.f), gbase);
        } else {
            float factor = (float)(exp(-tSegment * tSegment / 4.0) * (0.5 - tSegment * tSegment * 0.25));
            double mux = factor * 2.0 * x1, muy = factor * 2.0 * y1;
            float cov = (float)(0.5 - factor * factor * 2);
            double c, d;
            rand_normal_2(&s, &c, &d);
            double sc = sqrtf(cov);
            double x2 = c * sc + mux, y2 = d * sc + muy;
            g = v3_add(v3((float)x2, (float)y2, 0.f), gbase);
        }
        v3f grad_iso = frame_to_global(&coord, g);
        return cov_grad_local_to_world(m, grad_iso, 1.0f);
    }
    /* multi-resolution branch, SCN.cpp:188-238 */
    float si[4];
    kernel_scale_level_ratio(m, p, si);
    v4f bl = eval_noise1d_normalized(r, p, rayDir, seed, dens, cov_kernel_radius(m, 1, 1.0f), si[0], conditioning, 1);
    v4f bh = eval_noise1d_normalized(r, p, rayDir, seed, dens, cov_kernel_radius(m, 1, 1.0f), si[1], conditioning, 0);
    v3f gl_base = v3(bl.gx, bl.gy, bl.gz), gh_base = v3(bh.gx, bh.gy, bh.gz);
    pcg_set_state(&s, xy_reseed(info));
    double al, bl2, ah, bh2;
    rand_normal_2(&s, &al, &bl2);
    rand_normal_2(&s, &ah, &bh2);
    double x1l = al / inv_div, y1l = bl2 / inv_div, x1h = ah / inv_div, y1h = bh2 / inv_div;
    v3f gl, gh;
    if (tSegment == 0) {
        gl = v3_add(v3((float)(x1l * xs0), (float)(y1l * xs1), 0.f), gl_base);
        gh = v3_add(v3((float)(x1h * xs0), (float)(y1h * xs1), 0.f), gh_base);
    } else if (!plus_xy) {
        double cl, dl, ch, dh;
        rand_normal_2(&s, &cl, &dl);
        rand_normal_2(&s, &ch, &dh);
        gl = v3_add(v3((float)(cl / inv_div * xs0), (float)(dl / inv_div * xs1), 0.f), gl_base);
        gh = v3_add(v3((float)(ch / inv_div * xs0), (float)(dh / inv_div * xs1), 0.f), gh_base);
    } else {
        float factor = (float)(exp(-tSegment * tSegment / 4.0) * (0.5 - tSegment * tSegment * 0.25));
        float cov = (float)(0.5 - factor * factor * 2);
        double sc = sqrtf(cov);
        double cl, dl, ch, dh;
        rand_normal_2(&s, &cl, &dl);
        rand_normal_2(&s, &ch, &dh);
        gl = v3_add(v3((float)(cl * sc + factor * 2.0 * x1l), (float)(dl * sc + factor * 2.0 * y1l), 0.f), gl_base);
        gh = v3_add(v3((float)(ch * sc + factor * 2.0 * x1h), (float)(dh * sc + factor * 2.0 * y1h), 0.f), gh_base);
    }
    frame cl_ = frame_from_normal(v3_normalized(cov_pos_world_to_local(m, rayDir, si[0])));
    frame ch_ = frame_from_normal(v3_normalized(cov_pos_world_to_local(m, rayDir, si[1])));
    v3f gwl = cov_grad_local_to_world(m, frame_to_global(&cl_, gl), si[0]);
    v3f gwh = cov_grad_local_to_world(m, frame_to_global(&ch_, gh), si[1]);
    return v3_add(v3_scale(gwl, si[2]), v3_scale(gwh, si[3]));   /* scaleInfo.z()*low + scaleInfo.w()*high */
}

/* evaluateValue, SCN.cpp:73-89 */
static float evaluate_value(realization *r, v3f p, float t, v3f rayDir, ray_info info, int *GPId)
{
    const oracle_medium *m = r->m;
    (void)t;
    uint32_t seed = compute_seed(m, info);
    float amplitude = cov_amplitude(m, p);
    float noise_val = m->P.sampling_1d ? eval_value_noise1d(r, p, rayDir, seed, 1) : eval_noise3d(r, p, rayDir, seed, 1).v;
    double mean, id;
    mean_weight_space(m, v3d_of(p), &mean, &id);
    *GPId = (int)id;
    if (m->P.surf_vol_phase_separate) {
        /* getUnscaledVariance = 1 (GPF.hpp:1185) except in the grid flavour (GPF.cpp:1386-1391) */
        float uv = m->P.grid_nonstationary ? grid_unscaled_variance(m, v3d_of(p)) : 1.f;
        if (uv < m->P.surf_vol_phase_amp_thresh) *GPId = 0; else *GPId = 1;
    }
    return (float)(amplitude * noise_val + mean);
}
/* evaluateGradient, SCN.cpp:92-99 */
static v3f evaluate_gradient(realization *r, v3f p, float t, v3f rayDir, ray_info info)
{
    const oracle_medium *m = r->m;
    uint32_t seed = compute_seed(m, info);
    float amplitude = cov_amplitude(m, p);
    v3f noise_grad;
    if (m->P.sampling_1d) {
        noise_grad = eval_gradient_noise1d(r, p, t, rayDir, info, 1);
    } else {
        v4f n = eval_noise3d(r, p, rayDir, seed, 1);
        noise_grad = v3(n.gx, n.gy, n.gz);
    }
    double mean, id;
    mean_weight_space(m, v3d_of(p), &mean, &id);
    v3f mg = v3f_of(mean_grad(m, id == 0 ? 0 : 1, v3d_of(p)));
    return v3_add(v3_scale(noise_grad, amplitude), mg);   /* amplitude * noise_grad + mean_grad */
}

/* conditioning3D, SCN.cpp:441-517 (sanity re-evaluations only print; they are still evaluated
 * by the reference, so they are counted) */
static void conditioning3d(realization *r, v3f p, v3f rayDir, float targetVal, v3f targetGrad, ray_info info)
{
    const oracle_medium *m = r->m;
    r->c3.value_scale = 0.f;
    r->c3.gradient_scale[0] = r->c3.gradient_scale[1] = r->c3.gradient_scale[2] = 0.f;
    r->c3.ray_origin[0] = p.x; r->c3.ray_origin[1] = p.y; r->c3.ray_origin[2] = p.z;
    float ksi[4] = {0, 0, 0, 0};
    if (m->multi_res)
        kernel_scale_level_ratio(m, p, ksi);
    uint32_t seed = compute_seed(m, info);
    float amplitude = 0.f;
    double mean_d = 0., id = 0.;
    int ctx = m->P.correlation_context;
    if (ctx == GPIS_CTX_RENEWAL || ctx == GPIS_CTX_RENEWAL_PLUS) {
        amplitude = cov_amplitude(m, p);
        if (amplitude == 0)
            return;
        mean_weight_space(m, v3d_of(p), &mean_d, &id);
        float mean = (float)mean_d;
        float current_value = eval_noise3d(r, p, rayDir, seed, 0).v;
        r->c3.value_scale = (targetVal - mean) / amplitude - current_value;
        if (m->multi_res)
            r->c3.value_scale /= ksi[2] + ksi[3];
        (void)eval_noise3d(r, p, rayDir, seed, 1);   /* sanity check, SCN.cpp:470-474 */
    }
    if (ctx == GPIS_CTX_RENEWAL_PLUS) {
        v3f mg = v3f_of(mean_grad(m, id == 0 ? 0 : 1, v3d_of(p)));
        v4f cur = eval_noise3d(r, p, rayDir, seed, 0);
        v3f current_grad_world = v3(cur.gx, cur.gy, cur.gz);
        v3f delta = v3_sub(v3_div(v3_sub(targetGrad, mg), amplitude), current_grad_world);
        float S[9];
        if (m->P.isotropic_3d_sampling) {
            v3f gs_iso = cov_grad_world_to_local(m, delta, 1.0f);
            v3f ray_dir_iso = v3_normalized(cov_pos_world_to_local(m, rayDir, 1.0f));
            frame coord = frame_from_normal(ray_dir_iso);
            gs_iso = frame_to_local(&coord, gs_iso);
            cov_one_over_second_derivative(m, p, 1, S);
            v3f g = eig_matvec(S, gs_iso);
            if (m->multi_res) {
                float d = ksi[2] / ksi[0] + ksi[3] / ksi[1];
                g = v3_div(g, d);
            } else {
                float q = cov_ns_scale(m, p);
                g = v3_scale(g, q * q);
            }
            r->c3.gradient_scale[0] = g.x; r->c3.gradient_scale[1] = g.y; r->c3.gradient_scale[2] = g.z;
        } else {
            cov_one_over_second_derivative(m, p, 0, S);
            v3f g = eig_matvec(S, delta);
            if (m->multi_res) {
                float d = ksi[2] / (ksi[0] * ksi[0]) + ksi[3] / (ksi[1] * ksi[1]);
                g = v3_div(g, d);
            } else {
                float q = cov_lateral_scale(m, p);
                g = v3_scale(g, q * q);
            }
            r->c3.gradient_scale[0] = g.x; r->c3.gradient_scale[1] = g.y; r->c3.gradient_scale[2] = g.z;
        }
        (void)eval_noise3d(r, p, rayDir, seed, 1);   /* sanity check, SCN.cpp:512-515 */
    }
}
/* conditioning1D, SCN.cpp:519-595 */
static void conditioning1d(realization *r, v3f p, v3f rayDir, float targetVal, v3f targetGrad, ray_info info)
{
    const oracle_medium *m = r->m;
    r->c1.value_scale = 0.f;
    r->c1.gradient_scale[0] = r->c1.gradient_scale[1] = r->c1.gradient_scale[2] = 0.f;
    r->c1.ray_origin[0] = p.x; r->c1.ray_origin[1] = p.y; r->c1.ray_origin[2] = p.z;
    float ksi[4] = {0, 0, 0, 0};
    if (m->multi_res)
        kernel_scale_level_ratio(m, p, ksi);
    uint32_t seed = compute_seed(m, info);
    float amplitude = 0.f;
    double mean_d = 0., id = 0.;
    int ctx = m->P.correlation_context;
    if (ctx == GPIS_CTX_RENEWAL || ctx == GPIS_CTX_RENEWAL_PLUS) {
        amplitude = cov_amplitude(m, p);
        if (amplitude == 0)
            return;
        mean_weight_space(m, v3d_of(p), &mean_d, &id);
        float mean = (float)mean_d;
        float current_value = eval_value_noise1d(r, p, rayDir, seed, 0);
        r->c1.value_scale = (targetVal - mean) / amplitude - current_value;
        if (m->multi_res)
            r->c1.value_scale /= ksi[2] * ksi[2] + ksi[3] * ksi[3];
        (void)eval_value_noise1d(r, p, rayDir, seed, 1);   /* sanity check */
    }
    if (ctx == GPIS_CTX_RENEWAL_PLUS) {
        v3f mg = v3f_of(mean_grad(m, id == 0 ? 0 : 1, v3d_of(p)));
        v3f current_grad_world = eval_gradient_noise1d(r, p, 0.f, rayDir, info, 0);
        v3f delta = v3_sub(v3_div(v3_sub(targetGrad, mg), amplitude), current_grad_world);
        v3f ray_dir_iso = v3_normalized(cov_pos_world_to_local(m, rayDir, 1.0f));
        frame coord = frame_from_normal(ray_dir_iso);
        v3f gs_iso = cov_grad_world_to_local(m, delta, 1.0f);
        gs_iso = frame_to_local(&coord, gs_iso);
        float scaleZ = cov_splat_cov_1d(m, p, p), scaleX = cov_splat_cov_1d(m, p, p), scaleY = cov_splat_cov_1d(m, p, p);
        /* -2.f * gradient_scale_iso * Vec3f(sqr(scaleX), sqr(scaleY), sqr(scaleZ)) */
        v3f g = v3((-2.f * gs_iso.x) * (scaleX * scaleX), (-2.f * gs_iso.y) * (scaleY * scaleY), (-2.f * gs_iso.z) * (scaleZ * scaleZ));
        if (m->multi_res)
            g = v3_div(g, ksi[2] * ksi[2] + ksi[3] * ksi[3]);
        if (!m->P.correlation_xy) {
            g.x = 0; g.y = 0;
        }
        r->c1.gradient_scale[0] = g.x; r->c1.gradient_scale[1] = g.y; r->c1.gradient_scale[2] = g.z;
        (void)eval_gradient_noise1d(r, p, 0.f, rayDir, info, 1);   /* sanity check */
    }
}
/* conditioning, SCN.cpp:431-439 */
static void conditioning(realization *r, v3f p, v3f rayDir, float targetVal, v3f targetGrad, ray_info info)
{
    if (!r->m->activate_conditioning)
        return;
    if (r->m->P.sampling_1d)
        conditioning1d(r, p, rayDir, targetVal, targetGrad, info);
    else
        conditioning3d(r, p, rayDir, targetVal, targetGrad, info);
}
/* samplingScheme, SCN.cpp:51-58 */
static int sampling_scheme(const realization *r, v3f p)
{
    float amplitude = cov_amplitude(r->m, p);
    if (amplitude < 1e-6f)
        return GPIS_UNI;
    return r->m->scheme_1d_eff;
}

/* neeShared, SCN.cpp:601-648 */
typedef struct {
    float mtx_pt[9];
    v3f grad_constraint_3d, grad_3d_iso_ray, plane_normal_world;
    float plane_stretch_jacobian, isect_dist;
} nee_shared;
static void nee_shared_eval(realization *r, v3f rayDir, v3f normal, v3f p, ray_info info, nee_shared *o)
{
    const oracle_medium *m = r->m;
    float dens = m->P.impulse_density;
    uint32_t seed = compute_seed(m, info);
    float amplitude = cov_amplitude(m, p);
    double mean_d, id;
    mean_weight_space(m, v3d_of(p), &mean_d, &id);
    v3f mg = v3f_of(mean_grad(m, id == 0 ? 0 : 1, v3d_of(p)));
    v3f ray_dir_iso = v3_normalized(cov_pos_world_to_local(m, rayDir, 1.0f));
    frame coord = frame_from_normal(ray_dir_iso);
    float rc[9];
    M(rc, 0, 0) = coord.tangent.x; M(rc, 1, 0) = coord.tangent.y; M(rc, 2, 0) = coord.tangent.z;
    M(rc, 0, 1) = coord.bitangent.x; M(rc, 1, 1) = coord.bitangent.y; M(rc, 2, 1) = coord.bitangent.z;
    M(rc, 0, 2) = coord.normal.x; M(rc, 1, 2) = coord.normal.y; M(rc, 2, 2) = coord.normal.z;
    float l2wit[9], sa[9];
    cov_l2w_inv_transpose(m, l2wit);
    for (int i = 0; i < 9; ++i) sa[i] = amplitude * l2wit[i];   /* (amplitude * A) * B */
    mat3_matmul_eig(sa, rc, o->mtx_pt);
    float inv[9];
    eig_inverse3(o->mtx_pt, inv);
    r->cnt->n_eval++;
    if (!m->multi_res) {
        v4f n = eval_noise1d_normalized(r, p, rayDir, seed, dens, cov_kernel_radius(m, 1, 1.0f), 1.0f, 1, 0);
        o->grad_constraint_3d = v3(n.gx, n.gy, n.gz);
    } else {
        float mi[4];
        kernel_scale_level_ratio(m, p, mi);
        v4f lo = eval_noise1d_normalized(r, p, rayDir, seed, dens, cov_kernel_radius(m, 1, 1.0f), mi[0], 1, 1);
        v4f hi = eval_noise1d_normalized(r, p, rayDir, seed, dens, cov_kernel_radius(m, 1, 1.0f), mi[1], 1, 0);
        /* low * z / x + high * w / y, left to right */
        v3f a = v3_div(v3_scale(v3(lo.gx, lo.gy, lo.gz), mi[2]), mi[0]);
        v3f b = v3_div(v3_scale(v3(hi.gx, hi.gy, hi.gz), mi[3]), mi[1]);
        o->grad_constraint_3d = v3_add(a, b);
    }
    v3f plane_pt_world = v3_add(mult3(o->mtx_pt, v3(0.f, 0.f, o->grad_constraint_3d.z)), mg);
    v3f plane_x = mult3(o->mtx_pt, v3(1.f, 0.f, 0.f));
    v3f plane_y = mult3(o->mtx_pt, v3(0.f, 1.f, 0.f));
    v3f cp = v3_cross(plane_x, plane_y);
    o->plane_normal_world = v3_normalized(cp);
    o->plane_stretch_jacobian = v3_length(cp);
    o->isect_dist = v3_dot(plane_pt_world, o->plane_normal_world) / v3_dot(normal, o->plane_normal_world);
    v3f grad_3d_world = v3_scale(normal, o->isect_dist);   /* isect_dist * normal */
    o->grad_3d_iso_ray = mult3(inv, v3_sub(grad_3d_world, mg));
}
/* neeGrad, SCN.cpp:652-658 */
static v3f nee_grad(realization *r, v3f rayDir, v3f normal, v3f p, ray_info info)
{
    nee_shared s;
    nee_shared_eval(r, rayDir, normal, p, info, &s);
    return v3_scale(normal, s.isect_dist);
}
/* neePDF, SCN.cpp:661-743 */
static float nee_pdf(realization *r, v3f rayDir, v3f normal, v3f p, float tSegment, ray_info info)
{
    const oracle_medium *m = r->m;
    nee_shared s;
    nee_shared_eval(r, rayDir, normal, p, info, &s);
    if (s.isect_dist < 0.0)
        return 0.0f;
    float scaleX = cov_splat_cov_1d(m, p, p), scaleY = cov_splat_cov_1d(m, p, p);
    float sample_x, sample_y, pdf_area;
    int plus_xy = (m->P.correlation_context == GPIS_CTX_RENEWAL_PLUS && m->P.correlation_xy);
    pcg32 ps;
    const double inv_div = (double)sqrtf(2.f);
    if (!m->multi_res) {
        if (!plus_xy) {
            sample_x = (float)((s.grad_3d_iso_ray.x - s.grad_constraint_3d.x) * sqrt(2.0) * scaleX);
            sample_y = (float)((s.grad_3d_iso_ray.y - s.grad_constraint_3d.y) * sqrt(2.0) * scaleY);
            pdf_area = (float)(exp(-sample_x * sample_x / 2.0) * exp(-sample_y * sample_y / 2.0) / (2.0 * M_PI) * 2.0 * scaleX * scaleY);
        } else {
            pcg_set_state(&ps, xy_reseed(info));
            double a, b;
            rand_normal_2(&ps, &a, &b);
            double x1 = a / inv_div, y1 = b / inv_div;
            float factor = (float)(exp(-tSegment * tSegment / 4.0) * (0.5 - tSegment * tSegment * 0.25));
            double mux = factor * 2.0 * x1, muy = factor * 2.0 * y1;
            float cov = (float)(0.5 - factor * factor * 2);
            sample_x = (float)((s.grad_3d_iso_ray.x - s.grad_constraint_3d.x - mux) / sqrtf(cov));
            sample_y = (float)((s.grad_3d_iso_ray.y - s.grad_constraint_3d.y - muy) / sqrtf(cov));
            pdf_area = (float)(exp(-sample_x * sample_x / 2.0) * exp(-sample_y * sample_y / 2.0) / (2.0 * M_PI) / cov);
        }
    } else {
        float mi[4];
        kernel_scale_level_ratio(m, p, mi);
        float q0 = mi[2] / mi[0], q1 = mi[3] / mi[1];
        float nss = (float)(1.0 / sqrtf(q0 * q0 + q1 * q1));
        scaleX *= nss; scaleY *= nss;
        if (!plus_xy) {
            sample_x = (float)((s.grad_3d_iso_ray.x - s.grad_constraint_3d.x) * sqrt(2.0) * scaleX);
            sample_y = (float)((s.grad_3d_iso_ray.y - s.grad_constraint_3d.y) * sqrt(2.0) * scaleY);
            pdf_area = (float)(exp(-sample_x * sample_x / 2.0) * exp(-sample_y * sample_y / 2.0) / (2.0 * M_PI) * 2.0 * scaleX * scaleY);
        } else {
            pcg_set_state(&ps, xy_reseed(info));
            double a, b, c, d;
            rand_normal_2(&ps, &a, &b);
            rand_normal_2(&ps, &c, &d);
            double x1 = a / inv_div, y1 = b / inv_div, x2 = c / inv_div, y2 = d / inv_div;
            float factor = (float)(exp(-tSegment * tSegment / 4.0) * (0.5 - tSegment * tSegment * 0.25));
            /* factor * 2.0 * (xy_1 * z / x + xy_2 * w / y) */
            double mux = factor * 2.0 * (x1 * mi[2] / mi[0] + x2 * mi[3] / mi[1]);
            double muy = factor * 2.0 * (y1 * mi[2] / mi[0] + y2 * mi[3] / mi[1]);
            float cov = (float)(0.5 - factor * factor * 2);
            sample_x = (float)((s.grad_3d_iso_ray.x - s.grad_constraint_3d.x - mux) / sqrtf(cov) * nss);
            sample_y = (float)((s.grad_3d_iso_ray.y - s.grad_constraint_3d.y - muy) / sqrtf(cov) * nss);
            pdf_area = (float)(exp(-sample_x * sample_x / 2.0) * exp(-sample_y * sample_y / 2.0) / (2.0 * M_PI * cov) * (nss * nss));
        }
    }
    float cosTheta_light = fabsf(v3_dot(normal, s.plane_normal_world));
    float pdf_normal = pdf_area * (s.isect_dist * s.isect_dist) / cosTheta_light / s.plane_stretch_jacobian;
    v3f neg = v3(-rayDir.x, -rayDir.y, -rayDir.z);
    float pdf_omega = (float)(pdf_normal / (4.0 * v3_dot(normal, neg)));
    return pdf_omega;
}

/* ======================================================================================
 * Medium level (SCNM.cpp, GPM.cpp)
 * ==================================================================================== */

typedef struct {
    int first_scatter, bounce, last_gp_id, scheme;
    v3d last_aniso;
    float last_val;
    ray_info info;
} medium_state;

static inline v3d ray_at(v3d p, v3d rd, double t) { v3d r = {p.x + t * rd.x, p.y + t * rd.y, p.z + t * rd.z}; return r; }
static inline double lerp_d(double a, double b, double ratio) { return a * (1.0 - ratio) + b * ratio; }

/* SparseConvolutionNoiseMedium::intersectGP, SCNM.cpp:102-183.  Returns hit?; *t out. */
static int intersect_gp(realization *noise, v3f pos, v3f dir, float nearT, float farT, float u, medium_state *state, double *t_out)
{
    const oracle_medium *m = noise->m;
    v3d rd = v3d_of(dir);
    float step_size = (farT - nearT) / (float)m->P.min_step;
    if (m->P.step_size < step_size)
        step_size = m->P.step_size;
    int GPId = 0;
    v3d p = v3d_of(pos);
    double t = nearT;
    if (!state->first_scatter) {
        double targetVal = state->last_val;
        conditioning(noise, v3f_of(p), dir, (float)targetVal, v3f_of(state->last_aniso), state->info);
    }
    double f0 = evaluate_value(noise, v3f_of(ray_at(p, rd, t)), (float)t, dir, state->info, &GPId);
    int sign0 = f0 < 0 ? -1 : 1;
    double pf = f0;
    t = nearT + step_size * u;   /* float arithmetic, then widened */
    int step = 0;
    while (t < farT) {
        step++;
        double f_c = evaluate_value(noise, v3f_of(ray_at(p, rd, t)), (float)t, dir, state->info, &GPId);
        int signc = f_c < 0 ? -1 : 1;
        if (!state->first_scatter && step == 1) {
            sign0 = signc;
        } else if (signc != sign0) {
            double intp_factor = pf / (pf - f_c);
            double t_test, t_test_prev;
            int sign_test;
            t_test_prev = lerp_d(t - step_size, t, intp_factor);
            do {
                t_test = lerp_d(t - step_size, t, intp_factor);
                double f_test = evaluate_value(noise, v3f_of(ray_at(p, rd, t_test)), (float)t_test, dir, state->info, &GPId);
                sign_test = f_test < 0 ? -1 : 1;
                if (sign_test == sign0)
                    break;
                intp_factor *= 0.9;
                if (intp_factor <= 0.01) {
                    t_test_prev = t_test = 0;
                    break;
                }
                t_test_prev = t_test;
            } while (1);
            t = t_test_prev;
            state->last_gp_id = GPId;
            state->scheme = sampling_scheme(noise, v3f_of(ray_at(p, rd, t)));
            state->last_val = 0.0f;
            *t_out = t;
            return 1;
        }
        pf = f_c;
        t += step_size;
    }
    t = farT;
    state->scheme = GPIS_UNI;
    state->last_val = evaluate_value(noise, v3f_of(ray_at(p, rd, t)), (float)t, dir, state->info, &GPId);
    *t_out = t;
    return 0;
}

static inline int is_finite_f(float x) { return isfinite(x); }

static void state_from_ray(const gpis_ray_in *r, medium_state *s)
{
    s->first_scatter = r->first_scatter != 0;
    s->bounce = r->bounce;
    s->last_gp_id = r->last_gp_id;
    s->scheme = GPIS_UNI;
    s->last_aniso.x = r->last_aniso[0]; s->last_aniso.y = r->last_aniso[1]; s->last_aniso.z = r->last_aniso[2];
    s->last_val = r->last_val;
    s->info.px = r->pixel[0]; s->info.py = r->pixel[1]; s->info.spp = r->spp; s->info.seg = r->segment;
    s->info.scene_seed = r->scene_seed; s->info.t = r->info_t;
}

static int transmittance_one(const oracle_medium *m, oracle_counters *cnt, const gpis_ray_in *ray, medium_state *state);
static int transmittance_with(realization *noise_io, const gpis_ray_in *ray, medium_state *state);

/* GaussianProcessMedium::sampleDistance, GPM.cpp:221-341 */
static void sample_distance_one(const oracle_medium *m, oracle_counters *cnt, const gpis_ray_in *ray, gpis_seg_out *out, gpis_cond_coeff *coeff)
{
    memset(out, 0, sizeof *out);
    if (coeff) memset(coeff, 0, sizeof *coeff);
    cnt->n_seg++;
    uint64_t evals0 = cnt->n_eval;
    medium_state state;
    state_from_ray(ray, &state);
    realization noise;
    memset(&noise, 0, sizeof noise);
    noise.m = m; noise.cnt = cnt;
    v3f pos = v3(ray->pos[0], ray->pos[1], ray->pos[2]);
    v3f dir = v3(ray->dir[0], ray->dir[1], ray->dir[2]);
    double startT = ray->near_t;
    float farT = ray->far_t;
    if (!is_finite_f(farT))
        farT = (float)(startT + 2000);
    float maxT = farT;
    out->gp_id = state.last_gp_id;
    out->last_val = state.last_val;
    out->aniso[0] = state.last_aniso.x; out->aniso[1] = state.last_aniso.y; out->aniso[2] = state.last_aniso.z;
    if (state.bounce >= m->P.max_bounces) {
        out->ok = 0;
        return;
    }
    if (maxT == 0.f) {
        out->sample_t = maxT;
        out->weight[0] = out->weight[1] = out->weight[2] = 1.f;
        out->exited = 1;
        v3f pp = v3_add(pos, v3_scale(dir, out->sample_t));
        out->p[0] = pp.x; out->p[1] = pp.y; out->p[2] = pp.z;
        out->scheme = GPIS_UNI;
        out->ok = 1;
        return;
    }
    v3d aniso = state.last_aniso;
    if (m->absorption_only) {
        if (maxT == INFINITY) { out->ok = 0; return; }
        out->sample_t = maxT;
        medium_state st2 = state;
        int vis = transmittance_with(&noise, ray, &st2);
        state = st2; /* transmittance mutates the caller's state (GPM.cpp:254 passes &state) */
        out->weight[0] = out->weight[1] = out->weight[2] = vis ? 1.f : 0.f;
        out->exited = 1;
        out->scheme = GPIS_UNI;
        aniso = state.last_aniso;
    } else {
        double t = maxT;
        v3d ro = v3d_of(pos);
        v3d rd = v3d_of(dir);
        { double inv = 1.0 / v3d_length(rd); rd.x *= inv; rd.y *= inv; rd.z *= inv; }
        int exited;
        do {
            float nearT = (float)startT;
            exited = !intersect_gp(&noise, pos, dir, nearT, farT, ray->u_jitter, &state, &t);
            if (t < maxT) {
                v3d ip = ray_at(ro, rd, t);
                /* sampleGradient, SCNM.cpp:93-100: isect_t is narrowed to float */
                v3f g = evaluate_gradient(&noise, v3f_of(ip), (float)t, dir, state.info);
                aniso = v3d_of(g);
                state.last_aniso = aniso;
                state.first_scatter = 0;
                double avg = (aniso.x + aniso.y + aniso.z) / 3.0;   /* Vec::avg(): sum()/Size */
                if (!isfinite(avg)) {
                    out->aniso[0] = 1.; out->aniso[1] = 0.; out->aniso[2] = 0.;
                    out->t = t; out->exited = exited; out->ok = 0;
                    out->last_val = state.last_val; out->gp_id = state.last_gp_id;
                    goto done;
                }
            }
            startT = t;
        } while (t < maxT && exited);
        out->t = t;
        out->exited = exited;
        if (!exited) {
            double d = aniso.x * (double)dir.x; d += aniso.y * (double)dir.y; d += aniso.z * (double)dir.z;
            if (d > 0) {
                out->aniso[0] = aniso.x; out->aniso[1] = aniso.y; out->aniso[2] = aniso.z;
                out->last_val = state.last_val; out->gp_id = state.last_gp_id; out->ok = 0;
                goto done;
            }
            double l2 = 0.; l2 += aniso.x * aniso.x; l2 += aniso.y * aniso.y; l2 += aniso.z * aniso.z;
            if (l2 < 0.0000001f) {
                out->aniso[0] = 1.; out->aniso[1] = 0.; out->aniso[2] = 0.;
                out->last_val = state.last_val; out->gp_id = state.last_gp_id; out->ok = 0;
                goto done;
            }
            /* sample.weight = sample.continuedWeight = vec_conv<Vec3f>(_gp->color(ro + rd * t)), GPM.cpp:316; MeanFunction::color
             * is 1 without a "color" field (GPF.hpp:849-852) */
            double col[3] = {1., 1., 1.};
            if (m->P.mean_color.enabled) ramp_vec_of(&m->P.mean_color, ray_at(ro, rd, t), col);
            for (int c = 0; c < 3; ++c) out->weight[c] = out->continued_weight[c] = (float)col[c];
        } else {
            v3f g = evaluate_gradient(&noise, v3f_of(ray_at(ro, rd, t)), (float)t, dir, state.info);
            aniso = v3d_of(g);
            out->weight[0] = out->weight[1] = out->weight[2] = 1.f;
            out->continued_weight[0] = out->continued_weight[1] = out->continued_weight[2] = 1.f;
        }
        float ft = (float)t;
        out->sample_t = ft < maxT ? ft : maxT;     /* min(a,b) = a < b ? a : b */
        out->continued_t = (float)t;
        for (int c = 0; c < 3; ++c) {
            out->weight[c] *= m->sigma_s[c] / m->sigma_t[c];
            out->continued_weight[c] *= m->sigma_s[c] / m->sigma_t[c];
        }
        state.last_aniso = aniso;
        out->scheme = state.scheme;
        state.first_scatter = 0; state.scheme = GPIS_UNI; state.bounce++;   /* advance() */
    }
    {
        v3f pp = v3_add(pos, v3_scale(dir, out->sample_t));
        out->p[0] = pp.x; out->p[1] = pp.y; out->p[2] = pp.z;
    }
    out->aniso[0] = aniso.x; out->aniso[1] = aniso.y; out->aniso[2] = aniso.z;
    out->gp_id = state.last_gp_id;
    out->last_val = state.last_val;
    out->ok = 1;
done:
    if (coeff) {
        *coeff = m->P.sampling_1d ? noise.c1 : noise.c3;
        coeff->n_evals = (uint32_t)(cnt->n_eval - evals0);
    }
}

/* GaussianProcessMedium::transmittance, GPM.cpp:343-393 → 1 if the segment exits */
static int transmittance_with(realization *noise_io, const gpis_ray_in *ray, medium_state *state);
static int transmittance_one(const oracle_medium *m, oracle_counters *cnt, const gpis_ray_in *ray, medium_state *state)
{
    realization noise;
    memset(&noise, 0, sizeof noise);
    noise.m = m; noise.cnt = cnt;
    return transmittance_with(&noise, ray, state);
}
/* the same with the caller's realization: intersectGP leaves the realization it marched with in state.gpContext
 * (SCNM.cpp:165-167, 178-180), which is what MediumSample.ctxt points to after the absorption-only branch of
 * sampleDistance (GPM.cpp:254, 337) */
static int transmittance_with(realization *noise_io, const gpis_ray_in *ray, medium_state *state)
{
#define noise (*noise_io)
    v3f pos = v3(ray->pos[0], ray->pos[1], ray->pos[2]);
    v3f dir = v3(ray->dir[0], ray->dir[1], ray->dir[2]);
    double startT = ray->near_t;
    float farT = ray->far_t;
    if (!is_finite_f(farT))
        farT = (float)(startT + 2000);
    float maxT = farT;
    double t = maxT;
    v3d ro = v3d_of(pos);
    v3d rd = v3d_of(dir);
    { double inv = 1.0 / v3d_length(rd); rd.x *= inv; rd.y *= inv; rd.z *= inv; }
    int exited;
    do {
        float nearT = (float)startT;
        exited = !intersect_gp(&noise, pos, dir, nearT, farT, ray->u_jitter, state, &t);
        if (t < maxT) {
            v3d ip = ray_at(ro, rd, t);
            v3f g = evaluate_gradient(&noise, v3f_of(ip), (float)t, dir, state->info);
            state->last_aniso = v3d_of(g);
            state->first_scatter = 0;
            double avg = ((double)g.x + (double)g.y + (double)g.z) / 3.0;
            if (!isfinite(avg))
                return 0;
        }
        startT = t;
    } while (t < maxT && exited);
    return exited ? 1 : 0;
#undef noise
}

/* ======================================================================================
 * Construction ("fromJson")
 * ==================================================================================== */

void oracle_default_params(gpis_params *p)
{
    memset(p, 0, sizeof *p);
    p->abi_version = GPIS_ABI_VERSION;
    p->step_size = 0.01f; p->min_step = 8; p->seed = 0; p->impulse_density = 3.0f;   /* SCNM.cpp:22-32 */
    p->scheme_1d = GPIS_UNI;
    p->correlation_context = GPIS_CTX_RENEWAL_PLUS;                                  /* GPM.cpp:91 */
    p->max_bounces = 1024;                                                           /* Medium.cpp:27 */
    p->density = 1.f;
    p->sigma = 1.f; p->length_scale = 1.f;                                           /* GPF.hpp:1784 */
    p->aniso[0] = p->aniso[1] = p->aniso[2] = 1.f;
    p->aniso_mtx[0] = p->aniso_mtx[4] = p->aniso_mtx[8] = 1.f;
    p->local_scale = 3.0f;                                                           /* GPF.hpp:1729 */
    p->ls_min = 1.; p->ls_max = 500.; p->ls_start = 0.; p->ls_end = 1.;              /* GPF.hpp:694-695 */
    p->ls_min2 = 1.; p->ls_max2 = 500.; p->ls_start2 = 0.; p->ls_end2 = 1.;          /* GPF.hpp:697-698 */
    p->matern_v = 0.5f; p->gabor_a_inv = 1.f; p->gabor_f_inv = 1.f; p->gabor_omega[0] = 1.f;   /* GPF.hpp:1964, 2041, 2079 */
    p->fs_sample_points = 32; p->fs_step_size = 0.;                                  /* FunctionSpace...cpp:24-26 */
    {
        gpis_ramp *ramps[4] = {&p->var, &p->mean_color, &p->mean_emission, &p->aniso_field};
        for (int i = 0; i < 4; ++i) {
            ramps[i]->min = 1.; ramps[i]->max = 500.; ramps[i]->start = 0.; ramps[i]->end = 1.;
            ramps[i]->min2 = 1.; ramps[i]->max2 = 500.; ramps[i]->start2 = 0.; ramps[i]->end2 = 1.;
        }
    }
    p->mean.type = GPIS_MEAN_SPHERICAL; p->mean.radius = 1.f;                        /* SCNM.cpp:19 */
    p->mean.scale = 1.f; p->mean.min = -FLT_MAX; p->mean.dir[0] = 1.;
    p->mean_additional = p->mean;
}

static int fail(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); return GPIS_ERR_INVALID_ARG; }

int oracle_create(const gpis_params *params, oracle_medium **out)
{
    if (!params || !out) return fail("null argument");
    if (params->abi_version != GPIS_ABI_VERSION) return fail("abi_version mismatch");
    if (params->correlation_context < 0 || params->correlation_context > 3) return fail("invalid correlation context");
    if (params->scheme_1d < 0 || params->scheme_1d > 2) return fail("invalid sparse conv sampling scheme");
    if (!(params->impulse_density >= 0.f)) return fail("invalid impulse_density");
    if (params->nonstationary && (params->ls_ramp_type < 0 || params->ls_ramp_type > GPIS_NOISE_RUST)) return fail("invalid ls ramp type");
    if ((params->var.enabled && (params->var.type < 0 || params->var.type > GPIS_NOISE_RUST)) || (params->mean_color.enabled && (params->mean_color.type < 0 || params->mean_color.type > GPIS_NOISE_RUST)) ||
        (params->mean_emission.enabled && (params->mean_emission.type < 0 || params->mean_emission.type > GPIS_NOISE_RUST)))
        return fail("invalid procedural noise type");
    if (params->var.enabled && !params->nonstationary) return fail("a var field needs the proc_nonstationary wrapper");
    if (params->aniso_field.enabled) {
        if (params->aniso_field.type < 0 || params->aniso_field.type > GPIS_NOISE_RUST) return fail("invalid procedural noise type");
        if (!params->nonstationary) return fail("an aniso field needs the proc_nonstationary wrapper");
        if (params->sampling_1d) return fail("an aniso field is built for 3D sampling only");
    }
    if (params->grid_nonstationary && (!params->nonstationary || params->var.enabled || params->aniso_field.enabled))
        return fail("the grid flavour needs nonstationary = 1 and carries no var / aniso field");
    if (params->kernel_type < 0 || params->kernel_type > 3) return fail("invalid kernel type");
    if (params->kernel_type != GPIS_KERNEL_SQUARED_EXPONENTIAL) {
        if (params->kernel_type == GPIS_KERNEL_MATERN && params->matern_v != 0.5f && params->matern_v != 1.5f && params->matern_v != 2.5f)
            return fail("Matern kernel: v must be 0.5, 1.5 or 2.5");
        if (params->isotropic_3d_sampling || params->sampling_1d || params->nonstationary || params->correlation_context == GPIS_CTX_RENEWAL_PLUS)
            return fail("Matern / Gabor kernels: world-space 3D sampling with context none / global / renewal only");
    }
    oracle_medium *m = (oracle_medium *)calloc(1, sizeof *m);
    if (!m) return GPIS_ERR_DEVICE;
    m->P = *params;
    const gpis_params *P = &m->P;
    /* SE fromJson, GPF.cpp:654-679 */
    m->l_conv = P->length_scale * sqrtf(2.f) / 2;
    memset(m->l2w, 0, sizeof m->l2w);
    memset(m->w2l, 0, sizeof m->w2l);
    if (!P->use_aniso_mtx) {
        for (int i = 0; i < 3; ++i) {
            m->l_aniso[i] = m->l_conv * P->aniso[i];
            float inv = 1.0f / m->l_aniso[i];
            if (isinf(inv) || isnan(inv)) inv = 0;    /* filterWithZero, GPF.hpp:151-160 */
            m->l_aniso_inv[i] = inv;
            M(m->l2w, i, i) = m->l_aniso[i];          /* off-diagonals: uninitialised in the reference, zero here (SURVEY §7 vi) */
            M(m->w2l, i, i) = m->l_aniso_inv[i];
        }
    } else {
        for (int i = 0; i < 9; ++i) m->l2w[i] = m->l_conv * P->aniso_mtx[i];
        eig_inverse3(m->l2w, m->w2l);
        float wT[9];
        mat3_transpose(m->w2l, wT);
        mat3_matmul_eig(wT, m->w2l, m->cov_mtx_inv);
        m->cov_mtx_inv_det = eig_det3(m->cov_mtx_inv);
    }
    mat3_transpose(m->l2w, m->l2w_T);
    mat3_transpose(m->w2l, m->w2l_T);
    m->kernel_scale = P->local_scale;
    m->n_impulses = (uint32_t)P->impulse_density;
    /* SparseConvolutionNoiseRealization ctor, SCN.cpp:8-37 */
    m->activate_conditioning = !P->single_realization &&
                               (P->correlation_context == GPIS_CTX_RENEWAL || P->correlation_context == GPIS_CTX_RENEWAL_PLUS);
    m->scheme_1d_eff = GPIS_UNI;
    if (!P->single_realization && P->sampling_1d)
        m->scheme_1d_eff = P->scheme_1d;
    m->multi_res = P->nonstationary && P->multi_resolution_grid;
    m->base = 2.5f;
    /* ramp, GPF.hpp:773-774 and maxVal GPF.cpp:124-130 */
    m->gabor_a = (float)(1.0 / P->gabor_a_inv);
    m->gabor_f = (float)(1.0 / P->gabor_f_inv);
    {   /* Vec3f::normalize(): *this *= 1/length */
        float l2 = 0.f; l2 += P->gabor_omega[0] * P->gabor_omega[0]; l2 += P->gabor_omega[1] * P->gabor_omega[1]; l2 += P->gabor_omega[2] * P->gabor_omega[2];
        float inv = 1.0f / sqrtf(l2);
        for (int i = 0; i < 3; ++i) m->gabor_omega[i] = P->gabor_omega[i] * inv;
    }
    m->ls_scale = 1.0 / (P->ls_end - P->ls_start);
    m->ls_offset = -P->ls_start * m->ls_scale;
    {   /* ProceduralNoiseVec::maxVal, GPF.cpp:124-138 */
        double mx = P->ls_max > P->ls_min ? P->ls_max : P->ls_min;
        if (P->ls_ramp_type == GPIS_RAMP_BOTTOM_TOP_LEFT_RIGHT)
            mx = mx * (P->ls_max2 > P->ls_min2 ? P->ls_max2 : P->ls_min2);
        if (P->ls_ramp_type == GPIS_NOISE_SANDSTONE || P->ls_ramp_type == GPIS_NOISE_RUST)
            mx = 1.;                                         /* GPF.cpp:130-137 */
        m->ls_maxval = (float)mx;
        if (P->grid_nonstationary)                               /* sparseConvNoiseMaxLateralScale, GPF.cpp:1422-1427 */
            m->ls_maxval = P->grid_surf_vol_amp_separate ? (P->grid_surf_ls_scale < P->grid_vol_ls_scale ? P->grid_vol_ls_scale : P->grid_surf_ls_scale) : 1.f;
    }
    /* prepareForRender, GPM.cpp:152-158 */
    int all_zero = 1;
    for (int c = 0; c < 3; ++c) {
        m->sigma_a[c] = P->sigma_a[c] * P->density;
        m->sigma_s[c] = P->sigma_s[c] * P->density;
        m->sigma_t[c] = m->sigma_a[c] + m->sigma_s[c];
        if (m->sigma_s[c] != 0.0f) all_zero = 0;
    }
    m->absorption_only = all_zero;
    for (int w = 0; w < 2; ++w) {
        const gpis_mean *mu = w ? &P->mean_additional : &P->mean;
        v3d d = {mu->dir[0], mu->dir[1], mu->dir[2]};
        double len = v3d_length(d);
        double inv = len > 0 ? 1.0 / len : 0.0;
        m->lin_dir[w][0] = d.x * inv; m->lin_dir[w][1] = d.y * inv; m->lin_dir[w][2] = d.z * inv;
    }
    m->threads = 1;
    *out = m;
    return GPIS_OK;
}
void oracle_destroy(oracle_medium *m) { if (m) free(m->grid_vox); free(m); }
/* test export: getUnscaledVariance at n world points (doubles) */
void oracle_grid_unscaled_variance(const oracle_medium *m, size_t n, const double *p3, float *out)
{
    for (size_t i = 0; i < n; ++i) { v3d p = {p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]}; out[i] = grid_unscaled_variance(m, p); }
}
int oracle_set_variance_grid(oracle_medium *m, const gpis_variance_grid *g, const float *voxels)
{
    if (!m || !g || !voxels) return fail("null argument");
    if (!m->P.grid_nonstationary) return fail("the medium was not created with grid_nonstationary = 1");
    if (g->dims[0] < 1 || g->dims[1] < 1 || g->dims[2] < 1 || g->interpolate < 0 || g->interpolate > 1) return fail("invalid grid");
    size_t n = (size_t)g->dims[0] * (size_t)g->dims[1] * (size_t)g->dims[2];
    float *v = (float *)malloc(n * sizeof(float));
    if (!v) return GPIS_ERR_DEVICE;
    memcpy(v, voxels, n * sizeof(float));
    free(m->grid_vox);
    m->grid_vox = v;
    m->grid = *g;
    return GPIS_OK;
}
void oracle_set_threads(oracle_medium *m, int n) { m->threads = n < 1 ? 1 : (n > ORACLE_MAX_THREADS ? ORACLE_MAX_THREADS : n); }

int oracle_get_derived(const oracle_medium *m, gpis_derived *o)
{
    memset(o, 0, sizeof *o);
    memcpy(o->world_to_local, m->w2l, sizeof m->w2l);
    memcpy(o->local_to_world, m->l2w, sizeof m->l2w);
    o->kernel_radius_world = cov_kernel_radius(m, 0, 1.0f);
    o->kernel_radius_iso = cov_kernel_radius(m, 1, 1.0f);
    v3f z = v3(0, 0, 0);
    o->norm3d_world = sqrtf(cov_variance3d(m, z, m->P.impulse_density, o->kernel_radius_world, 0, cov_world_sampling_scale(m)));
    o->norm3d_iso = sqrtf(cov_variance3d(m, z, m->P.impulse_density, o->kernel_radius_iso, 1, 1.0f));
    o->norm1d = sqrtf(cov_variance1d(m, z, m->P.impulse_density, o->kernel_radius_iso));
    if (m->P.nonstationary) {   /* position dependent: not a constant of the medium (gpis.h) */
        o->norm3d_world = 0.f;
        o->norm3d_iso = 0.f;
        o->norm1d = 0.f;
    }
    o->impulses_per_cell = m->n_impulses;
    o->activate_conditioning = m->activate_conditioning;
    o->effective_scheme_1d = m->scheme_1d_eff;
    o->multi_resolution = m->multi_res;
    o->fast_path = 0;
    return GPIS_OK;
}

/* ======================================================================================
 * Batch entry points (threaded)
 * ==================================================================================== */

typedef void (*range_fn)(oracle_medium *m, oracle_counters *cnt, size_t i0, size_t i1, void *ctx);
typedef struct { oracle_medium *m; int tid; size_t n; int nthreads; range_fn fn; void *ctx; } worker_arg;

static void *worker_main(void *a)
{
    worker_arg *w = (worker_arg *)a;
    /* interleaved chunks of 64 for load balance */
    const size_t chunk = 64;
    for (size_t i0 = (size_t)w->tid * chunk; i0 < w->n; i0 += chunk * (size_t)w->nthreads) {
        size_t i1 = i0 + chunk < w->n ? i0 + chunk : w->n;
        w->fn(w->m, &w->m->counters[w->tid], i0, i1, w->ctx);
    }
    return NULL;
}
static void parallel_for(oracle_medium *m, size_t n, range_fn fn, void *ctx)
{
    int nt = m->threads;
    if (nt <= 1 || n < 128) {
        fn(m, &m->counters[0], 0, n, ctx);
        return;
    }
    pthread_t th[ORACLE_MAX_THREADS];
    worker_arg args[ORACLE_MAX_THREADS];
    for (int t = 0; t < nt; ++t) {
        args[t].m = m; args[t].tid = t; args[t].n = n; args[t].nthreads = nt; args[t].fn = fn; args[t].ctx = ctx;
        pthread_create(&th[t], NULL, worker_main, &args[t]);
    }
    for (int t = 0; t < nt; ++t)
        pthread_join(th[t], NULL);
}

typedef struct { const gpis_ray_in *rays; gpis_seg_out *out; gpis_cond_coeff *coeff; uint8_t *vis; } seg_ctx;
static void sd_range(oracle_medium *m, oracle_counters *cnt, size_t i0, size_t i1, void *c)
{
    seg_ctx *x = (seg_ctx *)c;
    for (size_t i = i0; i < i1; ++i)
        sample_distance_one(m, cnt, &x->rays[i], &x->out[i], x->coeff ? &x->coeff[i] : NULL);
}
static void tr_range(oracle_medium *m, oracle_counters *cnt, size_t i0, size_t i1, void *c)
{
    seg_ctx *x = (seg_ctx *)c;
    for (size_t i = i0; i < i1; ++i) {
        medium_state st;
        state_from_ray(&x->rays[i], &st);
        cnt->n_seg++;
        x->vis[i] = (uint8_t)transmittance_one(m, cnt, &x->rays[i], &st);
    }
}
int oracle_sample_distance_batch(oracle_medium *m, size_t n, const gpis_ray_in *rays, gpis_seg_out *out, gpis_cond_coeff *coeff)
{
    if (!m || (n && (!rays || !out))) return fail("null argument");
    seg_ctx c = {rays, out, coeff, NULL};
    parallel_for(m, n, sd_range, &c);
    return GPIS_OK;
}
int oracle_transmittance_batch(oracle_medium *m, size_t n, const gpis_ray_in *rays, uint8_t *visible)
{
    if (!m || (n && (!rays || !visible))) return fail("null argument");
    seg_ctx c = {rays, NULL, NULL, visible};
    parallel_for(m, n, tr_range, &c);
    return GPIS_OK;
}

static ray_info info_of_query(const gpis_query *q)
{
    ray_info i = {q->pixel[0], q->pixel[1], q->spp, q->segment, q->scene_seed, q->info_t};
    return i;
}
static void realization_of(const oracle_medium *m, oracle_counters *cnt, const gpis_cond_coeff *c, realization *r)
{
    memset(r, 0, sizeof *r);
    r->m = m; r->cnt = cnt;
    r->c3 = *c; r->c1 = *c;
}
typedef struct { const gpis_query *q; float *value; int32_t *gp_id; float *grad; const float *tv; const float *tg; gpis_cond_coeff *co; } q_ctx;
static void ev_range(oracle_medium *m, oracle_counters *cnt, size_t i0, size_t i1, void *c)
{
    q_ctx *x = (q_ctx *)c;
    for (size_t i = i0; i < i1; ++i) {
        const gpis_query *q = &x->q[i];
        realization r;
        realization_of(m, cnt, &q->coeff, &r);
        int id;
        x->value[i] = evaluate_value(&r, v3(q->p[0], q->p[1], q->p[2]), q->t_segment, v3(q->dir[0], q->dir[1], q->dir[2]), info_of_query(q), &id);
        if (x->gp_id) x->gp_id[i] = id;
    }
}
static void eg_range(oracle_medium *m, oracle_counters *cnt, size_t i0, size_t i1, void *c)
{
    q_ctx *x = (q_ctx *)c;
    for (size_t i = i0; i < i1; ++i) {
        const gpis_query *q = &x->q[i];
        realization r;
        realization_of(m, cnt, &q->coeff, &r);
        v3f g = evaluate_gradient(&r, v3(q->p[0], q->p[1], q->p[2]), q->t_segment, v3(q->dir[0], q->dir[1], q->dir[2]), info_of_query(q));
        x->grad[3 * i] = g.x; x->grad[3 * i + 1] = g.y; x->grad[3 * i + 2] = g.z;
    }
}
static void cd_range(oracle_medium *m, oracle_counters *cnt, size_t i0, size_t i1, void *c)
{
    q_ctx *x = (q_ctx *)c;
    for (size_t i = i0; i < i1; ++i) {
        const gpis_query *q = &x->q[i];
        realization r;
        gpis_cond_coeff zero;
        memset(&zero, 0, sizeof zero);
        realization_of(m, cnt, &zero, &r);
        uint64_t e0 = cnt->n_eval;
        conditioning(&r, v3(q->p[0], q->p[1], q->p[2]), v3(q->dir[0], q->dir[1], q->dir[2]), x->tv[i],
                     v3(x->tg[3 * i], x->tg[3 * i + 1], x->tg[3 * i + 2]), info_of_query(q));
        x->co[i] = m->P.sampling_1d ? r.c1 : r.c3;
        x->co[i].n_evals = (uint32_t)(cnt->n_eval - e0);
    }
}
int oracle_eval_value_batch(oracle_medium *m, size_t n, const gpis_query *q, float *value, int32_t *gp_id)
{
    if (!m || (n && (!q || !value))) return fail("null argument");
    q_ctx c = {q, value, gp_id, NULL, NULL, NULL, NULL};
    parallel_for(m, n, ev_range, &c);
    return GPIS_OK;
}
int oracle_eval_gradient_batch(oracle_medium *m, size_t n, const gpis_query *q, float *grad3)
{
    if (!m || (n && (!q || !grad3))) return fail("null argument");
    q_ctx c = {q, NULL, NULL, grad3, NULL, NULL, NULL};
    parallel_for(m, n, eg_range, &c);
    return GPIS_OK;
}
int oracle_conditioning_batch(oracle_medium *m, size_t n, const gpis_query *q, const float *target_val, const float *target_grad3, gpis_cond_coeff *coeff_out)
{
    if (!m || (n && (!q || !target_val || !target_grad3 || !coeff_out))) return fail("null argument");
    q_ctx c = {q, NULL, NULL, NULL, target_val, target_grad3, coeff_out};
    parallel_for(m, n, cd_range, &c);
    return GPIS_OK;
}

typedef struct { const gpis_nee_query *q; float *pdf; float *grad; } nee_ctx;
static ray_info info_of_nee(const gpis_nee_query *q)
{
    ray_info i = {q->pixel[0], q->pixel[1], q->spp, q->segment, q->scene_seed, q->info_t};
    return i;
}
static void np_range(oracle_medium *m, oracle_counters *cnt, size_t i0, size_t i1, void *c)
{
    nee_ctx *x = (nee_ctx *)c;
    for (size_t i = i0; i < i1; ++i) {
        const gpis_nee_query *q = &x->q[i];
        realization r;
        realization_of(m, cnt, &q->coeff, &r);
        v3f rd = v3(q->ray_dir[0], q->ray_dir[1], q->ray_dir[2]), nn = v3(q->normal[0], q->normal[1], q->normal[2]), p = v3(q->p[0], q->p[1], q->p[2]);
        if (x->pdf) x->pdf[i] = nee_pdf(&r, rd, nn, p, q->t_segment, info_of_nee(q));
        if (x->grad) {
            v3f g = nee_grad(&r, rd, nn, p, info_of_nee(q));
            x->grad[3 * i] = g.x; x->grad[3 * i + 1] = g.y; x->grad[3 * i + 2] = g.z;
        }
    }
}
int oracle_nee_pdf_batch(oracle_medium *m, size_t n, const gpis_nee_query *q, float *pdf)
{
    if (!m || (n && (!q || !pdf))) return fail("null argument");
    nee_ctx c = {q, pdf, NULL};
    parallel_for(m, n, np_range, &c);
    return GPIS_OK;
}
int oracle_nee_grad_batch(oracle_medium *m, size_t n, const gpis_nee_query *q, float *grad3)
{
    if (!m || (n && (!q || !grad3))) return fail("null argument");
    nee_ctx c = {q, NULL, grad3};
    parallel_for(m, n, np_range, &c);
    return GPIS_OK;
}

/* MeanFunction::color / emission, GPF.hpp:849-857 (1 / 0 without the field; ramp noises have three equal components) */
#include "gpis_fs_oracle.inc"

int oracle_mean_color_emission(oracle_medium *m, size_t n, const double *p3, float *color3, float *emission3)
{
    if (!m || (n && !p3)) return fail("null argument");
    for (size_t i = 0; i < n; ++i) {
        v3d p = {p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]};
        if (color3) {
            double c[3] = {1., 1., 1.};
            if (m->P.mean_color.enabled) ramp_vec_of(&m->P.mean_color, p, c);
            for (int k = 0; k < 3; ++k) color3[3 * i + k] = (float)c[k];
        }
        if (emission3) {
            double e[3] = {0., 0., 0.};
            if (m->P.mean_emission.enabled) ramp_vec_of(&m->P.mean_emission, p, e);
            for (int k = 0; k < 3; ++k) emission3[3 * i + k] = (float)e[k];
        }
    }
    return GPIS_OK;
}

int oracle_xxhash32_batch(size_t n, int arity, const uint32_t *w, uint32_t *out)
{
    if (arity < 1 || arity > 4 || (n && (!w || !out))) return fail("bad arity");
    for (size_t i = 0; i < n; ++i) {
        const uint32_t *p = w + i * (size_t)arity;
        out[i] = arity == 1 ? xxhash32_1(p[0]) : arity == 2 ? xxhash32_2(p[0], p[1]) : arity == 3 ? xxhash32_3(p[0], p[1], p[2]) : xxhash32_4(p[0], p[1], p[2], p[3]);
    }
    return GPIS_OK;
}
int oracle_pcg32_stream_batch(size_t n, const uint64_t *state, uint32_t count, uint32_t *out)
{
    for (size_t i = 0; i < n; ++i) {
        pcg32 s;
        pcg_set_state(&s, state[i]);
        for (uint32_t k = 0; k < count; ++k)
            out[i * (size_t)count + k] = pcg_next_i(&s);
    }
    return GPIS_OK;
}

void oracle_get_counters(oracle_medium *m, uint64_t *n_eval, uint64_t *n_seg)
{
    uint64_t e = 0, s = 0;
    for (int t = 0; t < ORACLE_MAX_THREADS; ++t) { e += m->counters[t].n_eval; s += m->counters[t].n_seg; }
    if (n_eval) *n_eval = e;
    if (n_seg) *n_seg = s;
}
void oracle_reset_counters(oracle_medium *m) { memset(m->counters, 0, sizeof m->counters); }

/* ======================================================================================
 * Scene S: the synthetic estimator the benchmark renders.  This is the
 * build's own workload definition (the reference ships no GPIS scene); the camera follows
 * PinholeCamera::sampleDirection (src/core/cameras/PinholeCamera.cpp:70-86).
 * ==================================================================================== */

void oracle_default_scene_s(gpis_scene_s *s, uint32_t width, uint32_t height, uint32_t spp)
{
    memset(s, 0, sizeof *s);
    s->width = width; s->height = height;
    s->spp_begin = 0; s->spp_count = spp;
    s->scene_seed = 0xBA5EBA11u;
    s->tile_size = 16;
    s->cam_pos[0] = 0.f; s->cam_pos[1] = 0.f; s->cam_pos[2] = 4.f;
    s->cam_fov_deg = 35.f;
    s->bound_radius = 1.5f;
    s->light_dir[0] = 0.5f; s->light_dir[1] = 0.7f; s->light_dir[2] = 0.5f;
    s->light_radiance = 1.f;
    s->y_begin = 0; s->y_count = height;
    s->shard_index = 0; s->shard_count = 1;
}

/* ray / sphere(|x| = R) intersection in double; returns 0 on miss */
static int sphere_chord(v3f o, v3f d, float R, float *t0, float *t1)
{
    double ox = o.x, oy = o.y, oz = o.z, dx = d.x, dy = d.y, dz = d.z;
    double a = dx * dx + dy * dy + dz * dz;
    double b = ox * dx + oy * dy + oz * dz;
    double c = ox * ox + oy * oy + oz * oz - (double)R * (double)R;
    double disc = b * b - a * c;
    if (!(disc > 0.0))
        return 0;
    double sq = sqrt(disc);
    double ta = (-b - sq) / a, tb = (-b + sq) / a;
    if (tb <= 0.0)
        return 0;
    if (ta < 0.0) ta = 0.0;
    *t0 = (float)ta; *t1 = (float)tb;
    return 1;
}

int oracle_scene_s_primary(const gpis_scene_s *s, uint32_t x, uint32_t y, uint32_t spp, gpis_ray_in *ray, float *u_shadow)
{
    /* per-sample draws: one PCG stream seeded from (x, y, spp, scene_seed) → jx, jy, u_primary, u_shadow */
    pcg32 g;
    pcg_set_state(&g, (uint64_t)(uint32_t)(xxhash32_4(x, y, spp, s->scene_seed) + 1u));
    float jx = pcg_next_1d(&g), jy = pcg_next_1d(&g), u0 = pcg_next_1d(&g), u1 = pcg_next_1d(&g);
    float fov_rad = s->cam_fov_deg * (PI_F / 180.0f);
    float plane_dist = 1.0f / tanf(fov_rad * 0.5f);
    float ratio = (float)s->height / (float)s->width;
    float psx = 1.0f / (float)s->width;
    v3f local = v3(-1.0f + ((float)x + jx) * 2.0f * psx, ratio - ((float)y + jy) * 2.0f * psx, plane_dist);
    local = v3_normalized(local);
    v3f d = v3(local.x, local.y, -local.z);   /* camera looks down -z, right = +x, up = +y */
    v3f o = v3(s->cam_pos[0], s->cam_pos[1], s->cam_pos[2]);
    memset(ray, 0, sizeof *ray);
    ray->pos[0] = o.x; ray->pos[1] = o.y; ray->pos[2] = o.z;
    ray->dir[0] = d.x; ray->dir[1] = d.y; ray->dir[2] = d.z;
    ray->pixel[0] = x; ray->pixel[1] = y; ray->spp = spp; ray->segment = 0;
    ray->scene_seed = s->scene_seed; ray->info_t = 0.f; ray->u_jitter = u0;
    ray->first_scatter = 1; ray->bounce = 0; ray->last_val = 0.f; ray->last_gp_id = 0;
    *u_shadow = u1;
    float t0, t1;
    if (!sphere_chord(o, d, s->bound_radius, &t0, &t1))
        return 0;
    ray->near_t = t0; ray->far_t = t1;
    return 1;
}

/* shading + shadow segment of one primary result; returns radiance, and fills `sh` / returns
 * has_shadow when a shadow segment must be traced */
static int scene_s_shadow_ray(const gpis_scene_s *s, const gpis_ray_in *prim, const gpis_seg_out *o, float u_shadow,
                              gpis_ray_in *sh, float *cosl)
{
    if (!o->ok || o->exited)
        return 0;
    v3f l = v3_normalized(v3(s->light_dir[0], s->light_dir[1], s->light_dir[2]));
    double ax = o->aniso[0], ay = o->aniso[1], az = o->aniso[2];
    double len = sqrt(ax * ax + ay * ay + az * az);
    v3f n = v3((float)(ax / len), (float)(ay / len), (float)(az / len));
    float c = v3_dot(n, l);
    if (!(c > 0.f))
        return 0;
    *cosl = c;
    memset(sh, 0, sizeof *sh);
    sh->pos[0] = o->p[0]; sh->pos[1] = o->p[1]; sh->pos[2] = o->p[2];
    sh->dir[0] = l.x; sh->dir[1] = l.y; sh->dir[2] = l.z;
    float t0, t1;
    if (!sphere_chord(v3(o->p[0], o->p[1], o->p[2]), l, s->bound_radius, &t0, &t1))
        return 0;
    sh->near_t = 0.f; sh->far_t = t1;
    sh->pixel[0] = prim->pixel[0]; sh->pixel[1] = prim->pixel[1]; sh->spp = prim->spp;
    sh->segment = prim->segment + 1;                       /* TraceBase.cpp:547-548 */
    sh->scene_seed = prim->scene_seed;
    sh->info_t = prim->info_t + o->sample_t;               /* GPM.cpp:338 */
    sh->u_jitter = u_shadow;
    sh->first_scatter = 0;                                 /* state.advance() */
    sh->bounce = prim->bounce + 1;
    sh->last_val = o->last_val;
    sh->last_gp_id = o->gp_id;
    sh->last_aniso[0] = o->aniso[0]; sh->last_aniso[1] = o->aniso[1]; sh->last_aniso[2] = o->aniso[2];
    return 1;
}

/* gpis_scene_s::shard_index/shard_count (include/gpis.h): tile rows counted from y_begin, dealt round-robin */
static int scene_row_in_shard(const gpis_scene_s *s, uint32_t py)
{
    if (s->shard_count <= 1u) return 1;
    return ((py - s->y_begin) / s->tile_size) % s->shard_count == s->shard_index;
}
typedef struct { const gpis_scene_s *s; float *rad; uint32_t *hits; } render_ctx;
static void render_range(oracle_medium *m, oracle_counters *cnt, size_t i0, size_t i1, void *c)
{
    render_ctx *x = (render_ctx *)c;
    const gpis_scene_s *s = x->s;
    for (size_t idx = i0; idx < i1; ++idx) {
        uint32_t py = s->y_begin + (uint32_t)(idx / s->width), px = (uint32_t)(idx % s->width);
        if (!scene_row_in_shard(s, py)) continue;
        float acc = 0.f;
        uint32_t hits = 0;
        for (uint32_t k = 0; k < s->spp_count; ++k) {
            gpis_ray_in ray, sh;
            gpis_seg_out o;
            float u_shadow, cosl;
            if (!oracle_scene_s_primary(s, px, py, s->spp_begin + k, &ray, &u_shadow))
                continue;
            sample_distance_one(m, cnt, &ray, &o, NULL);
            if (o.ok && !o.exited)
                hits++;
            if (!scene_s_shadow_ray(s, &ray, &o, u_shadow, &sh, &cosl))
                continue;
            medium_state st;
            state_from_ray(&sh, &st);
            cnt->n_seg++;
            int vis = transmittance_one(m, cnt, &sh, &st);
            acc += cosl * (vis ? 1.f : 0.f) * s->light_radiance;
        }
        size_t pix = (size_t)py * s->width + px;
        x->rad[pix] += acc;
        if (x->hits) x->hits[pix] += hits;
    }
}
int oracle_render_scene_s(oracle_medium *m, const gpis_scene_s *s, float *radiance_sum, uint32_t *hit_count)
{
    if (!m || !s || !radiance_sum) return fail("null argument");
    if (s->y_begin + s->y_count > s->height) return fail("row range outside the image");
    if (s->shard_count > 1u && (s->shard_index >= s->shard_count || s->tile_size == 0)) return fail("bad shard");
    render_ctx c = {s, radiance_sum, hit_count};
    parallel_for(m, (size_t)s->y_count * s->width, render_range, &c);
    return GPIS_OK;
}

/* --------------------------------------------------------------------------------------
 * Multi-bounce estimator on scene S (PathTracer.cpp:62-75, TraceBase.cpp:346-386, 539-563;
 * BRDFPhaseFunction.cpp:27-96 with a Lambert BSDF, LambertBsdf.cpp:27-47).  Per sample one PCG
 * stream: jx, jy, then per bounce u_march, [u_shadow when NEE runs], rejection draws of the bounce.
 * ------------------------------------------------------------------------------------ */
static const float INV_PI_F = 1.0f / 3.1415926536f;   /* Angle.hpp:12 */

static float scene_paths_sample(oracle_medium *m, oracle_counters *cnt, const gpis_scene_s *s, uint32_t px, uint32_t py, uint32_t spp,
                                int max_bounces, float albedo)
{
    gpis_ray_in ray;
    float u_unused;
    if (!oracle_scene_s_primary(s, px, py, spp, &ray, &u_unused))
        return 0.f;
    /* re-create the stream after the two pixel-jitter draws */
    pcg32 g;
    pcg_set_state(&g, (uint64_t)(uint32_t)(xxhash32_4(px, py, spp, s->scene_seed) + 1u));
    (void)pcg_next_1d(&g); (void)pcg_next_1d(&g);
    v3f l = v3_normalized(v3(s->light_dir[0], s->light_dir[1], s->light_dir[2]));
    float throughput = 1.f, emission = 0.f;
    for (int bounce = 0; bounce < max_bounces; ++bounce) {
        ray.segment = (uint32_t)bounce;
        ray.u_jitter = pcg_next_1d(&g);
        gpis_seg_out o;
        sample_distance_one(m, cnt, &ray, &o, NULL);
        if (!o.ok)
            break;
        throughput *= o.weight[0];
        if (o.exited)
            break;
        /* shading frame about the sampled normal (BRDFPhaseFunction::setEventIsectInfo) */
        double ax = o.aniso[0], ay = o.aniso[1], az = o.aniso[2];
        double len = sqrt(ax * ax + ay * ay + az * az);
        v3f n = v3((float)(ax / len), (float)(ay / len), (float)(az / len));
        frame fr = frame_from_normal(n);
        v3f dir = v3(ray.dir[0], ray.dir[1], ray.dir[2]);
        v3f wi = v3_normalized(frame_to_local(&fr, v3(-dir.x, -dir.y, -dir.z)));
        v3f p = v3(o.p[0], o.p[1], o.p[2]);
        if (bounce < max_bounces - 1) {   /* NEE, TraceBase.cpp:546-550 */
            v3f wo = v3_normalized(frame_to_local(&fr, l));
            if (wi.z > 0.0f && wo.z > 0.0f) {
                float f = albedo * INV_PI_F * wo.z;
                float t0, t1;
                if (sphere_chord(p, l, s->bound_radius, &t0, &t1)) {
                    gpis_ray_in sh;
                    memset(&sh, 0, sizeof sh);
                    sh.pos[0] = p.x; sh.pos[1] = p.y; sh.pos[2] = p.z;
                    sh.dir[0] = l.x; sh.dir[1] = l.y; sh.dir[2] = l.z;
                    sh.near_t = 0.f; sh.far_t = t1;
                    sh.pixel[0] = px; sh.pixel[1] = py; sh.spp = spp;
                    sh.segment = (uint32_t)bounce + 1;
                    sh.scene_seed = s->scene_seed;
                    sh.info_t = ray.info_t + o.sample_t;
                    sh.u_jitter = pcg_next_1d(&g);
                    sh.first_scatter = 0;
                    sh.bounce = ray.bounce + 1;
                    sh.last_val = o.last_val;
                    sh.last_gp_id = o.gp_id;
                    sh.last_aniso[0] = o.aniso[0]; sh.last_aniso[1] = o.aniso[1]; sh.last_aniso[2] = o.aniso[2];
                    medium_state st;
                    state_from_ray(&sh, &st);
                    cnt->n_seg++;
                    int vis = transmittance_one(m, cnt, &sh, &st);
                    emission += throughput * (f * (vis ? 1.f : 0.f) * s->light_radiance);
                }
            }
        }
        /* phase sample (LambertBsdf::sample): fails when the ray arrives from below the surface */
        if (!(wi.z > 0.0f))
            break;
        float dx, dy, d2;
        do {
            dx = 2.f * pcg_next_1d(&g) - 1.f;
            dy = 2.f * pcg_next_1d(&g) - 1.f;
            d2 = dx * dx + dy * dy;
        } while (!(d2 < 1.f));
        float rem = 1.0f - d2;
        v3f wo_l = v3(dx, dy, sqrtf(rem > 0.f ? rem : 0.f));
        v3f w = v3_normalized(frame_to_global(&fr, wo_l));
        throughput *= albedo;
        /* ray = ray.scatter(p, w, 0); state advanced by sampleDistance (GPM.cpp:329-338) */
        float t0, t1;
        if (!sphere_chord(p, w, s->bound_radius, &t0, &t1))
            break;
        gpis_ray_in next;
        memset(&next, 0, sizeof next);
        next.pos[0] = p.x; next.pos[1] = p.y; next.pos[2] = p.z;
        next.dir[0] = w.x; next.dir[1] = w.y; next.dir[2] = w.z;
        next.near_t = 0.f; next.far_t = t1;
        next.pixel[0] = px; next.pixel[1] = py; next.spp = spp;
        next.scene_seed = s->scene_seed;
        next.info_t = ray.info_t + o.sample_t;
        next.first_scatter = 0;
        next.bounce = ray.bounce + 1;
        next.last_val = o.last_val;
        next.last_gp_id = o.gp_id;
        next.last_aniso[0] = o.aniso[0]; next.last_aniso[1] = o.aniso[1]; next.last_aniso[2] = o.aniso[2];
        ray = next;
    }
    return emission;
}

typedef struct { const gpis_scene_s *s; float *rad; int max_bounces; float albedo; } paths_ctx;
static void paths_range(oracle_medium *m, oracle_counters *cnt, size_t i0, size_t i1, void *c)
{
    paths_ctx *x = (paths_ctx *)c;
    const gpis_scene_s *s = x->s;
    for (size_t idx = i0; idx < i1; ++idx) {
        uint32_t py = s->y_begin + (uint32_t)(idx / s->width), px = (uint32_t)(idx % s->width);
        if (!scene_row_in_shard(s, py)) continue;
        float acc = 0.f;
        for (uint32_t k = 0; k < s->spp_count; ++k)
            acc += scene_paths_sample(m, cnt, s, px, py, s->spp_begin + k, x->max_bounces, x->albedo);
        x->rad[(size_t)py * s->width + px] += acc;
    }
}
int oracle_render_scene_s_paths(oracle_medium *m, const gpis_scene_s *s, int max_path_bounces, float albedo, float *radiance_sum)
{
    if (!m || !s || !radiance_sum || max_path_bounces < 1) return fail("bad argument");
    if (s->y_begin + s->y_count > s->height) return fail("row range outside the image");
    if (s->shard_count > 1u && (s->shard_index >= s->shard_count || s->tile_size == 0)) return fail("bad shard");
    paths_ctx c = {s, radiance_sum, max_path_bounces, albedo};
    parallel_for(m, (size_t)s->y_count * s->width, paths_range, &c);
    return GPIS_OK;
}

/* --------------------------------------------------------------------------------------
 * Scene S with the specular NEE coupling (TraceBase.cpp:346-420, BRDFPhaseFunction.cpp:27-96,
 * ConductorBsdf.cpp:59-139, Fresnel.hpp:102-123, InfiniteSphereCap).  Per sample one PCG stream:
 * jx, jy, u_march, then [cap z, azimuth rejection draws, u_shadow] for the light sample and
 * [u_shadow] for the phase sample when its direction falls inside the cap.
 * ------------------------------------------------------------------------------------ */
static float conductor_reflectance(float eta, float k, float cosThetaI)   /* Fresnel.hpp:102-123 */
{
    if (eta == 0 && k == 0)
        return 1;
    float cosThetaISq = cosThetaI * cosThetaI;
    float sinThetaISq = 1.0f - cosThetaISq > 0.0f ? 1.0f - cosThetaISq : 0.0f;
    float sinThetaIQu = sinThetaISq * sinThetaISq;
    float innerTerm = eta * eta - k * k - sinThetaISq;
    float q = innerTerm * innerTerm + 4.0f * eta * eta * k * k;
    float aSqPlusBSq = sqrtf(q > 0.0f ? q : 0.0f);
    float h = (aSqPlusBSq + innerTerm) * 0.5f;
    float a = sqrtf(h > 0.0f ? h : 0.0f);
    float Rs = ((aSqPlusBSq + cosThetaISq) - (2.0f * a * cosThetaI)) /
               ((aSqPlusBSq + cosThetaISq) + (2.0f * a * cosThetaI));
    float Rp = ((cosThetaISq * aSqPlusBSq + sinThetaIQu) - (2.0f * a * cosThetaI * sinThetaISq)) /
               ((cosThetaISq * aSqPlusBSq + sinThetaIQu) + (2.0f * a * cosThetaI * sinThetaISq));
    return 0.5f * (Rs + Rs * Rp);
}
static float power_heuristic(float pdf0, float pdf1) { return (pdf0 * pdf0) / (pdf0 * pdf0 + pdf1 * pdf1); }   /* SampleWarp.hpp */

static float scene_nee_sample(oracle_medium *m, oracle_counters *cnt, const gpis_scene_s *s, const gpis_surface_s *sf,
                              uint32_t px, uint32_t py, uint32_t spp)
{
    gpis_ray_in ray;
    float u_unused;
    if (!oracle_scene_s_primary(s, px, py, spp, &ray, &u_unused))
        return 0.f;
    pcg32 g;
    pcg_set_state(&g, (uint64_t)(uint32_t)(xxhash32_4(px, py, spp, s->scene_seed) + 1u));
    (void)pcg_next_1d(&g); (void)pcg_next_1d(&g); (void)pcg_next_1d(&g);   /* jx, jy, u_march */
    gpis_seg_out o;
    gpis_cond_coeff coeff;
    sample_distance_one(m, cnt, &ray, &o, &coeff);
    if (!o.ok || o.exited)
        return 0.f;
    const int scheme = o.scheme;
    v3f capDir = v3_normalized(v3(s->light_dir[0], s->light_dir[1], s->light_dir[2]));
    const float pdf_l = (0.5f * INV_PI_F) / (1.0f - sf->cap_cos);   /* uniformSphericalCapPdf */
    /* BRDFPhaseFunction::setEventIsectInfo */
    double ax = o.aniso[0], ay = o.aniso[1], az = o.aniso[2];
    double len = sqrt(ax * ax + ay * ay + az * az);
    v3f n = v3((float)(ax / len), (float)(ay / len), (float)(az / len));
    frame fr = frame_from_normal(n);
    v3f dir = v3(ray.dir[0], ray.dir[1], ray.dir[2]);
    v3f wi = v3_normalized(frame_to_local(&fr, v3(-dir.x, -dir.y, -dir.z)));
    v3f p = v3(o.p[0], o.p[1], o.p[2]);
    ray_info rinfo = {px, py, spp, ray.segment, s->scene_seed, ray.info_t + o.sample_t};
    const float F = sf->albedo * conductor_reflectance(sf->eta, sf->k, wi.z);
    /* the state volumeEstimateDirect works on: handleVolume's copy with the segment word + 1 */
    gpis_ray_in sh0;
    memset(&sh0, 0, sizeof sh0);
    sh0.pos[0] = p.x; sh0.pos[1] = p.y; sh0.pos[2] = p.z;
    sh0.near_t = 0.f;
    sh0.pixel[0] = px; sh0.pixel[1] = py; sh0.spp = spp;
    sh0.segment = ray.segment + 1;
    sh0.scene_seed = s->scene_seed;
    sh0.info_t = ray.info_t + o.sample_t;
    sh0.first_scatter = 0;
    sh0.bounce = ray.bounce + 1;
    sh0.last_val = o.last_val;
    sh0.last_gp_id = o.gp_id;
    sh0.last_aniso[0] = o.aniso[0]; sh0.last_aniso[1] = o.aniso[1]; sh0.last_aniso[2] = o.aniso[2];
    float L = 0.f;
    realization r;
    if (scheme != GPIS_UNI) {   /* volumeLightSample */
        float z = pcg_next_1d(&g) * (1.0f - sf->cap_cos) + sf->cap_cos;
        float dx, dy, d2;
        do {
            dx = 2.f * pcg_next_1d(&g) - 1.f;
            dy = 2.f * pcg_next_1d(&g) - 1.f;
            d2 = dx * dx + dy * dy;
        } while (!(d2 < 1.f) || !(d2 > 1e-12f));
        float rr = 1.0f - z * z;
        float rad = sqrtf(rr > 0.f ? rr : 0.f) / sqrtf(d2);
        frame cf = frame_from_normal(capDir);
        v3f d = frame_to_global(&cf, v3(dx * rad, dy * rad, z));
        v3f wo = v3_normalized(frame_to_local(&fr, d));
        v3f nl = v3_scale(v3_add(wi, wo), 0.5f);
        v3f nw = v3_normalized(frame_to_global(&fr, nl));
        realization_of(m, cnt, &coeff, &r);
        float pdf = nee_pdf(&r, dir, nw, p, o.sample_t, rinfo);
        float f = F * pdf;
        if (f != 0.0f) {
            realization_of(m, cnt, &coeff, &r);
            v3f gr = nee_grad(&r, dir, nw, p, rinfo);
            float t0, t1;
            if (sphere_chord(p, d, s->bound_radius, &t0, &t1)) {
                gpis_ray_in sh = sh0;
                sh.dir[0] = d.x; sh.dir[1] = d.y; sh.dir[2] = d.z;
                sh.far_t = t1;
                sh.u_jitter = pcg_next_1d(&g);
                sh.last_aniso[0] = gr.x; sh.last_aniso[1] = gr.y; sh.last_aniso[2] = gr.z;
                medium_state st;
                state_from_ray(&sh, &st);
                cnt->n_seg++;
                int vis = transmittance_one(m, cnt, &sh, &st);
                float e = (vis ? 1.f : 0.f) * sf->cap_radiance;
                if (e != 0.0f) {
                    float lightF = f * e / pdf_l;
                    if (scheme != GPIS_NEE)
                        lightF *= power_heuristic(pdf_l, pdf);
                    L += lightF;
                }
            }
        }
    }
    if (scheme != GPIS_NEE) {   /* volumePhaseSample: ConductorBsdf::sample mirrors about the sampled normal */
        v3f wo = v3(-wi.x, -wi.y, wi.z);
        float pdf_p = 1.0f;
        if (scheme != GPIS_UNI) {
            realization_of(m, cnt, &coeff, &r);
            pdf_p = nee_pdf(&r, dir, n, p, o.sample_t, rinfo);
        }
        v3f w = v3_normalized(frame_to_global(&fr, wo));
        float t0, t1;
        if (!(v3_dot(w, capDir) < sf->cap_cos) && sphere_chord(p, w, s->bound_radius, &t0, &t1)) {
            gpis_ray_in sh = sh0;
            sh.dir[0] = w.x; sh.dir[1] = w.y; sh.dir[2] = w.z;
            sh.far_t = t1;
            sh.u_jitter = pcg_next_1d(&g);
            medium_state st;
            state_from_ray(&sh, &st);
            cnt->n_seg++;
            int vis = transmittance_one(m, cnt, &sh, &st);
            float e = (vis ? 1.f : 0.f) * sf->cap_radiance;
            if (e != 0.0f) {
                float phaseF = e * F;
                if (scheme != GPIS_UNI)
                    phaseF *= power_heuristic(pdf_p, pdf_l);
                L += phaseF;
            }
        }
    }
    return L;
}

typedef struct { const gpis_scene_s *s; const gpis_surface_s *sf; float *rad; } nee_scene_ctx;
static void nee_scene_range(oracle_medium *m, oracle_counters *cnt, size_t i0, size_t i1, void *c)
{
    nee_scene_ctx *x = (nee_scene_ctx *)c;
    const gpis_scene_s *s = x->s;
    for (size_t idx = i0; idx < i1; ++idx) {
        uint32_t py = s->y_begin + (uint32_t)(idx / s->width), px = (uint32_t)(idx % s->width);
        if (!scene_row_in_shard(s, py)) continue;
        float acc = 0.f;
        for (uint32_t k = 0; k < s->spp_count; ++k)
            acc += scene_nee_sample(m, cnt, s, x->sf, px, py, s->spp_begin + k);
        x->rad[(size_t)py * s->width + px] += acc;
    }
}
int oracle_render_scene_s_nee(oracle_medium *m, const gpis_scene_s *s, const gpis_surface_s *surf, float *radiance_sum)
{
    if (!m || !s || !surf || !radiance_sum) return fail("null argument");
    if (s->y_begin + s->y_count > s->height) return fail("row range outside the image");
    if (s->shard_count > 1u && (s->shard_index >= s->shard_count || s->tile_size == 0)) return fail("bad shard");
    if (!(surf->cap_cos < 1.0f) || !(surf->cap_cos > -1.0f)) return fail("cap_cos must lie in (-1, 1)");
    nee_scene_ctx c = {s, surf, radiance_sum};
    parallel_for(m, (size_t)s->y_count * s->width, nee_scene_range, &c);
    return GPIS_OK;
}

/* ======================================================================================
 * Pinning surface
 * ==================================================================================== */
float oracle_conductor_reflectance(float eta, float k, float cosThetaI) { return conductor_reflectance(eta, k, cosThetaI); }
float oracle_power_heuristic(float pdf0, float pdf1) { return power_heuristic(pdf0, pdf1); }
float oracle_spherical_cap_pdf(float cosThetaMax) { return (0.5f * INV_PI_F) / (1.0f - cosThetaMax); }


void oracle_cell3d_draws(uint64_t state, uint32_t k, float *out)
{
    pcg32 s;
    pcg_set_state(&s, state);
    for (uint32_t i = 0; i < k; ++i) {
        float pz = pcg_next_1d(&s), py = pcg_next_1d(&s), px = pcg_next_1d(&s);
        float w = pcg_next_1d(&s);
        out[4 * i] = px; out[4 * i + 1] = py; out[4 * i + 2] = pz; out[4 * i + 3] = w;
    }
}
float oracle_normalized_uint(uint32_t i) { return normalized_uint(i); }
void oracle_tangent_frame(const float *n, float *o)
{
    frame f = frame_from_normal(v3(n[0], n[1], n[2]));
    o[0] = f.tangent.x; o[1] = f.tangent.y; o[2] = f.tangent.z;
    o[3] = f.bitangent.x; o[4] = f.bitangent.y; o[5] = f.bitangent.z;
    o[6] = f.normal.x; o[7] = f.normal.y; o[8] = f.normal.z;
}
void oracle_frame_to_local(const float *n, const float *p, float *o)
{
    frame f = frame_from_normal(v3(n[0], n[1], n[2]));
    v3f r = frame_to_local(&f, v3(p[0], p[1], p[2]));
    o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
void oracle_frame_to_global(const float *n, const float *p, float *o)
{
    frame f = frame_from_normal(v3(n[0], n[1], n[2]));
    v3f r = frame_to_global(&f, v3(p[0], p[1], p[2]));
    o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
void oracle_vec3_normalized(const float *v, float *o)
{
    v3f r = v3_normalized(v3(v[0], v[1], v[2]));
    o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
void oracle_sample_standard_normal2(uint64_t state, uint32_t draws, double *out)
{
    pcg32 s;
    pcg_set_state(&s, state);
    for (uint32_t d = 0; d < draws; ++d)
        rand_normal_2(&s, &out[2 * d], &out[2 * d + 1]);
}
void oracle_sample_xy_over_sqrt2(uint64_t state, double *out)
{
    pcg32 s;
    pcg_set_state(&s, state);
    double a, b;
    rand_normal_2(&s, &a, &b);
    out[0] = a / (double)sqrtf(2.f);
    out[1] = b / (double)sqrtf(2.f);
}
float oracle_eig_dist2_ab(const float *ab, const float *m) { return eig_dist2_ab(v3(ab[0], ab[1], ab[2]), m); }
float oracle_eig_dot_col(const float *ab, const float *m, int c) { return eig_dot_col(v3(ab[0], ab[1], ab[2]), m, c); }
void oracle_eig_matvec_div(const float *m, const float *v, float s, float *o)
{
    v3f r = v3_div(eig_matvec(m, v3(v[0], v[1], v[2])), s);
    o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
void oracle_eig_matvec_mul(const float *m, const float *v, float s, float *o)
{
    v3f r = v3_scale(eig_matvec(m, v3(v[0], v[1], v[2])), s);
    o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
void oracle_eig_matvec(const float *m, const float *v, float *o)
{
    v3f r = eig_matvec(m, v3(v[0], v[1], v[2]));
    o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
void oracle_eig_inverse3(const float *m, float *o) { eig_inverse3(m, o); }
void oracle_eig_second_deriv_inv(const float *m, float *o)
{
    float A[9], S[9];
    for (int i = 0; i < 9; ++i) A[i] = m[i] * 0.25f;
    for (int i = 0; i < 9; ++i) S[i] = -2.f * A[i];
    eig_inverse3(S, o);
}
void oracle_eig_scaled_matmul(float amplitude, const float *a, const float *b, float *o)
{
    float sa[9];
    for (int i = 0; i < 9; ++i) sa[i] = amplitude * a[i];
    mat3_matmul_eig(sa, b, o);
}
void oracle_eig_gram(const float *w, float *o, float *det)
{
    float wT[9];
    mat3_transpose(w, wT);
    mat3_matmul_eig(wT, w, o);
    *det = eig_det3(o);
}
void oracle_eig_scale_and_inverse(float l_conv, const float *a, float *l2w, float *w2l)
{
    for (int i = 0; i < 9; ++i) l2w[i] = l_conv * a[i];
    eig_inverse3(l2w, w2l);
}
void oracle_eig_invcov_scale(const float *m, float globalScale, float localScale, int isCov, float *o)
{
    float A[9];
    memcpy(A, m, sizeof A);
    float g2 = globalScale * globalScale;
    for (int i = 0; i < 9; ++i) A[i] /= g2;
    if (isCov) for (int i = 0; i < 9; ++i) A[i] *= 0.5f;
    float l2 = localScale * localScale;
    for (int i = 0; i < 9; ++i) A[i] /= l2;
    for (int i = 0; i < 9; ++i) A[i] *= 0.5f;
    memcpy(o, A, sizeof A);
}
/* test surface: the Bessel functions of the Matern v = 3/2 kernel */
void oracle_bessel_k01(double x, double *k0, double *k1) { bessel_k01(x, k0, k1); }
