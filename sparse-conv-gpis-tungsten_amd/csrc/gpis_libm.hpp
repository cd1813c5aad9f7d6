// gpis_libm.hpp — bit-for-bit re-implementations of the libm functions the reference's double-precision code calls on this path,
// AS THE HOST EVALUATES THEM: glibc 2.35 selects, on an x86-64 CPU with FMA (the GPU box's host and the build container), the
// variant of its C sources compiled with -mfma, in which the compiler contracted multiply-adds.  Which ones it contracted is read
// off the disassembly of libm.so.6 (__exp_fma, __logf_fma, …) and restated here with explicit fma calls, so the device produces the
// bits the oracle — and a Tungsten build on the same host — gets from exp() / logf().  Without them, ocml's versions differ in the
// last bit on a fraction of the arguments, and a last bit in a covariance entry or a length scale flips a Cholesky pivot or a
// zero crossing: with them the function-space path, the non-stationary length-scale field and the NEE density agree with the oracle
// bit for bit instead of "within tolerance".  tests/test_libm_replica_cpu.py compiles this header for the host and compares every
// function with libm on 10^8 arguments; tests/test_gpu_libm.py does the same through the device.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define GPIS_LIBM_FN __device__ __forceinline__
#define GPIS_LIBM_TAB __device__ const
#else
#define GPIS_LIBM_FN static inline
#define GPIS_LIBM_TAB static const
#endif

namespace gpis {

GPIS_LIBM_TAB uint64_t kExpTab[256] = {
#include "gpis_exp_table.inc"
};

GPIS_LIBM_FN uint64_t libm_asu(double x) { return __builtin_bit_cast(uint64_t, x); }
GPIS_LIBM_FN double libm_asd(uint64_t u) { return __builtin_bit_cast(double, u); }

// exp(double): glibc 2.35 sysdeps/ieee754/dbl-64/e_exp.c (N = 128 table, degree-5 polynomial), FMA variant (libm.so.6 __exp_fma):
// z + Shift, the two steps of the reduction, C2 + r C3, C4 + r C5, the two Horner steps and scale + scale tmp are fused; tail + r,
// r r, r2 r2 and the subnormal path are not.
GPIS_LIBM_FN double exp_glibc(double x)
{
    const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    uint32_t abstop = (uint32_t)(libm_asu(x) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u > 0x3eu) {
        if ((int32_t)(abstop - 0x3c9u) < 0) return 1.0 + x;                  // |x| < 2^-54
        if (abstop > 0x408u) {                                                // |x| >= 1024, inf, nan
            if (libm_asu(x) == 0xfff0000000000000ull) return 0.0;
            if (abstop == 0x7ffu) return 1.0 + x;
            return (libm_asu(x) >> 63) ? 0.0 : __builtin_huge_val();          // __math_uflow / __math_oflow
        }
        abstop = 0;                                                           // 512 <= |x| < 1024: the scale may leave the normal range
    }
    const double kd0 = __builtin_fma(x, InvLn2N, Shift);
    const uint64_t ki = libm_asu(kd0);
    const double kd = kd0 - Shift;
    double r = __builtin_fma(kd, NegLn2hiN, x);
    r = __builtin_fma(kd, NegLn2loN, r);
    const uint64_t idx = 2 * (ki & 127);
    const uint64_t sbits = kExpTab[idx + 1] + (ki << 45);
    const double tail = libm_asd(kExpTab[idx]);
    const double r2 = r * r;
    const double p23 = __builtin_fma(C3, r, C2), p45 = __builtin_fma(r, C5, C4);
    double tmp = __builtin_fma(p23, r2, tail + r);
    tmp = __builtin_fma(r2 * r2, p45, tmp);
    if (abstop == 0) {
        if ((ki & 0x80000000ull) == 0) {                                      // k > 0
            const double scale = libm_asd(sbits - (1009ull << 52));
            return __builtin_fma(scale, tmp, scale) * 0x1p1009;
        }
        const double scale = libm_asd(sbits + (1022ull << 52));               // k < 0: care in the subnormal range
        const double st = scale * tmp;
        double y = scale + st;
        if (y < 1.0) {
            const double hi = 1.0 + y;
            double lo = (scale - y) + st;
            lo = ((1.0 - hi) + y) + lo;
            y = (lo + hi) - 1.0;
            if (y == 0.0) y = 0.0;
        }
        return 0x1p-1022 * y;
    }
    const double scale = libm_asd(sbits);
    return __builtin_fma(scale, tmp, scale);
}

// logf(float): glibc 2.35 sysdeps/ieee754/flt-32/e_logf.c (N = 16 table {invc, logc}, cubic in double), FMA variant (__logf_fma):
// z invc - 1, logc + k Ln2 and the three Horner steps are fused.  Table and coefficients: __logf_data (= ARM optimized-routines
// logf_data.c).
GPIS_LIBM_TAB double kLogfTab[32] = {
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2, 0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2,
    0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3, 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3,
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4, 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5,
    0x1.0000000000000p+0, 0x0.0p+0, 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5, 0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4,
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3, 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3, 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,
    0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2,
};
GPIS_LIBM_FN float logf_glibc(float x)
{
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = __builtin_bit_cast(uint32_t, x);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -__builtin_huge_valf();                      // __math_divzerof(1)
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");   // __math_invalidf
        ix = __builtin_bit_cast(uint32_t, x * 0x1p23f);                        // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = kLogfTab[2 * i], logc = kLogfTab[2 * i + 1];
    const double z = (double)__builtin_bit_cast(float, iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = __builtin_fma((double)k, Ln2, logc);
    const double r2 = r * r;
    double y = __builtin_fma(A1, r, A2);
    y = __builtin_fma(A0, r2, y);
    y = __builtin_fma(y, r2, y0 + r);
    return (float)y;
}

// log(double): glibc 2.35 sysdeps/ieee754/dbl-64/e_log.c (N = 128 table {invc, logc}; degree-11 polynomial with a split leading
// term near 1), FMA variant (__log_fma): r = fma(z, invc, -1) (the table-2 path of the non-FMA build is not compiled in), and the
// sums below are fused exactly where the listing shows vfmadd.  Coefficients and table: __log_data (= ARM optimized-routines log_data.c).
GPIS_LIBM_TAB double kLogTab[256] = {
#include "gpis_log_table.inc"
};
GPIS_LIBM_FN double log_glibc(double x)
{
    const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    const double A0 = -0x1.0000000000001p-1, A1 = 0x1.555555551305bp-2, A2 = -0x1.fffffffeb4590p-3, A3 = 0x1.999b324f10111p-3, A4 = -0x1.55575e506c89fp-3;
    const double B0 = -0x1.0000000000000p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3, B3 = 0x1.999999995dd0cp-3, B4 = -0x1.55555556745a7p-3,
                 B5 = 0x1.24924a344de30p-3, B6 = -0x1.fffffa4423d65p-4, B7 = 0x1.c7184282ad6cap-4, B8 = -0x1.999eb43b068ffp-4, B9 = 0x1.78182f7afd085p-4,
                 B10 = -0x1.5521375d145cdp-4;
    uint64_t ix = libm_asu(x);
    const uint32_t top = (uint32_t)(ix >> 48);
    if (ix - 0x3fee000000000000ull < 0x3090000000000ull) {                    // 1 - 2^-4 <= x < 1 + 0x1.09p-4
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0;
        const double r2 = r * r, r3 = r * r2;
        const double p1 = __builtin_fma(r2, B3, __builtin_fma(r, B2, B1));
        const double p2 = __builtin_fma(r2, B6, __builtin_fma(r, B5, B4));
        double p3 = __builtin_fma(r2, B9, __builtin_fma(r, B8, B7));
        p3 = __builtin_fma(r3, B10, p3);
        p3 = __builtin_fma(p3, r3, p2);
        const double P = __builtin_fma(p3, r3, p1);
        const double rw = __builtin_fma(r, 0x1p27, r);                         // r + w, w = r 2^27
        const double rhi = __builtin_fma(-0x1p27, r, rw);                      // (r + w) - w
        const double rlo = r - rhi;
        const double rhi2 = rhi * rhi;
        const double hi = __builtin_fma(rhi2, B0, r);
        double lo = __builtin_fma(rhi2, B0, r - hi);
        lo = __builtin_fma(B0 * rlo, rhi + r, lo);
        const double y = __builtin_fma(P, r3, lo);
        return hi + y;
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
        if (ix * 2 == 0) return -__builtin_huge_val();
        if (ix == 0x7ff0000000000000ull) return x;
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return __builtin_nan("");
        ix = libm_asu(x * 0x1p52);
        ix -= 52ull << 52;
    }
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127u);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double invc = kLogTab[2 * i], logc = kLogTab[2 * i + 1];
    const double z = libm_asd(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w = __builtin_fma(kd, Ln2hi, logc);
    const double hi = w + r;
    const double lo = __builtin_fma(kd, Ln2lo, (w - hi) + r);
    const double r2 = r * r;
    const double q = __builtin_fma(__builtin_fma(r, A4, A3), r2, __builtin_fma(r, A2, A1));
    const double y = __builtin_fma(r * r2, q, __builtin_fma(r2, A0, lo));
    return y + hi;
}

// sin(double), cos(double), sincos(double): glibc 2.35 sysdeps/ieee754/dbl-64/s_sin.c and s_sincos.c (IBM Accurate Mathematical
// Library: table of sin / cos at k / 128 in two words, short polynomials around the table point).  sin and cos resolve to the FMA
// variants (__sin_fma, __cos_fma), in which every a * b + c of do_sin / do_cos / TAYLOR_SIN / reduce_sincos is fused, as the listing
// shows; sincos has NO FMA variant in this libm (one SSE2 body at sincos@@GLIBC_2.2.5), so there nothing is fused — and gcc merges
// a sin and a cos of the same argument into one sincos call, so which of the two a caller gets depends on whether it takes both:
// the device code calls sincos_glibc exactly where the CPU code takes both (Box-Muller, the Gabor kernel's gradient).  FUSE selects
// between the two.  |x| >= 105414357.85 (the source's __branred range, 0x419921FB in the high word) is outside what this path feeds
// the functions (angles in [0, 2 pi), Gabor phases of a few thousand) and returns a quiet NaN here so that a caller leaving the
// range is seen, not silently different.
GPIS_LIBM_TAB double kSinCosTab[440] = {
#include "gpis_sincos_table.inc"
};
namespace libm_sincos {
constexpr double big = 0x1.8p45, hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
constexpr double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7, cs2 = 0.5, cs4 = -0x1.5555555555535p-5, cs6 = 0x1.6c16bedd9e239p-10;
constexpr double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7, s3 = -0x1.a01a019db08b8p-13, s4 = 0x1.71de27b9a7ed9p-19, s5 = -0x1.addffc2fcdf59p-26;
constexpr double hpinv = 0x1.45f306dc9c883p-1, toint = 0x1.8p52, mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde973c000000p-27;
constexpr double pp3 = -0x1.cb3b398000000p-55, pp4 = -0x1.d747f23e32ed7p-83;
}
GPIS_LIBM_FN double libm_copysign(double mag, double sgn) { return libm_asd((libm_asu(mag) & 0x7fffffffffffffffull) | (libm_asu(sgn) & 0x8000000000000000ull)); }
GPIS_LIBM_FN double libm_fabs(double x) { return libm_asd(libm_asu(x) & 0x7fffffffffffffffull); }
// a b + c and c - a b, in one rounding (FUSE) or two; the translation units are compiled with -ffp-contract=off, so the two-rounding
// form stays two roundings
template <bool FUSE> GPIS_LIBM_FN double libm_mad(double a, double b, double c)
{
    if (FUSE) return __builtin_fma(a, b, c);
    const double ab = a * b;
    return ab + c;
}
template <bool FUSE> GPIS_LIBM_FN double libm_nmad(double a, double b, double c)
{
    if (FUSE) return __builtin_fma(-a, b, c);
    const double ab = a * b;
    return c - ab;
}
template <bool FUSE> GPIS_LIBM_FN double libm_taylor_sin(double xx, double x, double dx)
{
    using namespace libm_sincos;
    const double p = libm_mad<FUSE>(xx, libm_mad<FUSE>(xx, libm_mad<FUSE>(xx, libm_mad<FUSE>(xx, s5, s4), s3), s2), s1);
    const double t = libm_mad<FUSE>(xx, libm_mad<FUSE>(p, x, -(0.5 * dx)), dx);
    return x + t;
}
template <bool FUSE> GPIS_LIBM_FN double libm_do_sin(double x, double dx)
{
    using namespace libm_sincos;
    const double xold = x;
    const double ax = libm_fabs(x);
    if (ax < 0x1.020c49ba5e354p-3) return libm_taylor_sin<FUSE>(x * x, x, dx);    // |x| < 0.126
    if (x <= 0) dx = -dx;
    const double u = big + ax;
    const double xr = ax - (u - big);
    const double xx = xr * xr;
    const double s = xr + libm_mad<FUSE>(xr * xx, libm_mad<FUSE>(xx, sn5, sn3), dx);
    const double c = libm_mad<FUSE>(xr, dx, xx * libm_mad<FUSE>(xx, libm_mad<FUSE>(xx, cs6, cs4), cs2));
    const uint32_t k = (uint32_t)libm_asu(u) << 2;
    const double sn = kSinCosTab[k], ssn = kSinCosTab[k + 1], cs = kSinCosTab[k + 2], ccs = kSinCosTab[k + 3];
    const double cor = libm_mad<FUSE>(s, cs, libm_nmad<FUSE>(c, sn, libm_mad<FUSE>(s, ccs, ssn)));
    return libm_copysign(sn + cor, xold);
}
template <bool FUSE> GPIS_LIBM_FN double libm_do_cos(double x, double dx)
{
    using namespace libm_sincos;
    if (x < 0) dx = -dx;
    const double ax = libm_fabs(x);
    const double u = big + ax;
    const double xr = (ax - (u - big)) + dx;
    const double xx = xr * xr;
    const double s = libm_mad<FUSE>(xr * xx, libm_mad<FUSE>(xx, sn5, sn3), xr);
    const double c = xx * libm_mad<FUSE>(xx, libm_mad<FUSE>(xx, cs6, cs4), cs2);
    const uint32_t k = (uint32_t)libm_asu(u) << 2;
    const double sn = kSinCosTab[k], ssn = kSinCosTab[k + 1], cs = kSinCosTab[k + 2], ccs = kSinCosTab[k + 3];
    const double cor = libm_nmad<FUSE>(s, sn, libm_nmad<FUSE>(c, cs, libm_nmad<FUSE>(s, ssn, ccs)));
    return cs + cor;
}
// x = n pi/2 + a + da, |a| <= pi/4 (reduce_sincos of the source); returns n & 3
template <bool FUSE> GPIS_LIBM_FN int libm_reduce_sincos(double x, double *a, double *da)
{
    using namespace libm_sincos;
    const double t = libm_mad<FUSE>(x, hpinv, toint);
    const double xn = t - toint;
    const int n = (int)((uint32_t)libm_asu(t) & 3u);
    const double y = libm_nmad<FUSE>(xn, mp2, libm_nmad<FUSE>(xn, mp1, x));
    const double t2 = libm_nmad<FUSE>(xn, pp3, y);
    double db = libm_nmad<FUSE>(xn, pp3, y - t2);
    const double b = libm_nmad<FUSE>(xn, pp4, t2);
    db = db + libm_nmad<FUSE>(xn, pp4, t2 - b);
    *a = b;
    *da = db;
    return n;
}
template <bool FUSE> GPIS_LIBM_FN double libm_do_sincos(double a, double da, int n)
{
    const double r = (n & 1) ? libm_do_cos<FUSE>(a, da) : libm_do_sin<FUSE>(a, da);
    return (n & 2) ? -r : r;
}
GPIS_LIBM_FN double sin_glibc(double x)
{
    using namespace libm_sincos;
    const uint32_t k = (uint32_t)(libm_asu(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e500000u) return x;                                            // |x| < 2^-26
    if (k < 0x3feb6000u) return libm_do_sin<true>(x, 0.0);                    // |x| < 0.855469
    if (k < 0x400368fdu) return libm_copysign(libm_do_cos<true>(hp0 - libm_fabs(x), hp1), x);   // |x| < 2.426265
    if (k < 0x419921fbu) {
        double a, da;
        const int n = libm_reduce_sincos<true>(x, &a, &da);
        return libm_do_sincos<true>(a, da, n);
    }
    return __builtin_nan("");
}
GPIS_LIBM_FN double cos_glibc(double x)
{
    using namespace libm_sincos;
    const uint32_t k = (uint32_t)(libm_asu(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e400000u) return 1.0;                                          // |x| < 2^-27
    if (k < 0x3feb6000u) return libm_do_cos<true>(x, 0.0);
    if (k < 0x400368fdu) {
        const double y = hp0 - libm_fabs(x);
        const double a = y + hp1;
        const double da = (y - a) + hp1;
        return libm_do_sin<true>(a, da);
    }
    if (k < 0x419921fbu) {
        double a, da;
        const int n = libm_reduce_sincos<true>(x, &a, &da);
        return libm_do_sincos<true>(a, da, n + 1);
    }
    return __builtin_nan("");
}
GPIS_LIBM_FN void sincos_glibc(double x, double *sinx, double *cosx)
{
    using namespace libm_sincos;
    const uint32_t k = (uint32_t)(libm_asu(x) >> 32) & 0x7fffffffu;
    if (k < 0x400368fdu) {
        if (k < 0x3e400000u) { *sinx = x; *cosx = 1.0; return; }
        if (k < 0x3feb6000u) { *sinx = libm_do_sin<false>(x, 0.0); *cosx = libm_do_cos<false>(x, 0.0); return; }
        const double y = hp0 - libm_fabs(x);
        const double a = y + hp1;
        const double da = (y - a) + hp1;
        *sinx = libm_copysign(libm_do_cos<false>(a, da), x);
        *cosx = libm_do_sin<false>(a, da);
        return;
    }
    if (k < 0x419921fbu) {
        double a, da;
        const int n = libm_reduce_sincos<false>(x, &a, &da);
        *sinx = libm_do_sincos<false>(a, da, n);
        *cosx = libm_do_sincos<false>(a, da, n + 1);
        return;
    }
    *sinx = *cosx = __builtin_nan("");
}

// pow(double, double): glibc 2.35 sysdeps/ieee754/dbl-64/e_pow.c (log of x in two words from a 128-entry table and a degree-7
// polynomial, then the exp above applied to y log x with the low word as tail), FMA variant (__pow_fma): r = fma(z, invc, -1),
// t1 = fma(kd, Ln2hi, logc), lo1 = fma(kd, Ln2lo, logctail), lo3 = fma(ar, r, -ar2), the polynomial in fused pairs and its product
// with ar3 fused into the sum of the four low terms; ehi = y hi, elo = fma(y, lo, fma(y, hi, -ehi)).  Covered: x > 0 finite
// (subnormals included) or x = +0, y finite with 2^-65 <= |y| < 2^63 — what this path calls it with (cubes and squares of lengths
// and scales).  Anything else returns a quiet NaN, so a caller leaving that domain is seen.  Table: __pow_log_data.
GPIS_LIBM_TAB double kPowLogTab[384] = {
#include "gpis_powlog_table.inc"
};
GPIS_LIBM_FN double pow_glibc(double x, double y)
{
    const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    const double A0 = -0x1.0000000000000p-1, A1 = -0x1.5555555555560p-1, A2 = 0x1.0000000000006p-1, A3 = 0x1.999999959554ep-1, A4 = -0x1.555555529a47ap-1,
                 A5 = -0x1.2495b9b4845e9p+0, A6 = 0x1.0002b8b263fc3p+0;
    const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    uint64_t ix = libm_asu(x);
    const uint64_t iy = libm_asu(y);
    const uint32_t topx = (uint32_t)(ix >> 52), topy = (uint32_t)(iy >> 52) & 0x7ffu;
    if (topy - 0x3beu >= 0x80u) return __builtin_nan("");
    if (topx - 1u >= 0x7feu) {
        if (ix == 0) return (iy >> 63) ? __builtin_huge_val() : 0.0;          // +0 ^ y
        if (topx != 0) return __builtin_nan("");                              // negative, inf, nan
        ix = libm_asu(x * 0x1p52);                                            // subnormal: normalise
        ix -= 52ull << 52;
    }
    // log_inline
    const uint64_t tmp = ix - 0x3fe6955500000000ull;
    const int i = (int)((tmp >> 45) & 127u);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double z = libm_asd(iz), kd = (double)k;
    const double invc = kPowLogTab[3 * i], logc = kPowLogTab[3 * i + 1], logctail = kPowLogTab[3 * i + 2];
    const double r = __builtin_fma(z, invc, -1.0);
    const double t1 = __builtin_fma(kd, Ln2hi, logc);
    const double t2 = t1 + r;
    const double lo1 = __builtin_fma(kd, Ln2lo, logctail);
    const double lo2 = (t1 - t2) + r;
    const double ar = A0 * r, ar2 = r * ar, ar3 = r * ar2;
    const double lhi = t2 + ar2;
    const double lo3 = __builtin_fma(ar, r, -ar2);
    const double lo4 = (t2 - lhi) + ar2;
    const double q = __builtin_fma(ar2, __builtin_fma(ar2, __builtin_fma(r, A6, A5), __builtin_fma(r, A4, A3)), __builtin_fma(r, A2, A1));
    const double llo = __builtin_fma(ar3, q, ((lo1 + lo2) + lo3) + lo4);
    const double hi = lhi + llo;
    const double lo = (lhi - hi) + llo;
    const double ehi = y * hi;
    const double elo = __builtin_fma(y, lo, __builtin_fma(y, hi, -ehi));
    // exp_inline(ehi, elo, 0)
    uint32_t abstop = (uint32_t)(libm_asu(ehi) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u > 0x3eu) {
        if ((int32_t)(abstop - 0x3c9u) < 0) return 1.0 + ehi;
        if (abstop > 0x408u) return (libm_asu(ehi) >> 63) ? 0.0 : __builtin_huge_val();
        abstop = 0;
    }
    const double kd0 = __builtin_fma(ehi, InvLn2N, Shift);
    const uint64_t ki = libm_asu(kd0);
    const double kde = kd0 - Shift;
    double re = __builtin_fma(kde, NegLn2hiN, ehi);
    re = __builtin_fma(kde, NegLn2loN, re);
    re = elo + re;
    const uint64_t idx = 2 * (ki & 127);
    const uint64_t sbits = kExpTab[idx + 1] + (ki << 45);
    const double tail = libm_asd(kExpTab[idx]);
    const double r2 = re * re;
    const double p23 = __builtin_fma(C3, re, C2), p45 = __builtin_fma(re, C5, C4);
    double tm = __builtin_fma(p23, r2, tail + re);
    tm = __builtin_fma(r2 * r2, p45, tm);
    if (abstop == 0) {
        if ((ki & 0x80000000ull) == 0) {
            const double scale = libm_asd(sbits - (1009ull << 52));
            return __builtin_fma(scale, tm, scale) * 0x1p1009;
        }
        const double scale = libm_asd(sbits + (1022ull << 52));
        const double st = scale * tm;
        double yy = scale + st;
        if (yy < 1.0) {
            const double one = 1.0;
            const double h = one + yy;
            double l = (scale - yy) + st;
            l = ((one - h) + yy) + l;
            yy = (l + h) - one;
            if (yy == 0.0) yy = 0.0;
        }
        return 0x1p-1022 * yy;
    }
    const double scale = libm_asd(sbits);
    return __builtin_fma(scale, tm, scale);
}

// sincosf(float): glibc 2.35 sysdeps/x86/fpu/sincosf_poly.h + sysdeps/ieee754/flt-32/s_sincosf.c (ARM optimized-routines sincosf:
// reduction by pi/2 and two short polynomials in double), FMA variant (__sincosf_fma; a cosf and a sinf of the same angle are one
// sincosf call in a gcc build).  |y| < 120 is covered (the rotation angle of the anisotropy field is at most pi / 2); beyond it a
// quiet NaN comes back.  Table: __sincosf_table ([1] = the coefficients for the quadrants where the cosine changes sign).
GPIS_LIBM_TAB double kSinCosfTab[2][8] = {
    {0x1.0000000000000p+0, -0x1.ffffffd0c621cp-2, -0x1.555545995a603p-3, 0x1.55553e1068f19p-5, 0x1.1107605230bc4p-7, -0x1.6c087e89a359dp-10,
     -0x1.994eb3774cf24p-13, 0x1.99343027bf8c3p-16},
    {-0x1.0000000000000p+0, 0x1.ffffffd0c621cp-2, -0x1.555545995a603p-3, -0x1.55553e1068f19p-5, 0x1.1107605230bc4p-7, 0x1.6c087e89a359dp-10,
     -0x1.994eb3774cf24p-13, -0x1.99343027bf8c3p-16},
};
GPIS_LIBM_FN void libm_sincosf_poly(double x, double x2, int tab, int n, float *sinp, float *cosp)
{
    const double *T = kSinCosfTab[tab];           // c0, c1, {s1, c2}, {s2, c3}, {s3, c4}
    const double x3 = x * x2, x4 = x2 * x2, x5 = x2 * x3, x6 = x2 * x4;
    const double c1v = __builtin_fma(x2, T[1], T[0]);
    const double sv = __builtin_fma(__builtin_fma(T[6], x2, T[4]), x5, __builtin_fma(x3, T[2], x));
    const double cv = __builtin_fma(__builtin_fma(T[7], x2, T[5]), x6, __builtin_fma(x4, T[3], c1v));
    if (n & 1) { *cosp = (float)sv; *sinp = (float)cv; }
    else { *sinp = (float)sv; *cosp = (float)cv; }
}
GPIS_LIBM_FN void sincosf_glibc(float y, float *sinp, float *cosp)
{
    const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p+0;
    const uint32_t top = (__builtin_bit_cast(uint32_t, y) >> 20) & 0x7ffu;
    double x = (double)y;
    if (top < 0x3f4u) {                                                       // |y| < pi / 4
        if (top < 0x398u) { *sinp = y; *cosp = 1.0f; return; }                // |y| < 2^-12
        libm_sincosf_poly(x, x * x, 0, 0, sinp, cosp);
        return;
    }
    if (top < 0x42fu) {                                                       // |y| < 120
        const double r = x * hpi_inv;
        const int n = ((int32_t)r + 0x800000) >> 24;
        x = __builtin_fma(-(double)n, hpi, x);
        const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        libm_sincosf_poly(x * sgn, x * x, (n & 2) ? 1 : 0, n, sinp, cosp);
        return;
    }
    *sinp = *cosp = __builtin_nanf("");
}

}   // namespace gpis
