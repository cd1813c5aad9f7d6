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

}   // namespace gpis
