// Compares csrc/gpis_libm.hpp (compiled for the host, -mfma -ffp-contract=off so that only the fma calls written in the header
// fuse) with the libm this process links, bit for bit, on N pseudo-random arguments per function drawn from the ranges the path
// feeds them plus raw bit patterns.  Exit code 0 = no mismatch.  Built and run by tests/test_libm_replica_cpu.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "gpis_libm.hpp"
using namespace gpis;
static inline uint32_t fu(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline bool same(double a, double b) { return libm_asu(a) == libm_asu(b) || (a != a && b != b); }
int main(int argc, char **argv)
{
    uint64_t s = 88172645463325252ull, n = argc > 1 ? strtoull(argv[1], 0, 10) : 20000000ull;
    uint64_t bad_e = 0, bad_lf = 0, bad_l = 0, bad_s = 0, bad_c = 0, bad_sc = 0, bad_p = 0, bad_scf = 0;
    double (*volatile p_sin)(double) = std::sin;
    double (*volatile p_cos)(double) = std::cos;
    void (*volatile p_sincos)(double, double *, double *) = sincos;
    double (*volatile p_pow)(double, double) = std::pow;
    void (*volatile p_sincosf)(float, float *, float *) = sincosf;
    for (uint64_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) * 0x1p-53;
        double x;
        switch (i & 7) {
        case 0: x = -u * 40.0; break; case 1: x = -u * 800.0; break; case 2: x = -u * 1e-3; break; case 3: x = (u - 0.5) * 1400.0; break;
        case 4: x = -std::exp(u * 14.0 - 7.0); break; case 5: x = -u * 2.0; break; case 6: x = -700.0 - u * 60.0; break; default: x = libm_asd(s); break;
        }
        const double a = std::exp(x), b = exp_glibc(x);
        if (!same(a, b)) { if (bad_e < 5) printf("exp x=%a libm=%a mine=%a\n", x, a, b); ++bad_e; }
        float xf;
        switch (i & 3) { case 0: xf = (float)(u * 4.0 + 1e-3); break; case 1: xf = (float)std::exp(u * 40.0 - 20.0); break; case 2: xf = (float)(0.9 + u * 0.2); break; default: { uint32_t w = (uint32_t)s; memcpy(&xf, &w, 4); } }
        const float af = logf(xf), bf = logf_glibc(xf);
        if (fu(af) != fu(bf) && !(af != af && bf != bf)) { if (bad_lf < 5) printf("logf x=%a libm=%a mine=%a\n", xf, af, bf); ++bad_lf; }
        double xl;
        switch (i & 7) { case 0: xl = 1.0 - (double)(float)u; break; case 1: xl = u * 4.0 + 1e-9; break; case 2: xl = 0.9 + u * 0.2; break; case 3: xl = std::exp(u * 1400.0 - 700.0); break;
                         case 4: xl = 1.0 - u * 0x1p-20; break; case 5: xl = 1.0 + u * 0.07; break; case 6: xl = u * 0x1p-1022; break; default: xl = libm_asd(s); break; }
        const double al = std::log(xl), bl = log_glibc(xl);
        if (!same(al, bl)) { if (bad_l < 5) printf("log x=%a libm=%a mine=%a\n", xl, al, bl); ++bad_l; }
        double xs;
        switch (i & 7) { case 0: xs = u * 6.283185307179586; break; case 1: xs = (u - 0.5) * 20.0; break; case 2: xs = (u - 0.5) * 2e4; break; case 3: xs = (u - 0.5) * 2e8; break;
                         case 4: xs = (double)(float)u * 6.283185307179586; break; case 5: xs = (u - 0.5) * 0.3; break; case 6: xs = std::exp(u * 40.0 - 30.0); break;
                         default: xs = libm_asd(s); break; }
        if (std::fabs(xs) < 105414357.0 || xs != xs) {
            // through pointers, so that the compiler cannot merge the separate calls into one sincos
            const double as = p_sin(xs), bs = sin_glibc(xs), ac = p_cos(xs), bc = cos_glibc(xs);
            if (!same(as, bs)) { if (bad_s < 5) printf("sin x=%a libm=%a mine=%a\n", xs, as, bs); ++bad_s; }
            if (!same(ac, bc)) { if (bad_c < 5) printf("cos x=%a libm=%a mine=%a\n", xs, ac, bc); ++bad_c; }
            double s2, c2, s3, c3;
            p_sincos(xs, &s2, &c2);
            sincos_glibc(xs, &s3, &c3);
            if (!same(s2, s3) || !same(c2, c3)) { if (bad_sc < 5) printf("sincos x=%a libm=%a %a mine=%a %a\n", xs, s2, c2, s3, c3); ++bad_sc; }
        }
        float xsf;
        switch (i & 3) { case 0: xsf = (float)(u * 1.5707963267948966); break; case 1: xsf = (float)((u - 0.5) * 12.0); break; case 2: xsf = (float)((u - 0.5) * 238.0); break;
                         default: xsf = (float)std::exp(u * 30.0 - 28.0); break; }
        {
            float s4, c4, s5, c5;
            p_sincosf(xsf, &s4, &c4);
            sincosf_glibc(xsf, &s5, &c5);
            if (fu(s4) != fu(s5) || fu(c4) != fu(c5)) { if (bad_scf < 5) printf("sincosf x=%a libm=%a %a mine=%a %a\n", xsf, s4, c4, s5, c5); ++bad_scf; }
        }
        double xp, yp;
        switch (i & 7) { case 0: xp = u * 4.0 + 1e-6; yp = 3.0; break; case 1: xp = std::exp(u * 60.0 - 30.0); yp = 3.0; break; case 2: xp = u * 3.0 + 1e-9; yp = 2.0; break;
                         case 3: xp = (double)(float)(u * 2.5 + 0.01); yp = 3.0; break; case 4: xp = std::exp(u * 1400.0 - 700.0); yp = (double)((int)(s >> 3 & 7) - 3) + 0.5; break;
                         case 5: xp = u * 0x1p-1020; yp = 0.5 + u; break; case 6: xp = 1.0 + (u - 0.5) * 0x1p-10; yp = 1e5 * u; break;
                         default: xp = libm_asd(s & 0x7fffffffffffffffull); yp = (double)(float)(u * 8.0 - 4.0); break; }
        if (xp == xp && xp < __builtin_huge_val() && std::fabs(yp) >= 0x1p-65 && std::fabs(yp) < 0x1p63) {
            const double ap = p_pow(xp, yp), bp = pow_glibc(xp, yp);
            if (!same(ap, bp)) { if (bad_p < 5) printf("pow x=%a y=%a libm=%a mine=%a\n", xp, yp, ap, bp); ++bad_p; }
        }
    }
    printf("%llu inputs each: exp %llu, logf %llu, log %llu, sin %llu, cos %llu, sincos %llu, pow %llu, sincosf %llu mismatches\n", (unsigned long long)n, (unsigned long long)bad_e,
           (unsigned long long)bad_lf, (unsigned long long)bad_l, (unsigned long long)bad_s, (unsigned long long)bad_c, (unsigned long long)bad_sc, (unsigned long long)bad_p, (unsigned long long)bad_scf);
    return (bad_e || bad_lf || bad_l || bad_s || bad_c || bad_sc || bad_p || bad_scf) ? 1 : 0;
}
