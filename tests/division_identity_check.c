/* Test infrastructure: the 1D lattice sum (csrc/gpis_path.inc, noise1d) replaces the reference's two divisions per impulse by the
 * loop-invariant den = 2 ls^2 (GPFunctions.cpp:835-845) with  y = RN(1/den); q = RN(a y); r = fma(-q, den, a); q' = fma(r, y, q).
 * Markstein's theorem says q' = RN(a / den).  This program compares q' with the compiler's IEEE division on random operand pairs
 * in the ranges the path produces (den in [2^-6, 2^6), a in [2^-30, 2^8)), with significands next to all-ones and next to powers of
 * two over-represented, in fp32 and fp64.  usage: division_identity_check <seed> <pairs>; prints "bad32=0 bad64=0" when exact. */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static inline uint64_t rng(uint64_t *s){ *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }
static float asf(uint32_t u){ float f; memcpy(&f,&u,4); return f; }
static double asd(uint64_t u){ double f; memcpy(&f,&u,8); return f; }
int main(int argc, char **argv){
    uint64_t s = 0x9E3779B97F4A7C15ULL ^ (uint64_t)atoi(argv[1]) * 0xD1B54A32D192ED03ULL;
    long n = atol(argv[2]); long bad32 = 0, bad64 = 0;
    for (long i = 0; i < n; ++i) {
        // den: float in [2^-6, 2^6), x: float in [2^-30, 2^8): mantissas random, with a share of special mantissas
        uint32_t md = (uint32_t)rng(&s) & 0x7FFFFF, mx = (uint32_t)rng(&s) & 0x7FFFFF;
        uint64_t sel = rng(&s);
        if ((sel & 15) == 0) md = 0x7FFFFF - ((uint32_t)(sel >> 8) & 7);      // mantissa near all ones
        if ((sel & 15) == 1) md = (uint32_t)(sel >> 8) & 7;                    // near a power of two
        if (((sel >> 4) & 15) == 0) mx = 0x7FFFFF - ((uint32_t)(sel >> 16) & 7);
        if (((sel >> 4) & 15) == 1) mx = (uint32_t)(sel >> 16) & 7;
        uint32_t ed = 121 + (uint32_t)((sel >> 24) % 12), ex = 97 + (uint32_t)((sel >> 32) % 38);
        float den = asf((ed << 23) | md), x = asf((ex << 23) | mx);
        float want = x / den;
        float r = 1.0f / den;
        float q = x * r; float rem = fmaf(-q, den, x); q = fmaf(rem, r, q);
        if (q != want) { if (bad32 < 5) printf("f32 mismatch x=%a den=%a want=%a got=%a\n", x, den, want, q); bad32++; }
        // f64: a = -2*ab (ab float), den as double of the float den
        double a = -2.0 * (double)asf(((100 + (uint32_t)((sel >> 40) % 30)) << 23) | ((uint32_t)rng(&s) & 0x7FFFFF));
        double dd = (double)den, wd = a / dd, rd = 1.0 / dd;
        double qd = a * rd; double remd = fma(-qd, dd, a); qd = fma(remd, rd, qd);
        if (qd != wd) { if (bad64 < 5) printf("f64 mismatch a=%a den=%a want=%a got=%a\n", a, dd, wd, qd); bad64++; }
    }
    printf("seed %s: n=%ld bad32=%ld bad64=%ld\n", argv[1], n, bad32, bad64);
    return 0;
}
