/* The vector division of the shading code (rt_amd/csrc/rt_device.h: divs) shares one reciprocal between the three
 * components: y = 1/s, q0 = a*y, r = fma(-s, q0, a), q = fma(r, y, q0), sign(q) := sign(a) ^ sign(s).  This program
 * restates it with the host's IEEE arithmetic (the same roundings as the device's: no contraction, fma exact) and
 * compares it bit for bit with a / s over random operands, wide exponent ranges, significands next to 1.0 and 2.0,
 * negative divisors and zero numerators of either sign.  It also shows that the sign step is needed for exactly one
 * class of operands: without it the only difference from IEEE division in the whole sweep is the sign of a zero
 * quotient (numerator -0 over a positive divisor, +0 over a negative one).  usage: div_shared_check [cases] */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t state = 88172645463325252ull;
static uint64_t rnd(void) { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; }
static double from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static uint64_t to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }
static double unit(void) { return (double)(rnd() >> 11) * 0x1.0p-53; }
static double edge(unsigned span) {      /* significand within `span` ulps of 1.0 or of 2.0 */
    return (rnd() & 1) ? from_bits(0x3ff0000000000000ull + rnd() % span) : from_bits(0x3fffffffffffffffull - rnd() % span);
}
static double div_unsigned(double a, double s, double y) {      /* the quotient steps alone */
    const double q0 = a * y;
    return fma(fma(-s, q0, a), y, q0);
}
static double div_shared(double a, double s, double y) {        /* rt_device.h: divs, one component */
    const uint64_t sg = (to_bits(a) ^ to_bits(s)) & 0x8000000000000000ull;
    return from_bits((to_bits(div_unsigned(a, s, y)) & 0x7fffffffffffffffull) | sg);
}

int main(int argc, char** argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 20000000L;
    long bad = 0, n = 0, zero_sign_only = 0, other_without_sign_step = 0;
    for (long i = 0; i < cases; ++i) {
        double a, s;
        switch (i % 7) {
        case 5: a = (rnd() & 1) ? 0.0 : -0.0; s = (rnd() & 1) ? unit() + 0x1.0p-60 : -(unit() + 0x1.0p-60); break;   /* zero numerators */
        case 6: a = 2.0 * unit() - 1.0; s = -(unit() + 0x1.0p-60); break;        /* negative divisors */
        case 0: a = 2.0 * unit() - 1.0; s = 1.8 * unit(); break;                 /* normalize(): component / length */
        case 1: a = unit(); s = unit(); break;                                   /* roulette: attenuation / p */
        case 2:                                                                  /* exponents 2^-400 .. 2^400 */
            a = from_bits((rnd() & 0x800fffffffffffffull) | ((uint64_t)(1023 - 400 + rnd() % 800) << 52));
            s = from_bits((rnd() & 0x000fffffffffffffull) | ((uint64_t)(1023 - 400 + rnd() % 800) << 52));
            break;
        case 3: a = from_bits((rnd() & 0x800fffffffffffffull) | (1023ull << 52)); s = edge(64); break;
        default: a = edge(4096); s = edge(4096); break;
        }
        if (s == 0.0) continue;
        const double q = div_shared(a, s, 1.0 / s), want = a / s, raw = div_unsigned(a, s, 1.0 / s);
        ++n;
        if (to_bits(raw) != to_bits(want)) {       /* without the sign step: only ever the sign of a zero quotient */
            if (want == 0.0 && raw == 0.0) ++zero_sign_only; else ++other_without_sign_step;
        }
        if (to_bits(q) != to_bits(want)) {
            if (bad < 10) printf("a=%a s=%a shared %a division %a\n", a, s, q, want);
            ++bad;
        }
    }
    printf("%ld cases, %ld mismatches\n", n, bad);
    printf("without the sign step: %ld zero quotients of the other sign, %ld other differences\n", zero_sign_only, other_without_sign_step);
    return bad != 0 || other_without_sign_step != 0 || zero_sign_only == 0;
}
