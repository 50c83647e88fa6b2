// checks gnumap_amd/csrc/gm_fmt.h: gm_put_g6 must write what printf("%g") writes (the reference's XA / XP columns, src/Driver.cpp:2196-2205)
#include "gm_fmt.h"
#include <cmath>
#include <cstdio>
#include <random>
int main() {
    std::mt19937_64 rng(7);
    char a[64], b[64]; long bad = 0, n = 0;
    auto chk = [&](double v) { char* e = gm_put_g6(a, v); *e = 0; snprintf(b, sizeof b, "%g", v); ++n; if (strcmp(a, b)) { if (bad < 10) printf("MISMATCH %.17g: %s vs %s\n", v, a, b); ++bad; } };
    for (long i = 0; i < 1500000; ++i) {
        const double u = (double)(rng() >> 11) / 9007199254740992.0;
        const int dec = (int)(rng() % 13) - 5;
        const float f = (float)(u * std::pow(10.0, dec));
        chk((double)f); chk(-(double)f); chk((double)f * (1.0 / 0.37)); chk(u * std::pow(10.0, dec));
    }
    for (int e = -6; e <= 7; ++e)
        for (int k = -3; k <= 3; ++k) {
            const double p = std::pow(10.0, e);
            chk(std::nextafter(p, k < 0 ? 0 : 1e300)); chk(p); chk(p * (1 + k * 1e-7)); chk(p * 9.999995); chk(p * 9.9999949); chk(p * 1.5); chk(p * 1.000005); chk(p * 2.000015);
        }
    for (long i = 0; i < 500000; ++i) {                      // exact halves of the sixth digit, and short binary fractions
        const double x = (double)(100000 + rng() % 900000) + 0.5; const int dec = (int)(rng() % 10) - 9;
        chk(x * std::pow(10.0, dec)); chk(std::ldexp((double)(rng() % (1 << 24)), -(int)(rng() % 30)));
    }
    const double fixed[] = { 0.0, -0.0, 1.0, 0.5, 999999.5, 999999.4999, 0.0001, 0.00009999995, 1e6, 123456.5, 1234565e-1, 0.1, 100000, INFINITY, -INFINITY, NAN, 5e-324, 1e-5, 1e300, 2.5e-7 };
    for (double v : fixed) chk(v);
    printf("%ld values, %ld mismatches\n", n, bad);
    return bad != 0;
}
