// gm_fmt.h — number formatting of the SAM writer (host).  The reference prints XA / XP with "%g" (src/Driver.cpp:2196-2205):
// gm_put_g6 writes exactly what printf("%g") / std::to_chars(general, 6) write, but for the values a SAM row holds (1e-4 <= |v| < 1e6)
// with one exact 128-bit multiply instead of the general shortest-digits machinery (80 ns -> ~12 ns per number; two per record).
#pragma once
#include <charconv>
#include <cstdint>
#include <cstring>

static inline char* gm_put_u64(char* p, uint64_t v) {
    char tmp[24]; int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) *p++ = tmp[--k];
    return p;
}

// "%g" of v.  Exact: the 6 significant digits are round-half-even of the exact binary value, as printf rounds them.
static inline char* gm_put_g6(char* w, double v) {
    uint64_t bits; memcpy(&bits, &v, 8);
    const int be = (int)((bits >> 52) & 0x7FF);
    const double av = v < 0 ? -v : v;
    if (be == 0 || be == 0x7FF || !(av >= 1e-4 && av < 1e6)) return std::to_chars(w, w + 40, v, std::chars_format::general, 6).ptr;
    const uint64_t m = (bits & ((1ull << 52) - 1)) | (1ull << 52);
    const int s = 1075 - be;                                 // v = m * 2^-s; 33 <= s <= 66 in this range
    static const double p10[11] = { 1e-4, 1e-3, 1e-2, 1e-1, 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6 };
    static const uint64_t i10[10] = { 1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull };
    int e = -4;
    while (e < 5 && av >= p10[e + 5]) ++e;                   // a guess (the small powers are not exact doubles): checked by the digit count below
    uint64_t N;
    for (;;) {
        const int k = 5 - e;                                 // N = round(v * 10^k), 0 <= k <= 9 (10 after a step down from e = -4: to_chars then)
        if (k < 0 || k > 9) return std::to_chars(w, w + 40, v, std::chars_format::general, 6).ptr;
        const unsigned __int128 P = (unsigned __int128)m * i10[k];
        const unsigned __int128 half = (unsigned __int128)1 << (s - 1);
        const unsigned __int128 r = P & ((half << 1) - 1);
        N = (uint64_t)(P >> s);
        if (r > half || (r == half && (N & 1))) ++N;
        if (N < 100000ull) { --e; continue; }
        if (N >= 1000000ull) { ++e; continue; }              // the guess was low, or the rounding carried into a seventh digit
        break;
    }
    if (v < 0) *w++ = '-';
    char d[6];
    for (int i = 5; i >= 0; --i) { d[i] = (char)('0' + N % 10); N /= 10; }
    int last = 5;
    while (last > 0 && d[last] == '0') --last;               // trailing zeros go (no '#' flag)
    if (e >= 0) {
        for (int i = 0; i <= e; ++i) *w++ = d[i];
        if (last > e) { *w++ = '.'; for (int i = e + 1; i <= last; ++i) *w++ = d[i]; }
    } else {
        *w++ = '0'; *w++ = '.';
        for (int i = -1; i > e; --i) *w++ = '0';
        for (int i = 0; i <= last; ++i) *w++ = d[i];
    }
    return w;
}
