// host check of sg_dd.h against the C library: counts results that differ from pow(2, y) / atan2(y, x)
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define SG_DD_FN static inline
#include "sg_dd.h"
static uint64_t rng_state = 88172645463325252ull;
static double urand() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (rng_state >> 11) * (1.0 / 9007199254740992.0); }
int main(int argc, char **argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 2000000;
    long bad_p = 0, bad_a = 0, bad_a_near = 0, n_near = 0;
    for (long i = 0; i < N; i++) {
        const double y = urand() * 16.0;
        if (sg::sg_exp2_cr(y) != pow(2.0, y)) bad_p++;
        double a = (urand() - 0.5) * 2.0, b = (urand() - 0.5) * 2.0;
        if (i % 4 == 1) b = a * (1.0 + (urand() - 0.5) * 1e-15);      // next to a diagonal
        if (i % 4 == 2) b = -a * (1.0 + (urand() - 0.5) * 4e-16);
        if (i % 4 == 3) b = a * 1e-17 * urand();                       // next to an axis
        const bool near = i % 4 != 0;
        const bool differ = sg::sg_atan2_cr(a, b) != atan2(a, b);
        if (differ) { bad_a++; if (near) bad_a_near++; if (bad_a <= 5) printf("  atan2(%a, %a): %a vs libm %a\n", a, b, sg::sg_atan2_cr(a, b), atan2(a, b)); }
        n_near += near;
    }
    // special values
    const double sp[][2] = {{1, 1}, {1, -1}, {-1, 1}, {-1, -1}, {0, 1}, {0, -1}, {1, 0}, {-1, 0}, {0, 0}, {3e-7, 3e-7}, {5, 0.0}};
    int bad_s = 0;
    for (auto &s : sp) if (sg::sg_atan2_cr(s[0], s[1]) != atan2(s[0], s[1])) { bad_s++; printf("  special atan2(%g,%g) %a vs %a\n", s[0], s[1], sg::sg_atan2_cr(s[0], s[1]), atan2(s[0], s[1])); }
    printf("exp2: %ld of %ld differ from pow(2,y); atan2: %ld of %ld differ (%ld of %ld next to a diagonal or axis); specials differing %d\n", bad_p, N, bad_a, N, bad_a_near, n_near, bad_s);
    return 0;
}
