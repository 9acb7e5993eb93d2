/*
 * selftest.c — the CPU oracle under AddressSanitizer + UBSan (host code only).  TEST INFRASTRUCTURE.
 *
 * Runs the two upstream known-answer vectors and a seeded ragged batch with padding through
 * rnnt_oracle_f32 / rnnt_oracle_f64 (oracle/rnnt_oracle.c is #included, so every line of it is
 * instrumented), checks costs and a few invariants, and exits non-zero on any mismatch; the
 * sanitizers abort on any out-of-bounds access, use of uninitialised padding or undefined arithmetic.
 * Built and run by tests/test_oracle.py::test_oracle_under_sanitizers (make -C oracle selftest_asan).
 */
#include <stdio.h>

#include "rnnt_oracle.c"

static unsigned long long rng_state = 88172645463325252ull;
static double rnd(void) { /* xorshift64*, uniform in [-2, 2) */
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return ((double)((rng_state * 2685821657736338717ull) >> 11) / 9007199254740992.0) * 4.0 - 2.0;
}

static int fail(const char *what, double got, double want) {
    fprintf(stderr, "selftest FAILED: %s: got %.9g, want %.9g\n", what, got, want);
    return 1;
}

int main(void) {
    /* KAT-1: B1 T2 U2 V5, cost 4.4956668 (tests/cases.py) */
    const double k1[30] = {0.1, 0.6, 0.1, 0.1, 0.1, 0.1, 0.1, 0.6, 0.1, 0.1, 0.1, 0.1, 0.2, 0.8, 0.1,
                           0.1, 0.6, 0.1, 0.1, 0.1, 0.1, 0.1, 0.2, 0.1, 0.1, 0.7, 0.1, 0.2, 0.1, 0.1};
    int lab1[2] = {1, 2}, al1[1] = {2}, ll1[1] = {2};
    double c1[1], g1[30], b1[1];
    if (rnnt_oracle_f64(k1, lab1, al1, ll1, 1, 2, 3, 5, 0, c1, g1, b1, 1) != 0) return fail("kat1 status", 1, 0);
    if (fabs(c1[0] - 4.495666773770733) > 1e-12) return fail("kat1 cost", c1[0], 4.495666773770733);
    if (fabs(b1[0] + c1[0]) > 1e-12) return fail("kat1 beta(0,0)", b1[0], -c1[0]);
    if (fabs(g1[0] + 0.13116686) > 1e-7) return fail("kat1 grad[0]", g1[0], -0.13116686);

    /* seeded ragged batch: B=5, T<=9, U<=4, V=11, blank=3; padding holds NaN and must be ignored */
    enum { B = 5, T = 9, U1 = 5, V = 11 };
    static float a32[B * T * U1 * V], g32[B * T * U1 * V];
    static double a64[B * T * U1 * V], g64[B * T * U1 * V];
    int labels[B * (U1 - 1)], al[B] = {9, 1, 4, 7, 9}, ll[B] = {4, 0, 2, 4, 1};
    for (int i = 0; i < B * T * U1 * V; ++i) a64[i] = rnd();
    for (int b = 0; b < B; ++b) {
        for (int u = 0; u < U1 - 1; ++u) labels[b * (U1 - 1) + u] = u < ll[b] ? (int)((rnd() + 2.0) * 2.4) % 3 + 4 * (u % 2) : 0;
        for (int t = 0; t < T; ++t)
            for (int u = 0; u < U1; ++u)
                if (t >= al[b] || u > ll[b])
                    for (int v = 0; v < V; ++v) a64[((b * T + t) * U1 + u) * V + v] = NAN;
    }
    for (int i = 0; i < B * T * U1 * V; ++i) a32[i] = (float)a64[i];
    float c32[B], b32[B];
    double c64[B], b64[B];
    if (rnnt_oracle_f64(a64, labels, al, ll, B, T, U1, V, 3, c64, g64, b64, 2) != 0) return fail("f64 status", 1, 0);
    if (rnnt_oracle_f32(a32, labels, al, ll, B, T, U1, V, 3, c32, g32, b32, 3) != 0) return fail("f32 status", 1, 0);
    for (int b = 0; b < B; ++b) {
        if (!(c64[b] > 0.0) || !isfinite(c64[b])) return fail("cost finite and positive", c64[b], 1.0);
        if (fabs(c32[b] - c64[b]) > 1e-5 * c64[b]) return fail("f32 vs f64 cost", c32[b], c64[b]);
        if (fabs(b64[b] + c64[b]) > 1e-10) return fail("alpha/beta likelihoods agree", b64[b], -c64[b]);
        for (int t = 0; t < T; ++t)
            for (int u = 0; u < U1; ++u) {
                double s = 0.0;
                for (int v = 0; v < V; ++v) {
                    const double g = g64[((b * T + t) * U1 + u) * V + v];
                    if (t >= al[b] || u > ll[b]) {
                        if (g != 0.0) return fail("padding gradient", g, 0.0);
                    } else {
                        s += g;
                        if (fabs(g32[((b * T + t) * U1 + u) * V + v] - g) > 1e-4) return fail("f32 vs f64 grad", g32[((b * T + t) * U1 + u) * V + v], g);
                    }
                }
                if (fabs(s) > 1e-12) return fail("row of the gradient sums to zero", s, 0.0);
            }
    }
    /* forward only (grads == NULL) and invalid arguments */
    if (rnnt_oracle_f64(a64, labels, al, ll, B, T, U1, V, 3, c64, NULL, NULL, 1) != 0) return fail("forward-only status", 1, 0);
    printf("oracle selftest ok (kat1 %.9f, batch costs %.6f %.6f %.6f %.6f %.6f)\n", c1[0], c64[0], c64[1], c64[2], c64[3], c64[4]);
    return 0;
}
