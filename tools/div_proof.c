/*
 * div_proof.c - is  r0 = rcp(b); r1 = r0 + r0 (1 - b r0); q0 = a r1; q1 = q0 + r1 (a - b q0)  the correctly rounded
 * quotient a / b for every pair of binary32 numbers with 0 < a <= b?  (The discriminator's z = min / max,
 * multifm/fast_atan2f.c:117-119; tsl-sdr_amd/csrc/mfm_numerics.h mfm_div_unit.)
 *
 * All operations scale exactly with powers of two (operands are conversions of int32: 1 <= a <= b <= 2^31, nothing
 * underflows), so only the 24-bit significands A, B in [2^23, 2^24) matter, in two cases: A >= B (quotient in [1, 2)) and
 * A < B (quotient in (1/2, 1)).
 *
 * q1 = RN(v) with v = q0 + e1 r1 exactly (one fma; e1 = a - b q0 is exact because q0 is within 1.5 ulp of a / b).
 * |v - a/b| = |e1 / b| |b r1 - 1| <= 1.5 ulp x 2^-24 (1 + eps) < 2^-23 ulp for any r0 within 2 ulp of 1 / b.  RN(v) can
 * differ from RN(a / b) only if a / b lies that close to a midpoint of two neighbouring floats, i.e. (quotient in [1, 2),
 * midpoints t 2^-24 with t odd)  |A 2^24 - t B| <= B 2^-22 <= 4, and the same with 2^25 for the other case.  For every B
 * there are at most a handful of such A: the solutions of A 2^k = rho (mod B), 0 < |rho| <= RHO_MAX.  This program
 * enumerates them for all 2^23 B, for r0 = RN(1 / B) moved by -2 .. +2 ulp (v_rcp_f32 is specified to 1 ulp), and runs
 * the float sequence on each.  It also prints the pairs (capped) so that a GPU test can run the same hard cases through the
 * real v_rcp_f32 (tests/test_numerics_host.py, tests/test_gpu_parity.py).
 *
 *   gcc -O2 -ffp-contract=off -o div_proof div_proof.c -lm -lpthread && ./div_proof [steps] [threads] [dump | -] [rcp table]
 * steps = residual steps behind q0 (2 = the form above with one more, the shipped form until round 3; 1 = the form above).
 * rcp table = "rn" (the correctly rounded reciprocal for every B) or a file of 2^23 signed bytes written by tools/rcp_check.hip on the GPU: by how many ulps v_rcp_f32(B) differs from
 * RN(1 / B); with it every sequence runs on the reciprocal the device really returns instead of on five candidates.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RHO_MAX 8

static int g_steps = 1, g_threads = 8;
static const int8_t *g_rcp_dev; /* [2^23]: ulps by which the device's v_rcp_f32(B) differs from RN(1 / B) (tools/rcp_check.hip) */

static inline float step_ulps(float x, int n)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    u += (uint32_t)n;
    memcpy(&x, &u, 4);
    return x;
}

static inline float quotient(float a, float b, float r0, int steps)
{
    const float e0 = fmaf(-b, r0, 1.0f);
    const float r1 = fmaf(e0, r0, r0);
    float q = a * r1;
    for (int i = 0; i < steps; i++) {
        const float e = fmaf(-b, q, a);
        q = fmaf(e, r1, q);
    }
    return q;
}

/* x^-1 mod m, m odd */
static uint64_t inv_mod(uint64_t x, uint64_t m)
{
    int64_t t = 0, nt = 1, r = (int64_t)m, nr = (int64_t)(x % m);
    while (nr) {
        int64_t q = r / nr, tmp = t - q * nt;
        t = nt, nt = tmp;
        tmp = r - q * nr;
        r = nr, nr = tmp;
    }
    return (uint64_t)(t < 0 ? t + (int64_t)m : t);
}

struct job {
    uint32_t b_lo, b_hi;
    uint64_t tested, hard, failed;
    uint32_t fail_a[64], fail_b[64];
    int fail_c[64];
    uint32_t *hard_a, *hard_b;
    size_t hard_cap, hard_n;
};

static void test_pair(struct job *j, uint32_t A, uint32_t B)
{
    const float a = (float)A, b = (float)B;
    volatile float want = a / b; /* IEEE division: correctly rounded */
    const float rn = 1.0f / b;
    j->hard++;
    if (j->hard_n < j->hard_cap) {
        j->hard_a[j->hard_n] = A;
        j->hard_b[j->hard_n] = B;
        j->hard_n++;
    }
    for (int c = g_rcp_dev ? g_rcp_dev[B - (1u << 23)] : -2; c <= (g_rcp_dev ? g_rcp_dev[B - (1u << 23)] : 2); c++) {
        const float got = quotient(a, b, step_ulps(rn, c), g_steps);
        j->tested++;
        if (got != want) {
            if (j->failed < 64) {
                j->fail_a[j->failed] = A, j->fail_b[j->failed] = B, j->fail_c[j->failed] = c;
            }
            j->failed++;
        }
    }
}

static void *worker(void *arg)
{
    struct job *j = arg;
    for (uint32_t B = j->b_lo; B < j->b_hi; B++) {
        for (int k = 24; k <= 25; k++) {
            /* A 2^k - t B = rho */
            const uint32_t lo = k == 24 ? B : (1u << 23), hi = k == 24 ? (1u << 24) : B; /* A in [lo, hi) */
            if (lo >= hi) {
                continue;
            }
            int v = __builtin_ctz(B);
            if (v > k) {
                v = k;
            }
            const uint64_t g = 1ull << v, Bp = B >> v;
            if (Bp == 1) {
                continue; /* B a power of two: every quotient is exact */
            }
            const uint64_t inv = inv_mod((1ull << (k - v)) % Bp, Bp);
            for (int rho = -RHO_MAX; rho <= RHO_MAX; rho++) {
                if (0 == rho || (rho % (int)g) != 0) {
                    continue;
                }
                const int64_t rg = rho / (int64_t)g;
                const uint64_t r = (uint64_t)(((rg % (int64_t)Bp) + (int64_t)Bp) % (int64_t)Bp);
                const uint64_t A0 = (unsigned __int128)r * inv % Bp;
                for (uint64_t A = A0; A < hi; A += Bp) {
                    if (A < lo) {
                        continue;
                    }
                    const int64_t num = (int64_t)(A << k) - rho;
                    if (num % (int64_t)B != 0) {
                        continue;
                    }
                    const int64_t t = num / (int64_t)B;
                    if (!(t & 1)) {
                        continue; /* next to a representable number, not to a midpoint */
                    }
                    test_pair(j, (uint32_t)A, B);
                }
            }
        }
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc > 1) {
        g_steps = atoi(argv[1]);
    }
    if (argc > 2) {
        g_threads = atoi(argv[2]);
    }
    const char *dump = argc > 3 && argv[3][0] != '-' ? argv[3] : NULL;
    if (argc > 4 && 0 == strcmp(argv[4], "rn")) {
        g_rcp_dev = calloc(1u << 23, 1); /* the correctly rounded reciprocal everywhere: Markstein's premise */
    } else if (argc > 4) {
        FILE *t = fopen(argv[4], "rb");
        int8_t *tab = malloc(1u << 23);
        if (!t || fread(tab, 1, 1u << 23, t) != (1u << 23)) {
            fprintf(stderr, "cannot read the reciprocal table %s\n", argv[4]);
            return 2;
        }
        fclose(t);
        g_rcp_dev = tab;
    }
    struct job *jobs = calloc((size_t)g_threads, sizeof(*jobs));
    pthread_t *th = calloc((size_t)g_threads, sizeof(*th));
    const uint32_t span = (1u << 23) / (uint32_t)g_threads;
    for (int i = 0; i < g_threads; i++) {
        jobs[i].b_lo = (1u << 23) + (uint32_t)i * span;
        jobs[i].b_hi = i == g_threads - 1 ? (1u << 24) : jobs[i].b_lo + span;
        jobs[i].hard_cap = dump ? (1u << 16) : 0;
        jobs[i].hard_a = malloc(4 * (jobs[i].hard_cap + 1));
        jobs[i].hard_b = malloc(4 * (jobs[i].hard_cap + 1));
        pthread_create(&th[i], NULL, worker, &jobs[i]);
    }
    uint64_t tested = 0, hard = 0, failed = 0;
    FILE *f = dump ? fopen(dump, "wb") : NULL;
    for (int i = 0; i < g_threads; i++) {
        pthread_join(th[i], NULL);
        tested += jobs[i].tested, hard += jobs[i].hard, failed += jobs[i].failed;
        for (uint64_t k = 0; k < jobs[i].failed && k < 64; k++) {
            printf("FAIL a=%u b=%u r0=RN(1/b)%+d ulp\n", jobs[i].fail_a[k], jobs[i].fail_b[k], jobs[i].fail_c[k]);
        }
        if (f) {
            for (size_t k = 0; k < jobs[i].hard_n; k++) {
                fwrite(&jobs[i].hard_a[k], 4, 1, f);
                fwrite(&jobs[i].hard_b[k], 4, 1, f);
            }
        }
    }
    if (f) {
        fclose(f);
    }
    /* sanity: random pairs, all five reciprocal candidates */
    uint64_t rnd_bad = 0, x = 88172645463325252ull;
    for (int i = 0; i < 50000000; i++) {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        uint32_t A = (1u << 23) | (uint32_t)(x & 0x7fffff), B = (1u << 23) | (uint32_t)((x >> 24) & 0x7fffff);
        if (A > B) {
            uint32_t t = A; A = B; B = t;
        }
        const float a = (float)A * ((x >> 60) & 1 ? 1.0f : 0.5f), b = (float)B;
        volatile float want = (a <= b ? a : b) / (a <= b ? b : a);
        const float lo_ = a <= b ? a : b, hi_ = a <= b ? b : a;
        uint32_t hb;
        memcpy(&hb, &hi_, 4);
        const int cdev = g_rcp_dev ? g_rcp_dev[hb & 0x7fffff] : (int)((x >> 50) % 5) - 2;
        const float got = quotient(lo_, hi_, step_ulps(1.0f / hi_, cdev), g_steps);
        rnd_bad += got != want;
    }
    printf("residual steps %d: %llu near-midpoint pairs (|A 2^k - t B| <= %d, t odd) over all 2^23 divisors, %llu sequences run "
           "(r0 = %s), %llu wrong; 5e7 random pairs: %llu wrong\n", g_steps, (unsigned long long)hard, RHO_MAX,
           (unsigned long long)tested, g_rcp_dev ? "the device's v_rcp_f32, from its table" : "RN(1/b) -2..+2 ulp",
           (unsigned long long)failed, (unsigned long long)rnd_bad);
    return failed || rnd_bad ? 1 : 0;
}
