// How many VALU instructions hide behind v_mfma_f32_16x16x4_f32 (8 passes, 32 clk) on gfx950: same-wave streams of
// 4 x (MFMA + k fillers) per trip, W waves per SIMD all running the same stream; wall time per trip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_shadow_f32.hip -o tools/ubench_shadow_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

#define MF(acc) "v_mfma_f32_16x16x4_f32 %" #acc ", %12, %13, %" #acc "\n\t"
#define A(r) "v_add_u32_e32 %" #r ", %14, %" #r "\n\t"
#define M(r) "v_fma_f32 %" #r ", %14, %15, %" #r "\n\t"
#define C(r) "v_cndmask_b32_e32 %" #r ", %15, %" #r ", vcc\n\t"
#define P(r) "v_max_f32_e32 %" #r ", %14, %" #r "\n\t"

#define KERNEL(NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(1024) void NAME(int iters, float *out)                                      \
    {                                                                                                        \
        float a = (float)threadIdx.x, b = 1.0f + blockIdx.x;                                                 \
        v4f c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;                                                  \
        float r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;                      \
        float k = 3.0f, k2 = 0.5f;                                                                           \
        for (int i = 0; i < iters; i++) {                                                                    \
            asm volatile(BODY                                                                                \
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3),   \
                           "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                                            \
                         : "v"(a), "v"(b), "v"(k), "v"(k2)                                                   \
                         : "vcc");                                                                           \
        }                                                                                                    \
        out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; \
    }

#define F1(X) X(4)
#define F2(X) X(4) X(5)
#define F3(X) X(4) X(5) X(6)
#define F4(X) X(4) X(5) X(6) X(7)
#define F5(X) X(4) X(5) X(6) X(7) X(8)
#define F6(X) X(4) X(5) X(6) X(7) X(8) X(9)
#define F8(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
#define TRIP(F) MF(0) F MF(1) F MF(2) F MF(3) F

KERNEL(k_m4, TRIP())
KERNEL(k_a1, TRIP(F1(A)))
KERNEL(k_a2, TRIP(F2(A)))
KERNEL(k_a3, TRIP(F3(A)))
KERNEL(k_a4, TRIP(F4(A)))
KERNEL(k_a5, TRIP(F5(A)))
KERNEL(k_a6, TRIP(F6(A)))
KERNEL(k_a8, TRIP(F8(A)))
KERNEL(k_m2, TRIP(F2(M)))
KERNEL(k_m4f, TRIP(F4(M)))
KERNEL(k_m6, TRIP(F6(M)))
KERNEL(k_c2, TRIP(F2(C)))
KERNEL(k_c4, TRIP(F4(C)))
KERNEL(k_p2, TRIP(F2(P)))
KERNEL(k_p4, TRIP(F4(P)))
KERNEL(k_a32, F8(A) F8(A) F8(A) F8(A))

typedef void (*kfn_t)(int, float *);
static float *d_out;

// ---- two roles on one SIMD: waves with (wave / 4) even run 4 MFMA per trip, the others 16 fillers: do a matrix-only
//      wave and a VALU-only wave overlap (time = max) or add up (time = sum)?  role 1 both, 2 MFMA waves only, 3 VALU only ----
#define ROLES(NAME, FILL)                                                                                    \
    __global__ __launch_bounds__(1024) void NAME(int iters, int role, float *out)                            \
    {                                                                                                        \
        const int wave = threadIdx.x >> 6;                                                                   \
        float a = (float)threadIdx.x, b = 1.0f + blockIdx.x;                                                 \
        v4f c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;                                                  \
        float r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;                      \
        float k = 3.0f, k2 = 0.5f;                                                                           \
        const bool mf = ((wave >> 2) & 1) == 0;                                                              \
        for (int i = 0; i < iters; i++) {                                                                    \
            if (mf) {                                                                                        \
                if (role != 3) {                                                                             \
                    asm volatile(MF(0) MF(1) MF(2) MF(3)                                                     \
                                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), \
                                   "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                                    \
                                 : "v"(a), "v"(b), "v"(k), "v"(k2) : "vcc");                                 \
                }                                                                                            \
            } else if (role != 2) {                                                                          \
                asm volatile(F8(FILL) F8(FILL)                                                               \
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3),     \
                               "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                                        \
                             : "v"(a), "v"(b), "v"(k), "v"(k2) : "vcc");                                     \
            }                                                                                                \
        }                                                                                                    \
        out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; \
    }
ROLES(k_roles_add, A)
ROLES(k_roles_fma, M)
ROLES(k_roles_cnd, C)
typedef void (*rfn_t)(int, int, float *);
static void run_roles(const char *name, rfn_t fn, int wps, int role, const char *what)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 10000;
    hipLaunchKernelGGL(fn, dim3(256), dim3(256 * wps), 0, 0, iters, role, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(256), dim3(256 * wps), 0, 0, iters, role, d_out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s waves/SIMD=%d role=%d: %7.2f ns per trip | %s\n", name, wps, role, ms * 1e6 / iters, what);
}

static void run(const char *name, kfn_t fn, int wps, const char *what)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 10000;
    hipLaunchKernelGGL(fn, dim3(256), dim3(256 * wps), 0, 0, iters, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(256), dim3(256 * wps), 0, 0, iters, d_out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-8s waves/SIMD=%d: %7.2f ns per trip of all waves of a SIMD | %s\n", name, wps, ms * 1e6 / iters, what);
}

int main()
{
    (void)hipMalloc(&d_out, 256 * 1024 * 4);
    struct { const char *n; kfn_t f; const char *w; } t[] = {
        { "m4", k_m4, "4 MFMA" }, { "a1", k_a1, "4 x (MFMA, 1 v_add_u32)" }, { "a2", k_a2, "4 x (MFMA, 2 v_add_u32)" },
        { "a3", k_a3, "4 x (MFMA, 3 v_add_u32)" }, { "a4", k_a4, "4 x (MFMA, 4 v_add_u32)" }, { "a5", k_a5, "4 x (MFMA, 5 v_add_u32)" },
        { "a6", k_a6, "4 x (MFMA, 6 v_add_u32)" }, { "a8", k_a8, "4 x (MFMA, 8 v_add_u32)" }, { "fma2", k_m2, "4 x (MFMA, 2 v_fma_f32)" },
        { "fma4", k_m4f, "4 x (MFMA, 4 v_fma_f32)" }, { "fma6", k_m6, "4 x (MFMA, 6 v_fma_f32)" }, { "cnd2", k_c2, "4 x (MFMA, 2 v_cndmask)" },
        { "cnd4", k_c4, "4 x (MFMA, 4 v_cndmask)" }, { "max2", k_p2, "4 x (MFMA, 2 v_max_f32)" }, { "max4", k_p4, "4 x (MFMA, 4 v_max_f32)" },
        { "add32", k_a32, "32 v_add_u32 alone" },
    };
    for (int wps : { 4, 2, 1 }) {
        for (auto &x : t) {
            run(x.n, x.f, wps, x.w);
        }
    }
    for (int wps : { 2, 4 }) {
        run_roles("roles add", k_roles_add, wps, 2, "MFMA waves only (half of the waves, 4 MFMA per trip)");
        run_roles("roles add", k_roles_add, wps, 3, "v_add waves only (the other half, 16 per trip)");
        run_roles("roles add", k_roles_add, wps, 1, "both");
        run_roles("roles fma", k_roles_fma, wps, 3, "v_fma_f32 waves only (16 per trip)");
        run_roles("roles fma", k_roles_fma, wps, 1, "both");
        run_roles("roles cnd", k_roles_cnd, wps, 3, "v_cndmask waves only (16 per trip)");
        run_roles("roles cnd", k_roles_cnd, wps, 1, "both");
    }
    return 0;
}
