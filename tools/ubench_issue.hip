// Issue model of one gfx950 SIMD, measured in shader cycles (s_memtime) with the instruction streams fixed by inline
// asm: how long v_mfma_i32_16x16x64_i8 occupies the SIMD, how many ordinary VALU instructions fit beside it (same
// wave, interleaved) and whether another wave's VALU stream runs under it.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o tools/ubench_issue
//
// One workgroup per CU; WPS waves per SIMD = blockDim / 256.  Every test prints cycles per loop trip as seen by wave 0
// of block 0 (s_memtime) and the wall time per trip of the whole launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));

// fillers: independent single-register ops on v[6..13]-style operands (%6..%13)
#define F_ADD(r) "v_add_u32 %" #r ", %" #r ", %14\n\t"
#define F_FMA(r) "v_fma_f32 %" #r ", %" #r ", %15, %15\n\t"
#define F_PKF(r) "v_pk_fma_f32 %" #r ", %" #r ", %16, %16\n\t" /* 64-bit operands */
#define F_DOT(r) "v_dot2_i32_i16 %" #r ", %" #r ", %14, %14\n\t"
#define F_RCP(r) "v_rcp_f32 %" #r ", %" #r "\n\t"
#define F_SDWA(r) "v_lshrrev_b32_sdwa %" #r ", 14, %" #r " dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
#define F_LSHLADD(r) "v_lshl_add_u32 %" #r ", %" #r ", 8, %14\n\t"
#define F_CVT(r) "v_cvt_f32_i32 %" #r ", %" #r "\n\t"
#define F_PERM(r) "v_perm_b32 %" #r ", %" #r ", %14, %14\n\t"

#define OPS "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)

#define KERNEL(NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(1024) void NAME(int iters, int role_split, unsigned long long *cyc, int *out) \
    {                                                                                                        \
        v4i a = { (int)threadIdx.x, 2, 3, 4 }, b = { 5, 6, 7, (int)blockIdx.x };                             \
        v4i c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;                                                  \
        int r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7, k = 3;                 \
        float kf = 1.0001f;                                                                                  \
        double kp = 1.0;                                                                                     \
        (void)role_split;                                                                                    \
        const unsigned long long t0 = __builtin_readcyclecounter();                                          \
        for (int i = 0; i < iters; i++) {                                                                    \
            asm volatile(BODY : OPS, "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6),   \
                         "+v"(r7)                                                                            \
                         : "v"(a), "v"(b), "v"(k), "v"(kf), "v"(kp));                                        \
        }                                                                                                    \
        const unsigned long long t1 = __builtin_readcyclecounter();                                          \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                                                           \
            cyc[0] = t1 - t0;                                                                                \
        }                                                                                                    \
        out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; \
    }

// operand numbering inside BODY: %0-%3 accumulators, %4 r0 ... %11 r7, %12 a, %13 b, %14 k, %15 kf, %16 kp
#define MF(acc) "v_mfma_i32_16x16x64_i8 %" #acc ", %12, %13, %" #acc "\n\t"
// one "unit" = MFMA + n fillers on rotating registers
#define U0(acc) MF(acc)
#define U1(F, acc, p) MF(acc) F(p)
#define U2(F, acc, p, q) MF(acc) F(p) F(q)
#define U3(F, acc, p, q, r) MF(acc) F(p) F(q) F(r)
#define U4(F, acc, p, q, r, s) MF(acc) F(p) F(q) F(r) F(s)
#define U6(F, acc) MF(acc) F(4) F(5) F(6) F(7) F(8) F(9)
#define U8(F, acc) MF(acc) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11)

// trip = 4 MFMA (four accumulators) + 4*n fillers
KERNEL(k_m4, U0(0) U0(1) U0(2) U0(3))
KERNEL(k_m4_add1, U1(F_ADD, 0, 4) U1(F_ADD, 1, 5) U1(F_ADD, 2, 6) U1(F_ADD, 3, 7))
KERNEL(k_m4_add2, U2(F_ADD, 0, 4, 5) U2(F_ADD, 1, 6, 7) U2(F_ADD, 2, 8, 9) U2(F_ADD, 3, 10, 11))
KERNEL(k_m4_add3, U3(F_ADD, 0, 4, 5, 6) U3(F_ADD, 1, 7, 8, 9) U3(F_ADD, 2, 10, 11, 4) U3(F_ADD, 3, 5, 6, 7))
KERNEL(k_m4_add4, U4(F_ADD, 0, 4, 5, 6, 7) U4(F_ADD, 1, 8, 9, 10, 11) U4(F_ADD, 2, 4, 5, 6, 7) U4(F_ADD, 3, 8, 9, 10, 11))
KERNEL(k_m4_add6, U6(F_ADD, 0) U6(F_ADD, 1) U6(F_ADD, 2) U6(F_ADD, 3))
KERNEL(k_m4_add8, U8(F_ADD, 0) U8(F_ADD, 1) U8(F_ADD, 2) U8(F_ADD, 3))
// fillers alone: 4*n per trip
#define V8(F) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11)
KERNEL(k_add16, V8(F_ADD) V8(F_ADD))
KERNEL(k_add32, V8(F_ADD) V8(F_ADD) V8(F_ADD) V8(F_ADD))
KERNEL(k_fma32, V8(F_FMA) V8(F_FMA) V8(F_FMA) V8(F_FMA))
KERNEL(k_dot32, V8(F_DOT) V8(F_DOT) V8(F_DOT) V8(F_DOT))
KERNEL(k_rcp32, V8(F_RCP) V8(F_RCP) V8(F_RCP) V8(F_RCP))
KERNEL(k_sdwa32, V8(F_SDWA) V8(F_SDWA) V8(F_SDWA) V8(F_SDWA))
KERNEL(k_lshladd32, V8(F_LSHLADD) V8(F_LSHLADD) V8(F_LSHLADD) V8(F_LSHLADD))
KERNEL(k_cvt32, V8(F_CVT) V8(F_CVT) V8(F_CVT) V8(F_CVT))
KERNEL(k_perm32, V8(F_PERM) V8(F_PERM) V8(F_PERM) V8(F_PERM))
// other filler kinds beside the MFMAs, 4 per MFMA
KERNEL(k_m4_fma4, U4(F_FMA, 0, 4, 5, 6, 7) U4(F_FMA, 1, 8, 9, 10, 11) U4(F_FMA, 2, 4, 5, 6, 7) U4(F_FMA, 3, 8, 9, 10, 11))
KERNEL(k_m4_dot4, U4(F_DOT, 0, 4, 5, 6, 7) U4(F_DOT, 1, 8, 9, 10, 11) U4(F_DOT, 2, 4, 5, 6, 7) U4(F_DOT, 3, 8, 9, 10, 11))
KERNEL(k_m4_dot2, U2(F_DOT, 0, 4, 5) U2(F_DOT, 1, 6, 7) U2(F_DOT, 2, 8, 9) U2(F_DOT, 3, 10, 11))
KERNEL(k_m4_fma6, U6(F_FMA, 0) U6(F_FMA, 1) U6(F_FMA, 2) U6(F_FMA, 3))

// packed fp32 needs 64-bit registers: separate kernel with its own operands
__global__ __launch_bounds__(1024) void k_pk32(int iters, int, unsigned long long *cyc, int *out)
{
    double p0 = threadIdx.x, p1 = 1, p2 = 2, p3 = 3, p4 = 4, p5 = 5, p6 = 6, p7 = 7, kp = 1.5;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#define PK(r) "v_pk_fma_f32 %" #r ", %" #r ", %8, %8\n\t"
        asm volatile(PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7)
                         PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7)
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                     : "v"(kp));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        cyc[0] = t1 - t0;
    }
    out[blockIdx.x * 1024 + threadIdx.x] = (int)(p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7);
}

// two roles in one workgroup: waves with (wave / 4) even run 4 MFMA per trip, the others 16 v_add per trip (r01's
// ubench_overlap, cycle-counted).  role_split = 1: both; 2: MFMA waves only; 3: VALU waves only.
__global__ __launch_bounds__(1024) void k_roles(int iters, int role, unsigned long long *cyc, int *out)
{
    const int wave = threadIdx.x >> 6;
    v4i a = { (int)threadIdx.x, 2, 3, 4 }, b = { 5, 6, 7, (int)blockIdx.x };
    v4i c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
    int r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7, k = 3;
    const bool mf = ((wave >> 2) & 1) == 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (mf) {
        if (role != 3) {
            for (int i = 0; i < iters; i++) {
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %4, %5, %0\n\tv_mfma_i32_16x16x64_i8 %1, %4, %5, %1\n\t"
                             "v_mfma_i32_16x16x64_i8 %2, %4, %5, %2\n\tv_mfma_i32_16x16x64_i8 %3, %4, %5, %3\n\t"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                             : "v"(a), "v"(b));
            }
        }
    } else if (role != 2) {
        for (int i = 0; i < iters; i++) {
#define AD(r) "v_add_u32 %" #r ", %" #r ", %8\n\t"
            asm volatile(AD(0) AD(1) AD(2) AD(3) AD(4) AD(5) AD(6) AD(7) AD(0) AD(1) AD(2) AD(3) AD(4) AD(5) AD(6) AD(7)
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                         : "v"(k));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) {
        cyc[threadIdx.x ? 1 : 0] = t1 - t0; /* wave 0 = MFMA role, wave 4 = VALU role */
    }
    out[blockIdx.x * 1024 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

typedef void (*kfn_t)(int, int, unsigned long long *, int *);

static void run(const char *name, kfn_t fn, int threads, int role, int iters, unsigned long long *d_cyc, int *d_out,
                const char *what)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipMemset(d_cyc, 0, 16);
    hipLaunchKernelGGL(fn, dim3(256), dim3(threads), 0, 0, iters, role, d_cyc, d_out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(256), dim3(threads), 0, 0, iters, role, d_cyc, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2];
    hipMemcpy(c, d_cyc, 16, hipMemcpyDeviceToHost);
    printf("%-14s waves/SIMD=%d role=%d: %7.1f cyc/trip (wave0) %7.1f (wave4)  wall %6.2f ns/trip  | %s\n", name,
           threads / 256, role, (double)c[0] / iters, (double)c[1] / iters, ms * 1e6 / iters, what);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main()
{
    unsigned long long *d_cyc;
    int *d_out;
    hipMalloc(&d_cyc, 16);
    hipMalloc(&d_out, 256 * 1024 * 4);
    const int iters = 20000;
    struct {
        const char *name;
        kfn_t fn;
        const char *what;
    } tests[] = {
        { "m4", k_m4, "4 MFMA" },
        { "m4+add1", k_m4_add1, "4 x (MFMA + 1 v_add)" },
        { "m4+add2", k_m4_add2, "4 x (MFMA + 2 v_add)" },
        { "m4+add3", k_m4_add3, "4 x (MFMA + 3 v_add)" },
        { "m4+add4", k_m4_add4, "4 x (MFMA + 4 v_add)" },
        { "m4+add6", k_m4_add6, "4 x (MFMA + 6 v_add)" },
        { "m4+add8", k_m4_add8, "4 x (MFMA + 8 v_add)" },
        { "m4+fma4", k_m4_fma4, "4 x (MFMA + 4 v_fma_f32)" },
        { "m4+fma6", k_m4_fma6, "4 x (MFMA + 6 v_fma_f32)" },
        { "m4+dot2", k_m4_dot2, "4 x (MFMA + 2 v_dot2)" },
        { "m4+dot4", k_m4_dot4, "4 x (MFMA + 4 v_dot2)" },
        { "add16", k_add16, "16 v_add" },
        { "add32", k_add32, "32 v_add" },
        { "fma32", k_fma32, "32 v_fma_f32" },
        { "pkfma32", k_pk32, "32 v_pk_fma_f32" },
        { "dot32", k_dot32, "32 v_dot2_i32_i16" },
        { "rcp32", k_rcp32, "32 v_rcp_f32" },
        { "sdwa32", k_sdwa32, "32 v_lshrrev_b32_sdwa" },
        { "lshladd32", k_lshladd32, "32 v_lshl_add_u32" },
        { "cvt32", k_cvt32, "32 v_cvt_f32_i32" },
        { "perm32", k_perm32, "32 v_perm_b32" },
    };
    for (int wps : { 1, 2, 4 }) {
        for (auto &t : tests) {
            run(t.name, t.fn, 256 * wps, 0, iters, d_cyc, d_out, t.what);
        }
    }
    for (int threads : { 512, 1024 }) {
        run("roles", k_roles, threads, 2, iters, d_cyc, d_out, "MFMA-role waves only (4 MFMA/trip)");
        run("roles", k_roles, threads, 3, iters, d_cyc, d_out, "VALU-role waves only (16 v_add/trip)");
        run("roles", k_roles, threads, 1, iters, d_cyc, d_out, "both roles");
    }
    return 0;
}
