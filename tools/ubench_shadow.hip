// What fits in the shadow of v_mfma_i32_16x16x64_i8 on gfx950, and what does not: same-wave streams of 4 x (MFMA + fillers)
// per loop trip with W waves per SIMD all running the same stream; wall time per trip, and the time of the fillers alone.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_shadow.hip -o tools/ubench_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

// operands: %0-%3 accumulators (v4i), %4-%11 r0..r7, %12/%13 ds_read_b128 destinations (v4i), %14 a, %15 b (v4i),
// %16 k, %17 k2, %18 lds address
#define MF(acc) "v_mfma_i32_16x16x64_i8 %" #acc ", %14, %15, %" #acc "\n\t"
#define F_ADD(r) "v_add_u32_e32 %" #r ", %16, %" #r "\n\t"
#define F_FMA(r) "v_fma_f32 %" #r ", %16, %17, %" #r "\n\t"
#define F_LSHLADD(r) "v_lshl_add_u32 %" #r ", %" #r ", 8, %16\n\t"
#define F_SDWA(r) "v_lshrrev_b32_sdwa %" #r ", 14, %" #r " dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
#define F_PERM(r) "v_perm_b32 %" #r ", %" #r ", %16, %17\n\t"
#define F_CVT(r) "v_cvt_f32_i32_e32 %" #r ", %" #r "\n\t"
#define F_RCP(r) "v_rcp_f32_e32 %" #r ", %" #r "\n\t"
#define F_DOT(r) "v_dot2_i32_i16 %" #r ", %16, %17, %" #r "\n\t"
#define F_MAD16(r) "v_mad_i32_i16 %" #r ", %16, %17, %" #r "\n\t"
#define F_MAXF(r) "v_max_f32_e32 %" #r ", %16, %" #r "\n\t"
#define F_CMPCND(r) "v_cmp_gt_f32_e32 vcc, %16, %" #r "\n\tv_cndmask_b32_e32 %" #r ", %17, %" #r ", vcc\n\t"
#define F_CND64(r) "v_cndmask_b32_e64 %" #r ", %17, %" #r ", s[10:11]\n\t"
#define L_B128(d) "ds_read_b128 %" #d ", %18\n\t"
#define L_B128o(d) "ds_read_b128 %" #d ", %18 offset:4096\n\t"
#define WAITL "s_waitcnt lgkmcnt(0)\n\t"

#define KERNEL(NAME, BODY)                                                                                   \
    __global__ __launch_bounds__(1024) void NAME(int iters, int role, int *out)                              \
    {                                                                                                        \
        extern __shared__ int lds[];                                                                         \
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;                                     \
        __syncthreads();                                                                                     \
        v4i a = { (int)threadIdx.x, 2, 3, 4 }, b = { 5, 6, 7, (int)blockIdx.x };                             \
        v4i c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0, d0 = c0, d1 = c0;                                \
        int r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7, k = 3, k2 = 0x00010002; \
        unsigned la = (threadIdx.x & 15) * 224 + ((threadIdx.x >> 4) & 3) * 16;                              \
        (void)role;                                                                                          \
        for (int i = 0; i < iters; i++) {                                                                    \
            asm volatile(BODY WAITL                                                                          \
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3),   \
                           "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(d0), "+v"(d1)                        \
                         : "v"(a), "v"(b), "v"(k), "v"(k2), "v"(la)                                          \
                         : "vcc", "s10", "s11", "memory");                                                   \
        }                                                                                                    \
        out[blockIdx.x * 1024 + threadIdx.x] =                                                               \
            c0[0] + c1[1] + c2[2] + c3[3] + r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + d0[0] + d1[1];           \
    }

// ---- 4 MFMA per trip with fillers ----
KERNEL(k_m4, MF(0) MF(1) MF(2) MF(3))
KERNEL(k_m4_add2, MF(0) F_ADD(4) F_ADD(5) MF(1) F_ADD(6) F_ADD(7) MF(2) F_ADD(8) F_ADD(9) MF(3) F_ADD(10) F_ADD(11))
KERNEL(k_m4_fma2, MF(0) F_FMA(4) F_FMA(5) MF(1) F_FMA(6) F_FMA(7) MF(2) F_FMA(8) F_FMA(9) MF(3) F_FMA(10) F_FMA(11))
KERNEL(k_m4_lsa1, MF(0) F_LSHLADD(4) MF(1) F_LSHLADD(6) MF(2) F_LSHLADD(8) MF(3) F_LSHLADD(10))
KERNEL(k_m4_lsa2, MF(0) F_LSHLADD(4) F_LSHLADD(5) MF(1) F_LSHLADD(6) F_LSHLADD(7) MF(2) F_LSHLADD(8) F_LSHLADD(9) MF(3) F_LSHLADD(10) F_LSHLADD(11))
KERNEL(k_m4_add1lsa1, MF(0) F_ADD(4) F_LSHLADD(5) MF(1) F_ADD(6) F_LSHLADD(7) MF(2) F_ADD(8) F_LSHLADD(9) MF(3) F_ADD(10) F_LSHLADD(11))
KERNEL(k_m4_sdwa2, MF(0) F_SDWA(4) F_SDWA(5) MF(1) F_SDWA(6) F_SDWA(7) MF(2) F_SDWA(8) F_SDWA(9) MF(3) F_SDWA(10) F_SDWA(11))
KERNEL(k_m4_perm2, MF(0) F_PERM(4) F_PERM(5) MF(1) F_PERM(6) F_PERM(7) MF(2) F_PERM(8) F_PERM(9) MF(3) F_PERM(10) F_PERM(11))
KERNEL(k_m4_cvt2, MF(0) F_CVT(4) F_CVT(5) MF(1) F_CVT(6) F_CVT(7) MF(2) F_CVT(8) F_CVT(9) MF(3) F_CVT(10) F_CVT(11))
KERNEL(k_m4_rcp1, MF(0) F_RCP(4) MF(1) F_RCP(6) MF(2) F_RCP(8) MF(3) F_RCP(10))
KERNEL(k_m4_mad2, MF(0) F_MAD16(4) F_MAD16(5) MF(1) F_MAD16(6) F_MAD16(7) MF(2) F_MAD16(8) F_MAD16(9) MF(3) F_MAD16(10) F_MAD16(11))
KERNEL(k_m4_dot1, MF(0) F_DOT(4) MF(1) F_DOT(6) MF(2) F_DOT(8) MF(3) F_DOT(10))
KERNEL(k_m4_lds1, MF(0) L_B128(12) MF(1) L_B128o(13) MF(2) L_B128(12) MF(3) L_B128o(13))
KERNEL(k_m4_lds2, MF(0) L_B128(12) L_B128o(13) MF(1) L_B128(12) L_B128o(13) MF(2) L_B128(12) L_B128o(13) MF(3) L_B128(12) L_B128o(13))
KERNEL(k_m4_add2lds1, MF(0) F_ADD(4) F_ADD(5) L_B128(12) MF(1) F_ADD(6) F_ADD(7) L_B128o(13) MF(2) F_ADD(8) F_ADD(9) L_B128(12) MF(3) F_ADD(10) F_ADD(11) L_B128o(13))
KERNEL(k_m4_add2lds2, MF(0) F_ADD(4) F_ADD(5) L_B128(12) L_B128o(13) MF(1) F_ADD(6) F_ADD(7) L_B128(12) L_B128o(13) MF(2) F_ADD(8) F_ADD(9) L_B128(12) L_B128o(13) MF(3) F_ADD(10) F_ADD(11) L_B128(12) L_B128o(13))
// 8 fillers per trip placed unevenly: 1 after the first three MFMAs, 5 after the last
KERNEL(k_m4_add1115, MF(0) F_ADD(4) MF(1) F_ADD(5) MF(2) F_ADD(6) MF(3) F_ADD(7) F_ADD(8) F_ADD(9) F_ADD(10) F_ADD(11))
// 2 after each + a block of 24 behind the trip (epilogue-like): is the block's cost just its own?
#define B8(F) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11)
KERNEL(k_m4_add2_blk24, MF(0) F_ADD(4) F_ADD(5) MF(1) F_ADD(6) F_ADD(7) MF(2) F_ADD(8) F_ADD(9) MF(3) F_ADD(10) F_ADD(11) B8(F_ADD) B8(F_ADD) B8(F_ADD))
KERNEL(k_m4_blk32, MF(0) MF(1) MF(2) MF(3) B8(F_ADD) B8(F_ADD) B8(F_ADD) B8(F_ADD))
// ---- fillers alone (32 per trip) ----
KERNEL(k_add32, B8(F_ADD) B8(F_ADD) B8(F_ADD) B8(F_ADD))
KERNEL(k_cmpcnd16, B8(F_CMPCND) B8(F_CMPCND))
KERNEL(k_cnd64_32, B8(F_CND64) B8(F_CND64) B8(F_CND64) B8(F_CND64))
KERNEL(k_maxf32, B8(F_MAXF) B8(F_MAXF) B8(F_MAXF) B8(F_MAXF))
KERNEL(k_lds32, L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13) L_B128(12) L_B128o(13))

// ---- two roles: waves with (wave/4) even run 4 MFMA per trip, the others 16 fillers of a kind ----
#define ROLES(NAME, FILL)                                                                                    \
    __global__ __launch_bounds__(1024) void NAME(int iters, int role, int *out)                              \
    {                                                                                                        \
        const int wave = threadIdx.x >> 6;                                                                   \
        v4i a = { (int)threadIdx.x, 2, 3, 4 }, b = { 5, 6, 7, (int)blockIdx.x };                             \
        v4i c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0, d0 = c0, d1 = c0;                                \
        int r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7, k = 3, k2 = 0x00010002; \
        unsigned la = 0;                                                                                     \
        const bool mf = ((wave >> 2) & 1) == 0;                                                              \
        for (int i = 0; i < iters; i++) {                                                                    \
            if (mf) {                                                                                        \
                if (role != 3) {                                                                             \
                    asm volatile(MF(0) MF(1) MF(2) MF(3)                                                     \
                                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), \
                                   "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(d0), "+v"(d1)                \
                                 : "v"(a), "v"(b), "v"(k), "v"(k2), "v"(la));                                \
                }                                                                                            \
            } else if (role != 2) {                                                                          \
                asm volatile(B8(FILL) B8(FILL)                                                               \
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3),     \
                               "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(d0), "+v"(d1)                    \
                             : "v"(a), "v"(b), "v"(k), "v"(k2), "v"(la));                                    \
            }                                                                                                \
        }                                                                                                    \
        out[blockIdx.x * 1024 + threadIdx.x] =                                                               \
            c0[0] + c1[1] + c2[2] + c3[3] + r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + d0[0] + d1[1];           \
    }
ROLES(k_roles_add, F_ADD)
ROLES(k_roles_dot, F_DOT)
ROLES(k_roles_lsa, F_LSHLADD)

typedef void (*kfn_t)(int, int, int *);
static int *d_out;
static double run(const char *name, kfn_t fn, int wps, int role, const char *what)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 10000;
    hipLaunchKernelGGL(fn, dim3(256), dim3(256 * wps), 16384 * 4, 0, iters, role, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(256), dim3(256 * wps), 16384 * 4, 0, iters, role, d_out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / iters;
    printf("%-16s waves/SIMD=%d role=%d: %7.2f ns per trip of all waves of a SIMD | %s\n", name, wps, role, ns, what);
    return ns;
}

int main()
{
    (void)hipMalloc(&d_out, 256 * 1024 * 4);
    struct {
        const char *name;
        kfn_t fn;
        const char *what;
    } tests[] = {
        { "m4", k_m4, "4 MFMA" },
        { "m4+add2", k_m4_add2, "4 x (MFMA, 2 v_add_u32)" },
        { "m4+fma2", k_m4_fma2, "4 x (MFMA, 2 v_fma_f32)" },
        { "m4+lsa1", k_m4_lsa1, "4 x (MFMA, 1 v_lshl_add_u32)" },
        { "m4+lsa2", k_m4_lsa2, "4 x (MFMA, 2 v_lshl_add_u32)" },
        { "m4+add1lsa1", k_m4_add1lsa1, "4 x (MFMA, v_add, v_lshl_add)" },
        { "m4+sdwa2", k_m4_sdwa2, "4 x (MFMA, 2 sdwa shifts)" },
        { "m4+perm2", k_m4_perm2, "4 x (MFMA, 2 v_perm_b32)" },
        { "m4+cvt2", k_m4_cvt2, "4 x (MFMA, 2 v_cvt_f32_i32)" },
        { "m4+rcp1", k_m4_rcp1, "4 x (MFMA, 1 v_rcp_f32)" },
        { "m4+mad2", k_m4_mad2, "4 x (MFMA, 2 v_mad_i32_i16)" },
        { "m4+dot1", k_m4_dot1, "4 x (MFMA, 1 v_dot2_i32_i16)" },
        { "m4+lds1", k_m4_lds1, "4 x (MFMA, 1 ds_read_b128)" },
        { "m4+lds2", k_m4_lds2, "4 x (MFMA, 2 ds_read_b128)" },
        { "m4+add2lds1", k_m4_add2lds1, "4 x (MFMA, 2 v_add, 1 ds_read_b128)" },
        { "m4+add2lds2", k_m4_add2lds2, "4 x (MFMA, 2 v_add, 2 ds_read_b128)" },
        { "m4+add1115", k_m4_add1115, "MFMA add MFMA add MFMA add MFMA 5 x add" },
        { "m4+add2+blk24", k_m4_add2_blk24, "4 x (MFMA, 2 v_add) then 24 v_add" },
        { "m4+blk32", k_m4_blk32, "4 MFMA then 32 v_add" },
        { "add32", k_add32, "32 v_add_u32" },
        { "cmp+cnd x16", k_cmpcnd16, "16 x (v_cmp_gt_f32_e32 vcc; v_cndmask_b32_e32 vcc)" },
        { "cnd_e64 x32", k_cnd64_32, "32 v_cndmask_b32_e64 (sgpr pair)" },
        { "max_f32 x32", k_maxf32, "32 v_max_f32_e32" },
        { "lds b128 x32", k_lds32, "32 ds_read_b128 (B-fragment address pattern, rs 224)" },
    };
    for (int wps : { 4, 3, 2, 1 }) {
        for (auto &t : tests) {
            run(t.name, t.fn, wps, 0, t.what);
        }
    }
    for (int wps : { 2, 4 }) {
        run("roles add", k_roles_add, wps, 2, "MFMA waves only");
        run("roles add", k_roles_add, wps, 3, "v_add waves only (16 per trip)");
        run("roles add", k_roles_add, wps, 1, "both");
        run("roles dot", k_roles_dot, wps, 3, "v_dot2 waves only (16 per trip)");
        run("roles dot", k_roles_dot, wps, 1, "both");
        run("roles lsa", k_roles_lsa, wps, 3, "v_lshl_add waves only (16 per trip)");
        run("roles lsa", k_roles_lsa, wps, 1, "both");
    }
    return 0;
}
