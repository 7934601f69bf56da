// Issue cost of individual gfx950 VALU / LDS instructions with every SIMD holding W waves that all run the same stream
// of 32 independent instructions per loop trip (inline asm, eight rotating destination registers).  Printed: wall
// ns per instruction per SIMD and the same in cycles of the add_u32 reference (assumed 2 cycles: SIMD-32, wave64).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_ops.hip -o tools/ubench_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY32(X) REP8(X) REP8(X) REP8(X) REP8(X)

// operands: %0-%7 = r0..r7 (32-bit, "+v"), %8 = k (v), %9 = k2 (v), %10 = 64-bit pair p (v), %11 = lds address (v)
#define KERNEL32(NAME, X)                                                                                    \
    __global__ __launch_bounds__(1024) void NAME(int iters, int *out)                                        \
    {                                                                                                        \
        __shared__ int lds[4096];                                                                            \
        lds[threadIdx.x] = threadIdx.x;                                                                      \
        __syncthreads();                                                                                     \
        int r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7, k = 3, k2 = 0x01020304; \
        double p = 1.5;                                                                                      \
        unsigned la = (threadIdx.x & 63) * 16;                                                               \
        for (int i = 0; i < iters; i++) {                                                                    \
            asm volatile(BODY32(X) "s_waitcnt lgkmcnt(0)\n\t"                                                \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)    \
                         : "v"(k), "v"(k2), "v"(p), "v"(la)                                                  \
                         : "vcc", "s10", "s11", "memory");                                                                 \
        }                                                                                                    \
        out[blockIdx.x * 1024 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + lds[5];              \
    }

// 64-bit destination variants: %0-%7 = 64-bit pairs
#define KERNEL64(NAME, X)                                                                                    \
    __global__ __launch_bounds__(1024) void NAME(int iters, int *out)                                        \
    {                                                                                                        \
        __shared__ int lds[4096];                                                                            \
        lds[threadIdx.x] = threadIdx.x;                                                                      \
        __syncthreads();                                                                                     \
        double r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7, p = 1.5;            \
        int k = 3, k2 = 5;                                                                                   \
        unsigned la = (threadIdx.x & 63) * 16;                                                               \
        for (int i = 0; i < iters; i++) {                                                                    \
            asm volatile(BODY32(X) "s_waitcnt lgkmcnt(0)\n\t"                                                \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)    \
                         : "v"(k), "v"(k2), "v"(p), "v"(la)                                                  \
                         : "vcc", "s10", "s11", "memory");                                                                 \
        }                                                                                                    \
        out[blockIdx.x * 1024 + threadIdx.x] = (int)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7) + lds[5];      \
    }

#define I1(name, txt)                                                                                        \
    KERNEL32(k_##name, X_##name)
// --- 32-bit encodings (VOP1 / VOP2 / VOPC) ---
#define X_add_u32(r) "v_add_u32_e32 %" #r ", %8, %" #r "\n\t"
#define X_sub_u32(r) "v_sub_u32_e32 %" #r ", %8, %" #r "\n\t"
#define X_lshlrev(r) "v_lshlrev_b32_e32 %" #r ", 3, %" #r "\n\t"
#define X_lshrrev(r) "v_lshrrev_b32_e32 %" #r ", 3, %" #r "\n\t"
#define X_ashrrev(r) "v_ashrrev_i32_e32 %" #r ", 3, %" #r "\n\t"
#define X_and(r) "v_and_b32_e32 %" #r ", %8, %" #r "\n\t"
#define X_xor(r) "v_xor_b32_e32 %" #r ", %8, %" #r "\n\t"
#define X_mov(r) "v_mov_b32_e32 %" #r ", %8\n\t"
#define X_mul_f32(r) "v_mul_f32_e32 %" #r ", %8, %" #r "\n\t"
#define X_add_f32(r) "v_add_f32_e32 %" #r ", %8, %" #r "\n\t"
#define X_fmac_f32(r) "v_fmac_f32_e32 %" #r ", %8, %9\n\t"
#define X_max_f32(r) "v_max_f32_e32 %" #r ", %8, %" #r "\n\t"
#define X_cndmask(r) "v_cndmask_b32_e32 %" #r ", %8, %" #r ", vcc\n\t"
#define X_cvt_f32_i32(r) "v_cvt_f32_i32_e32 %" #r ", %" #r "\n\t"
#define X_cvt_i32_f32(r) "v_cvt_i32_f32_e32 %" #r ", %" #r "\n\t"
#define X_cvt_f32_u32(r) "v_cvt_f32_u32_e32 %" #r ", %" #r "\n\t"
#define X_cvt_f32_ubyte0(r) "v_cvt_f32_ubyte0_e32 %" #r ", %" #r "\n\t"
#define X_fract(r) "v_fract_f32_e32 %" #r ", %" #r "\n\t"
#define X_floor(r) "v_floor_f32_e32 %" #r ", %" #r "\n\t"
#define X_rcp(r) "v_rcp_f32_e32 %" #r ", %" #r "\n\t"
#define X_mul_i24(r) "v_mul_i32_i24_e32 %" #r ", %8, %" #r "\n\t"
#define X_max_i32(r) "v_max_i32_e32 %" #r ", %8, %" #r "\n\t"
#define X_cmp_e32(r) "v_cmp_gt_f32_e32 vcc, %8, %" #r "\n\t"
#define X_cmp_i_e32(r) "v_cmp_gt_i32_e32 vcc, %8, %" #r "\n\t"
#define X_dot2c(r) "v_dot2c_i32_i16_e32 %" #r ", %8, %9\n\t"
#define X_dot4c(r) "v_dot4c_i32_i8_e32 %" #r ", %8, %9\n\t"
#define X_fmaak(r) "v_fmaak_f32 %" #r ", %8, %" #r ", 0x3f800000\n\t"
#define X_add_lit(r) "v_add_u32_e32 %" #r ", 0x12345678, %" #r "\n\t"
// --- 64-bit encodings (VOP3 / VOP3P / SDWA / DPP) ---
#define X_add_u32_e64(r) "v_add_u32_e64 %" #r ", %8, %" #r "\n\t"
#define X_fma_f32(r) "v_fma_f32 %" #r ", %8, %9, %" #r "\n\t"
#define X_mul_f32_e64(r) "v_mul_f32_e64 %" #r ", %8, %" #r "\n\t"
#define X_max_abs(r) "v_max_f32_e64 %" #r ", |%8|, |%" #r "|\n\t"
#define X_mad_i24(r) "v_mad_i32_i24 %" #r ", %8, %9, %" #r "\n\t"
#define X_mad_i16(r) "v_mad_i32_i16 %" #r ", %8, %9, %" #r "\n\t"
#define X_lshl_add(r) "v_lshl_add_u32 %" #r ", %" #r ", 8, %8\n\t"
#define X_add3(r) "v_add3_u32 %" #r ", %" #r ", %8, %9\n\t"
#define X_and_or(r) "v_and_or_b32 %" #r ", %" #r ", %8, %9\n\t"
#define X_bfi(r) "v_bfi_b32 %" #r ", %8, %9, %" #r "\n\t"
#define X_perm(r) "v_perm_b32 %" #r ", %" #r ", %8, %9\n\t"
#define X_alignbit(r) "v_alignbit_b32 %" #r ", %" #r ", %8, 14\n\t"
#define X_bfe_i32(r) "v_bfe_i32 %" #r ", %" #r ", 14, 16\n\t"
#define X_med3(r) "v_med3_f32 %" #r ", %" #r ", %8, %9\n\t"
#define X_mul_lo(r) "v_mul_lo_u32 %" #r ", %" #r ", %8\n\t"
#define X_mul_hi(r) "v_mul_hi_u32 %" #r ", %" #r ", %8\n\t"
#define X_cndmask_e64(r) "v_cndmask_b32_e64 %" #r ", %8, %" #r ", s[10:11]\n\t"
#define X_cmp_e64(r) "v_cmp_gt_f32_e64 s[10:11], %8, %" #r "\n\t"
#define X_dot2(r) "v_dot2_i32_i16 %" #r ", %8, %9, %" #r "\n\t"
#define X_dot4(r) "v_dot4_i32_i8 %" #r ", %8, %9, %" #r "\n\t"
#define X_pk_add_i16(r) "v_pk_add_i16 %" #r ", %8, %" #r "\n\t"
#define X_pk_mul_lo_u16(r) "v_pk_mul_lo_u16 %" #r ", %8, %" #r "\n\t"
#define X_pk_mad_i16(r) "v_pk_mad_i16 %" #r ", %8, %9, %" #r "\n\t"
#define X_pk_ashr_i16(r) "v_pk_ashrrev_i16 %" #r ", 3, %" #r "\n\t"
#define X_pk_fma_f16(r) "v_pk_fma_f16 %" #r ", %8, %9, %" #r "\n\t"
#define X_cvt_pk_i16(r) "v_cvt_pk_i16_i32 %" #r ", %8, %" #r "\n\t"
#define X_pack_b32(r) "v_pack_b32_f16 %" #r ", %8, %" #r " op_sel:[1,1,0]\n\t"
#define X_sdwa_shift(r) "v_lshrrev_b32_sdwa %" #r ", 14, %" #r " dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
#define X_sdwa_add(r) "v_add_u32_sdwa %" #r ", %8, %" #r " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
#define X_dpp_mov(r) "v_mov_b32_dpp %" #r ", %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define X_dpp_add(r) "v_add_u32_dpp %" #r ", %8, %" #r " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define X_permlane16(r) "v_permlane16_swap_b32 %" #r ", %8\n\t"
#define X_permlane32(r) "v_permlane32_swap_b32 %" #r ", %8\n\t"
// --- LDS ---
#define X_ds_b32(r) "ds_read_b32 %" #r ", %11\n\t"
#define X_ds_bperm(r) "ds_bpermute_b32 %" #r ", %11, %8\n\t"
#define X_ds_swz(r) "ds_swizzle_b32 %" #r ", %8 offset:swizzle(BROADCAST,32,5)\n\t"
// --- 64-bit destinations ---
#define X_pk_fma_f32(r) "v_pk_fma_f32 %" #r ", %10, %10, %" #r "\n\t"
#define X_pk_mul_f32(r) "v_pk_mul_f32 %" #r ", %10, %" #r "\n\t"
#define X_pk_add_f32(r) "v_pk_add_f32 %" #r ", %10, %" #r "\n\t"
#define X_pk_mov(r) "v_pk_mov_b32 %" #r ", %10, %" #r "\n\t"
#define X_ds_b64(r) "ds_read_b64 %" #r ", %11\n\t"
#define X_lshl_add_u64(r) "v_lshl_add_u64 %" #r ", %" #r ", 3, %10\n\t"

#define LIST32(F)                                                                                            \
    F(add_u32) F(sub_u32) F(lshlrev) F(lshrrev) F(ashrrev) F(and) F(xor) F(mov) F(mul_f32) F(add_f32) F(fmac_f32)      \
    F(max_f32) F(cndmask) F(cvt_f32_i32) F(cvt_i32_f32) F(cvt_f32_u32) F(cvt_f32_ubyte0) F(fract) F(floor) F(rcp)      \
    F(mul_i24) F(max_i32) F(cmp_e32) F(cmp_i_e32) F(dot2c) F(dot4c) F(fmaak) F(add_lit) F(add_u32_e64) F(fma_f32)     \
    F(mul_f32_e64) F(max_abs) F(mad_i24) F(mad_i16) F(lshl_add) F(add3) F(and_or) F(bfi) F(perm) F(alignbit)          \
    F(bfe_i32) F(med3) F(mul_lo) F(mul_hi) F(cndmask_e64) F(cmp_e64) F(dot2) F(dot4) F(pk_add_i16) F(pk_mul_lo_u16)   \
    F(pk_mad_i16) F(pk_ashr_i16) F(pk_fma_f16) F(cvt_pk_i16) F(pack_b32) F(sdwa_shift) F(sdwa_add) F(dpp_mov)         \
    F(dpp_add) F(permlane16) F(permlane32) F(ds_b32) F(ds_bperm) F(ds_swz)
#define LIST64(F) F(pk_fma_f32) F(pk_mul_f32) F(pk_add_f32) F(pk_mov) F(ds_b64) F(lshl_add_u64)

#define DEF32(n) KERNEL32(k_##n, X_##n)
#define DEF64(n) KERNEL64(k_##n, X_##n)
LIST32(DEF32)
LIST64(DEF64)

typedef void (*kfn_t)(int, int *);
struct T {
    const char *name;
    kfn_t fn;
};
#define ENT(n) { #n, k_##n },
static T tests[] = { LIST32(ENT) LIST64(ENT) };

int main()
{
    int *d_out;
    (void)hipMalloc(&d_out, 256 * 1024 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps : { 4, 2 }) {
        double ref = 0;
        for (auto &t : tests) {
            hipLaunchKernelGGL(t.fn, dim3(256), dim3(256 * wps), 0, 0, iters, d_out);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(t.fn, dim3(256), dim3(256 * wps), 0, 0, iters, d_out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double ns_per = ms * 1e6 / iters / 32.0 / wps; /* per wave-instruction on one SIMD */
            if (ref == 0) {
                ref = ns_per;
            }
            printf("waves/SIMD=%d %-16s %6.3f ns per wave-instruction  = %5.2f cycles (v_add_u32 := 2)\n", wps, t.name, ns_per,
                   2.0 * ns_per / ref);
        }
    }
    return 0;
}
