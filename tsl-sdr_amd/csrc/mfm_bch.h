/*
 * mfm_bch.h - table form of the reference's BCH(31,21) decoder (pager/bch_code.c:307-398), shared by the pager
 * stages (mfm_pocsag.hip builds the tables and owns the per-device copies; mfm_flex.hip uses them for the frame
 * information word).
 */
#pragma once

#include <stdint.h>

struct MfmBchTables {
    uint32_t flips[1024];  /* index S1 | S3 << 5 : bits to flip, bit 31 = "uncorrectable" */
    uint16_t syn[4][256];  /* syndrome contribution of each byte of the word */
};

/* bch_code_decode on one word: returns the word after correction, *rc = its return value (0 or 1) */
template <class T>
__host__ __device__ __forceinline__ uint32_t mfm_bch_fix(const T *t, uint32_t w, uint32_t *rc)
{
    const uint32_t s = t->syn[0][w & 255u] ^ t->syn[1][(w >> 8) & 255u] ^ t->syn[2][(w >> 16) & 255u] ^ t->syn[3][w >> 24];
    const uint32_t f = t->flips[s];
    *rc = f >> 31;
    return w ^ (f & 0x7fffffffu);
}

/* the device copy of the tables for `device` (created on first use); MFM_OK or an MFM_E_* code */
extern "C" __attribute__((visibility("hidden"))) int mfm_internal_bch_device_tables(int device, MfmBchTables **out);
/* the host copy (the host side of the FLEX stage corrects message words with it) */
extern "C" __attribute__((visibility("hidden"))) const MfmBchTables *mfm_internal_bch_host_tables(void);
