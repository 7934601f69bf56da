/*
 * mfm_group_seq.h - the ORDER of operations of a device group's push and fetch (mfm_group.hip), separated from HIP and
 * RCCL so that a CPU test can run it over shards that refuse or fail at chosen steps
 * (tests/hoststub/group_seq_test.cpp).  Plain C++, no device types.
 *
 * The reference hands a sample_buf to every channel thread or to none (multifm/receiver.c:78-98 walks the whole list
 * before it returns).  A device group has to keep that property across GPUs: its shards move in lock step, block by
 * block, and a block that one shard cannot take must not have been taken by another.
 *
 *   push:  1. every shard has room for the block's outputs      (no side effects; MFM_E_BUSY = fetch / release and retry)
 *          2. every non-root shard names its input buffer        (waits for the kernel that last read it; no side effects)
 *          3. the root stages the block                          (H2D into its own input buffer; nothing advances yet)
 *          4. the exchange                                        (RCCL; a failure here leaves every shard where it was)
 *          5. every shard submits, under the group's lock        (from here a failure is fatal: the group is `broken`
 *                                                                  and every later call returns MFM_E_DEVICE)
 *   fetch: under the same lock, "does every shard hold a finished-or-running block" - a consumer thread can therefore
 *          never see shard 0's block without shard 1's (round 2 failed the whole receiver with "shards out of step" when
 *          the drain thread looked between two submits) - then the per-shard waits outside the lock.
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifndef MFM_OK
#define MFM_OK 0
#define MFM_E_INVAL (-1)
#define MFM_E_DEVICE (-4)
#define MFM_E_STATE (-5)
#define MFM_E_DONE (-6)
#endif

#define MFM_GROUP_SEQ_MAX 16

/*
 * Ops (duck typed):
 *   size_t shards();
 *   int room(size_t shard, size_t nr_samples);                       MFM_OK / MFM_E_BUSY
 *   bool takes_bytes(size_t shard, int format, size_t nr_samples);   8-bit block readable as bytes by this shard's kernel
 *   int acquire(size_t shard, bool raw, int format, void **dst, size_t *cap_samples);
 *   int stage_root(const void *data, size_t nr_samples, int format, bool raw, void **d_root);
 *   int exchange(void *d_root, void *const *dst, size_t bytes);      dst[0] = d_root
 *   int submit(size_t shard, size_t nr_samples);
 *   int pending(size_t shard);                                       blocks submitted and not yet released
 *   int fetch(size_t shard, Block *blk);   uint64_t first_output(const Block &);   size_t nr_outputs(const Block &);
 *   void lock(); void unlock();
 *   int fail(int code, const char *what, size_t shard);              records the message, returns code
 */
template <class Ops>
int mfm_group_push_seq(Ops &ops, bool *broken, const void *data, size_t nr_samples, int format, bool is_8bit, size_t *bytes_out)
{
    if (*broken) {
        return ops.fail(MFM_E_DEVICE, "the device group failed in the middle of an earlier block: its shards are out of step", 0);
    }
    const size_t S = ops.shards();
    if (S < 1 || S > MFM_GROUP_SEQ_MAX) {
        return ops.fail(MFM_E_INVAL, "shard count", S);
    }
    for (size_t i = 0; i < S; i++) {
        const int rc = ops.room(i, nr_samples);
        if (rc != MFM_OK) {
            return rc; /* nothing has happened yet */
        }
    }
    /* an 8-bit block crosses the links as bytes when every member's kernel can read it so (half the exchange) */
    bool raw = is_8bit;
    for (size_t i = 0; i < S && raw; i++) {
        raw = ops.takes_bytes(i, format, nr_samples);
    }
    void *dst[MFM_GROUP_SEQ_MAX] = { nullptr };
    for (size_t i = 1; i < S; i++) {
        size_t cap = 0;
        const int rc = ops.acquire(i, raw, format, &dst[i], &cap);
        if (rc != MFM_OK) {
            return rc; /* naming a buffer changes nothing: the shards are where they were */
        }
        if (cap < nr_samples) {
            return ops.fail(MFM_E_INVAL, "a shard's input buffer is too small for the block", i);
        }
    }
    int rc = ops.stage_root(data, nr_samples, format, raw, &dst[0]);
    if (rc != MFM_OK) {
        return rc; /* MFM_E_BUSY cannot come from here any more (step 1); a copy that failed advanced nothing */
    }
    const size_t bytes = nr_samples * (raw ? 2u : 4u);
    rc = ops.exchange(dst[0], dst, bytes);
    if (rc != MFM_OK) {
        *broken = true; /* nothing was submitted, but a collective that failed half-way leaves the communicators unusable */
        return rc;
    }
    ops.lock();
    for (size_t i = 0; i < S; i++) {
        rc = ops.submit(i, nr_samples);
        if (rc != MFM_OK) {
            *broken = i > 0 || S > 1; /* shards before this one have taken the block */
            break;
        }
    }
    ops.unlock();
    if (rc == MFM_OK && bytes_out) {
        *bytes_out = bytes;
    }
    return rc;
}

template <class Ops, class Block>
int mfm_group_fetch_seq(Ops &ops, const bool *broken, Block *blks)
{
    if (*broken) {
        return ops.fail(MFM_E_DEVICE, "the device group failed in the middle of an earlier block: its shards are out of step", 0);
    }
    const size_t S = ops.shards();
    ops.lock();
    size_t with = 0;
    for (size_t i = 0; i < S; i++) {
        with += ops.pending(i) > 0 ? 1u : 0u;
    }
    ops.unlock();
    if (0 == with) {
        return MFM_E_DONE;
    }
    if (with != S) {
        /* under the lock a push is either before its first or behind its last submit */
        return ops.fail(MFM_E_STATE, "shards out of step: not every shard holds a block", with);
    }
    for (size_t i = 0; i < S; i++) {
        const int rc = ops.fetch(i, &blks[i]);
        if (rc != MFM_OK) {
            return rc == MFM_E_DONE ? ops.fail(MFM_E_STATE, "shards out of step: a shard lost its block", i) : rc;
        }
        if (ops.first_output(blks[i]) != ops.first_output(blks[0]) || ops.nr_outputs(blks[i]) != ops.nr_outputs(blks[0])) {
            return ops.fail(MFM_E_STATE, "shards out of step: blocks of different stream positions", i);
        }
    }
    return MFM_OK;
}
