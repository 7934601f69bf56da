/*
 * mfm_group_seq.h - the ORDER of operations of a device group's push and fetch (mfm_group.hip), separated from HIP and
 * RCCL so that a CPU test can run it over shards that refuse or fail at chosen steps
 * (tests/hoststub/group_seq_test.cpp).  Plain C++, no device types.
 *
 * The reference hands a sample_buf to every channel thread or to none (multifm/receiver.c:78-98 walks the whole list
 * before it returns).  A device group has to keep that property across GPUs: its shards move in lock step, block by
 * block, and a block that one shard cannot take must not have been taken by another.
 *
 *   push:  0. accepted blocks of another sample format are flushed first, on every shard or on none
 *          1. every shard says what the block would mean for it  (no side effects): must its buffer be launched with it,
 *             may it (a free output slot), does its policy want to (mfm_engine_config::coalesce_samples).  ONE decision for
 *             all shards - launch when one must, or when the root wants to and all may - so that every shard's launches
 *             cover the same samples; a launch that must happen and may not on some shard = MFM_E_BUSY, nothing done
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
 *   int plan(size_t shard, size_t nr_samples, bool *must, bool *may, bool *want);   what nr_samples more would mean (0: what
 *                                                                    is there); no side effects
 *   bool conflict(size_t shard, int format);                         accepted, unlaunched blocks of another format
 *   int unlaunched(size_t shard);                                    samples accepted and not yet launched
 *   int flush(size_t shard);                                         launch them
 *   bool takes_bytes(size_t shard, int format, size_t nr_samples);   8-bit block readable as bytes by this shard's kernel
 *   int acquire(size_t shard, bool raw, int format, void **dst, size_t *cap_samples);
 *   int stage_root(const void *data, size_t nr_samples, int format, bool raw, void **d_root);
 *   int exchange(void *d_root, void *const *dst, size_t bytes);      dst[0] = d_root
 *   int submit(size_t shard, size_t nr_samples, bool launch);
 *   int pending(size_t shard);                                       blocks submitted and not yet released
 *   int fetch(size_t shard, Block *blk);   uint64_t first_output(const Block &);   size_t nr_outputs(const Block &);
 *   void lock(); void unlock();
 *   int fail(int code, const char *what, size_t shard);              records the message, returns code
 */
#ifndef MFM_E_BUSY
#define MFM_E_BUSY (-3)
#endif

/* launch, on every shard or on none, what has been accepted and not launched (a change of sample format, the end of a
 * backlog, a sync) */
template <class Ops, class Flag>
int mfm_group_flush_seq(Ops &ops, Flag *broken)
{
    if (*broken) {
        return ops.fail(MFM_E_DEVICE, "the device group failed in the middle of an earlier block: its shards are out of step", 0);
    }
    const size_t S = ops.shards();
    ops.lock();
    bool any = false, may_all = true;
    int rc = MFM_OK;
    for (size_t i = 0; i < S && rc == MFM_OK; i++) {
        bool must = false, may = true, want = false;
        any = any || ops.unlaunched(i) > 0;
        rc = ops.plan(i, 0, &must, &may, &want);
        may_all = may_all && may;
    }
    if (rc == MFM_OK && any) {
        if (!may_all) {
            rc = ops.fail(MFM_E_BUSY, "an output ring is full: fetch / release, then flush again", 0);
        } else {
            for (size_t i = 0; i < S; i++) {
                rc = ops.flush(i);
                if (rc != MFM_OK) {
                    *broken = i > 0 || S > 1;
                    break;
                }
            }
        }
    }
    ops.unlock();
    return rc;
}

template <class Ops, class Flag>
int mfm_group_push_seq(Ops &ops, Flag *broken, const void *data, size_t nr_samples, int format, bool is_8bit, size_t *bytes_out)
{
    if (*broken) {
        return ops.fail(MFM_E_DEVICE, "the device group failed in the middle of an earlier block: its shards are out of step", 0);
    }
    const size_t S = ops.shards();
    if (S < 1 || S > MFM_GROUP_SEQ_MAX) {
        return ops.fail(MFM_E_INVAL, "shard count", S);
    }
    /* an 8-bit block crosses the links as bytes when every member's kernel can read it so (half the exchange) */
    bool raw = is_8bit;
    for (size_t i = 0; i < S && raw; i++) {
        raw = ops.takes_bytes(i, format, nr_samples);
    }
    bool conflict = false;
    for (size_t i = 0; i < S; i++) {
        conflict = conflict || ops.conflict(i, raw ? format : 0);
    }
    if (conflict) {
        const int rc = mfm_group_flush_seq(ops, broken);
        if (rc != MFM_OK) {
            return rc; /* MFM_E_BUSY: nothing of THIS block has happened */
        }
    }
    bool must = false, may = true, want = false;
    for (size_t i = 0; i < S; i++) {
        bool m = false, y = true, w = false;
        const int rc = ops.plan(i, nr_samples, &m, &y, &w);
        if (rc != MFM_OK) {
            return rc; /* nothing has happened yet */
        }
        must = must || m;
        may = may && y;
        want = i == 0 ? w : want;
    }
    if (must && !may) {
        return ops.fail(MFM_E_BUSY, "an output ring is full", 0); /* nothing has happened yet */
    }
    const bool launch = must || (want && may);
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
        rc = ops.submit(i, nr_samples, launch);
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

template <class Ops, class Flag, class Block>
int mfm_group_fetch_seq(Ops &ops, const Flag *broken, Block *blks)
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
