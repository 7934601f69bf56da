/*
 * mfm_engine_internal.h - engine entry points used by other translation units of libmultifm_hip.so (the device
 * group, mfm_group.hip).  Not part of the public C ABI (include/multifm_hip.h); hidden visibility.
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

struct mfm_engine;

extern "C" {
/* stage one block (MFM_IN_* format) into the engine's next input buffer on its copy stream, without submitting;
 * MFM_E_BUSY when the output ring has no free slot for it.  *d_dst = device address of the staged samples: int16 pairs,
 * unless allow_raw is set and the engine chose to keep an 8-bit block as bytes for its matrix kernel (then the next
 * submit of THIS engine knows; nobody else can use the address). */
#define MFM_STAGE_ALLOW_RAW 1 /* an 8-bit block may stay bytes when the engine's kernel reads them so */
#define MFM_STAGE_PINNED 2    /* `data` is page-locked (mfm_host_alloc): no staging copy, the H2D reads it where it lies; the
                                 push gets a ticket (mfm_engine_copy_ticket) that says when it has been read */
__attribute__((visibility("hidden"))) int mfm_engine_stage(struct mfm_engine *e, const void *data, size_t nr_samples,
                                                           int format, int flags, void **d_dst);
__attribute__((visibility("hidden"))) uint64_t mfm_engine_copy_ticket(struct mfm_engine *e);
/* 1 when the engine would keep a block of nr_samples 8-bit samples of this format as bytes now */
__attribute__((visibility("hidden"))) int mfm_engine_can_take_bytes(struct mfm_engine *e, int format, size_t nr_samples);
/* the stream mfm_engine_stage() queues its work on (hipStream_t) */
__attribute__((visibility("hidden"))) void *mfm_engine_copy_stream(struct mfm_engine *e);
/* blocks submitted (with outputs) and not yet released */
__attribute__((visibility("hidden"))) int mfm_engine_pending_blocks(struct mfm_engine *e);
/* samples accepted and not yet launched (mfm_engine_config::coalesce_samples) */
__attribute__((visibility("hidden"))) int mfm_engine_pending_samples(struct mfm_engine *e);
/* What accepting a block of nr_samples would mean for this engine, without doing anything: *must - the buffer has to be
 * launched with it (no coalescing, or coalesce_samples gathered); *may - a launch would find a free output slot; *want -
 * the engine's own policy would launch although it does not have to (device idle, or one launch in flight and a quarter
 * of its samples gathered).  A device group asks every shard and decides once for all of them. */
__attribute__((visibility("hidden"))) int mfm_engine_plan(struct mfm_engine *e, size_t nr_samples, int *must, int *may, int *want);
/* 1 when the buffer being filled holds accepted blocks of another format than fmt (they have to be flushed first) */
__attribute__((visibility("hidden"))) int mfm_engine_format_conflict(struct mfm_engine *e, int fmt);
#define MFM_SUBMIT_AUTO 0
#define MFM_SUBMIT_DEFER 1
#define MFM_SUBMIT_LAUNCH 2
/* mfm_engine_submit() with the launch decision made by the caller (a device group, for all its shards alike) */
__attribute__((visibility("hidden"))) int mfm_engine_submit_mode(struct mfm_engine *e, size_t nr_samples, void *producer_stream,
                                                                 int wait_producer, int mode);
}
