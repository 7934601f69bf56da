/*
 * multifm_hip.h - C ABI of the MI355X multifm channel engine (libmultifm_hip.so).
 *
 * This is the drop-in boundary for multifm's per-channel hot path.  In the reference
 * (pvachon/tsl-sdr) every channel is a pthread that runs
 *
 *     demod_thread_process()            multifm/demod.c:48-121
 *       direct_fir_push_sample_buf()    filter/direct_fir.c:118-146
 *       direct_fir_process()            filter/direct_fir.c:422-453   (complex-tap decimating FIR
 *                                                                      + Q14 derotator)
 *       multifm_fm_demod_process()      multifm/fm_demod.c:36-85      (fast_atan2f discriminator)
 *       write(fifo_fd, pcm)             multifm/demod.c:93
 *
 * on every struct sample_buf that receiver_sample_buf_deliver() (multifm/receiver.c:78-98) hands it.
 * Here ONE engine object owns all channels of a receiver and runs that whole loop as one fused HIP
 * kernel per block of wideband samples.  The entry points below are what a reference-side
 * replacement of demod.c / receiver.c binds:
 *
 *   reference call                                   engine call
 *   ------------------------------------------------ -----------------------------------------
 *   demod_thread_new(.., offset_hz, samp_hz,         mfm_engine_add_channel()
 *       out_fifo, decimation, lpf_taps, nr_taps,
 *       fir_debug_output, gain)   demod.h:104-116
 *   (receiver_start)              receiver.c:268      mfm_engine_commit()
 *   receiver_sample_buf_deliver() receiver.c:78-98    mfm_engine_push() | acquire_input()+submit()
 *   write(fifo_fd, ...)           demod.c:93          mfm_engine_fetch() / mfm_engine_release()
 *   demod_thread_delete()         demod.c:163-190     mfm_engine_destroy()
 *
 * Conventions follow the reference's aresult_t style: every call returns an int, 0 (MFM_OK) on
 * success and a negative MFM_E_* on failure; nothing throws; handles are opaque; plain pointers and
 * sizes only.  All sample data is interleaved int16 I,Q exactly as in struct sample_buf::data_buf
 * (filter/sample_buf.h:59-102).  Output PCM is the same int16 stream a channel thread writes to its
 * FIFO; optional filtered IQ is the signalDebugFile stream (demod.c:75-81).
 *
 * The library has no CPU fallback: without a usable HIP device mfm_engine_create() fails with
 * MFM_E_DEVICE.
 */
#ifndef MULTIFM_HIP_H
#define MULTIFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFM_OK 0
#define MFM_E_INVAL (-1)  /* bad argument (TSL_ASSERT_ARG failures in the reference) */
#define MFM_E_NOMEM (-2)  /* host or device allocation failed */
#define MFM_E_BUSY (-3)   /* no free output slot / input buffer (direct_fir.c:136 A_E_BUSY analogue) */
#define MFM_E_DEVICE (-4) /* HIP runtime error or no device */
#define MFM_E_STATE (-5)  /* call not valid in this state (e.g. add_channel after commit) */
#define MFM_E_DONE (-6)   /* nothing to fetch (A_E_DONE analogue) */

#define MFM_ABI_VERSION 4 /* 2: mfm_resampler_config grew flags + reserved; mfm_flex_*, mfm_group_* added
                             3: mfm_stats grew timed_launches, rot_exact_channels, rot_fast_slices, k_steps, tap_hi_mask, taps_resident; MFM_F_TIMING_SPARSE, MFM_F_STREAM_TAPS;
                                mfm_group_config.exchange
                             4: mfm_engine_config / mfm_group_config grew coalesce_samples (+ a third ext_input); mfm_stats grew submits,
                                pending_samples; mfm_engine_flush, mfm_group_flush, mfm_engine_input_bytes_cfg, mfm_engine_replay,
                                mfm_engine_last_launch_input, mfm_engine_seek, mfm_host_alloc/free, mfm_*_push_pinned,
                                mfm_*_copy_done/_wait, mfm_devtest_discriminate; MFM_F_GATHER, MFM_F_OVERLAP */

/* flags for mfm_engine_config::flags */
#define MFM_F_DEVICE_ONLY 0x1u /* keep outputs in HBM; no host mirror, fetch() unavailable */
#define MFM_F_TIMING 0x2u      /* bracket every kernel launch with HIP events */
#define MFM_F_FORCE_DOT2 0x4u  /* run the v_dot2 (packed int16 VALU) kernel even where the matrix-core kernel applies:
                                  both produce the same bits; parity tests and A/B timing select it here */
#define MFM_F_FORCE_MFMA_V1 0x8u /* where both matrix-core kernels apply, run the first-generation one (31-output column
                                  blocks, 2-byte PCM stores) instead of the second (64-output tiles, 8-byte stores) */

#define MFM_F_TIMING_SPARSE 0x20u /* with MFM_F_TIMING: bracket one launch in four only (a back-to-back stream then runs
                                     without an event pair between most kernels; mfm_stats.timed_launches says how many
                                     durations kernel_ms sums) */
#define MFM_F_GROUP_SHARED_DEVICE 0x40u /* mfm_group_config only, a TEST AID: the same device may be listed several times, so that
                                           a group of several shards runs on a one-GPU box (real RCCL refuses such a communicator;
                                           tests/hoststub/fake_rccl.cpp stands in for it) */
#define MFM_F_STREAM_TAPS 0x80u /* filters of 129..512 taps on the matrix kernel: re-read the taps from L2 in every iteration
                                  (the round-1 form: 128 registers, two workgroups per CU) instead of keeping all of them in
                                  registers (256 registers, one workgroup per CU); same bits; parity tests and A/B timing */
#define MFM_F_GATHER 0x100u    /* with coalesce_samples: launch only once coalesce_samples have gathered, or on mfm_engine_flush() /
                                  mfm_engine_sync() - never because the device happens to be idle (a producer that knows when
                                  its backlog ends and flushes then; deterministic launch boundaries for tests) */
#define MFM_F_OVERLAP 0x200u   /* second-generation kernel: consecutive launches alternate between two compute streams.  A launch
                                  depends on the one before it through input samples only (it recomputes the output in front of
                                  it and finds its rotator position from the stream's output count), so the next launch's
                                  workgroups take the slots the current one's shorter chunks free up instead of waiting for
                                  its last tile and a dispatch.  mfm_engine_stream() then returns the stream of the most
                                  recent launch; per-launch durations (MFM_F_TIMING) include the time a launch waits for
                                  slots and are no longer kernel time.  Ignored by the other kernels. */
#define MFM_F_V3L_ONE_ROW_BLOCK 0x400u /* long filters on the second generation (mfm_kernel_v3l.hip): one row block (8 channels) per
                                  wave - slices of 64 channels - also where two would fit (slices of 128: the default for more
                                  than 64 channels when two row blocks' taps fit 128 registers); same bits; parity tests, A/B timing */
#define MFM_F_SLICE_128 0x800u  /* 128-tap filters (filter/direct_fir.c:363-384 at multifm's 2.4 MS/s -> 25 kS/s geometry) on slices of
                                  128 channels - two row blocks per wave share every B fragment and every staged image
                                  (mfm_kernel_v3l.hip) - whatever the channel count; the default takes them from 512 channels on,
                                  where they measured faster than slices of 64 (by 0.6-1.4 %).  Same bits; parity tests, A/B timing */
#define MFM_F_SLICE_64 0x1000u  /* ... never: slices of 64 channels (mfm_kernel_v3.hip) at any channel count */
#define MFM_F_PCM_WRITE_BACK 0x2000u /* second-generation kernels: PCM stores never go through to memory with system scope (the default does
                                  that for launches of 512 channels and more: less L2-miss traffic there); same bits; A/B timing */
#define MFM_F_WIDEN_8BIT 0x10u /* mfm_engine_push_bytes: always widen 8-bit blocks to int16 in HBM first, also where the
                                  matrix kernel could read the bytes themselves (same bits; parity tests and A/B timing) */

struct mfm_engine_config {
    uint32_t abi_version;       /* MFM_ABI_VERSION */
    int32_t device;             /* HIP device ordinal */
    uint32_t sample_rate_hz;    /* receiver sampleRateHz, receiver.c:138 */
    uint32_t decimation;        /* decimationFactor, receiver.c:160-172 */
    uint32_t max_block_samples; /* largest block one push()/submit() may carry */
    uint32_t flags;             /* MFM_F_* */
    /* Optional caller-owned device memory for the input staging buffers (two; three with coalesce_samples), each at
     * least mfm_engine_input_bytes_cfg() bytes (lets a caller hand in torch/RCCL-registered memory).
     * NULL = the engine allocates. */
    void *ext_input[3];
    /*
     * Backlog coalescing.  A channel thread of the reference takes whatever its work queue holds - up to 128 queued
     * sample_bufs (multifm/demod.c:297) - and runs them back to back (demod.c:134-150): its cost per sample does not depend
     * on the size of the buffers a front end delivers (4096 samples from file_if.c:18, 131072 from rtl_sdr_if.c:46).  A
     * kernel launch has a fixed cost, so the engine does the same with launches: with coalesce_samples > 0 a submitted block
     * is APPENDED to the input buffer being filled, and the buffer is launched as one pass
     *   - at once when the device has nothing to do (a live stream keeps its latency),
     *   - with one launch in flight, as soon as a quarter of that launch's samples have gathered (the device never runs dry),
     *   - otherwise when coalesce_samples have gathered, or on mfm_engine_flush() / mfm_engine_sync().
     * The output stream is the same in every case (it never depended on the blocking: filter/direct_fir.c:328-417 walks
     * sample by sample); mfm_engine_fetch() returns one block per LAUNCH.  0 = every submit is a launch (rounds 1-3).
     */
    uint32_t coalesce_samples;
    uint32_t reserved;          /* 0 */
};

struct mfm_engine; /* opaque */

struct mfm_block {
    uint64_t first_output; /* stream index of pcm[.][0] */
    size_t nr_outputs;     /* outputs per channel in this block */
    size_t stride;         /* elements between consecutive channels */
    const int16_t *pcm;    /* [nr_channels][stride] host memory, valid until release() */
    const int16_t *iq;     /* [nr_channels][2*stride] filtered I,Q or NULL */
};

struct mfm_stats {
    uint64_t samples_in;       /* wideband samples accepted */
    uint64_t outputs;          /* outputs produced per channel */
    uint64_t launches;         /* kernel launches */
    double kernel_ms;          /* sum of the durations of `timed_launches` launches (MFM_F_TIMING), HIP events */
    uint32_t nr_channels;
    uint32_t nr_taps;
    uint32_t outputs_per_tile; /* kernel geometry, informational */
    uint32_t lds_bytes;
    uint32_t grid_last;        /* workgroups of the last launch */
    uint32_t tail_samples;     /* unconsumed samples carried to the next block */
    uint64_t rot_table_entries;
    uint32_t kernel_variant;   /* 0 = v_dot2 kernel, 1 = int8-MFMA (FIR-as-GEMM) kernel, 2 = its second generation */
    uint32_t pending_blocks;   /* finished or in-flight blocks not yet fetched + released */
    uint64_t launches_8bit;    /* of `launches`: those that read an 8-bit block as bytes (mfm_engine_push_bytes) */
    uint64_t timed_launches;   /* launches whose duration is in kernel_ms: all of them with MFM_F_TIMING, one in four
                                  with MFM_F_TIMING_SPARSE as well */
    uint32_t rot_exact_channels; /* channels whose rotator (filter/direct_fir.c:151-172) is exactly +-(16384, 0) for ever:
                                    offsets at multiples of half the output rate; their derotation is the identity or a
                                    sign flip */
    uint32_t rot_fast_slices;  /* 64-channel slices of the second-generation kernel made of such channels only */
    uint32_t k_steps;          /* matrix kernels: k-steps of 64 int16 elements (32 complex taps) per output, padded */
    uint32_t tap_hi_mask;      /* matrix kernels: bit k set = k-step k has taps beyond one byte, so its two products with
                                  the high-byte tap plane are issued (4 matrix instructions for that k-step, else 2) */
    uint32_t taps_resident;    /* first-generation matrix kernel, filters of 129..512 taps: 1 = int16 blocks run an instance that
                                  keeps every k-step of taps in registers, 0 = the taps are streamed from L2 (MFM_F_STREAM_TAPS,
                                  or no resident instance for the geometry) */
    uint32_t slice_channels;   /* matrix kernels: channels whose workgroup shares one staged image: 64, or 128 on the long-filter kernel with
                                  two row blocks per wave (129..512 taps at more than 64 channels; 128 taps from 512 channels on or with
                                  MFM_F_SLICE_128); 0: the v_dot2 kernel */
    uint64_t submits;          /* blocks accepted (mfm_engine_submit / push); with coalesce_samples several of them share a launch */
    uint64_t pending_samples;  /* samples accepted and not yet launched (coalesce_samples; mfm_engine_flush launches them) */
};

/* Size in bytes of one input staging buffer for this configuration and tap count (coalesce_samples = 0). */
size_t mfm_engine_input_bytes(uint32_t max_block_samples, uint32_t nr_taps);
/* The same for a configuration with coalesce_samples; also how many buffers the engine uses (2, or 3 when it coalesces:
 * one being read, one queued behind it, one being filled), i.e. how many ext_input pointers it wants. */
size_t mfm_engine_input_bytes_cfg(const struct mfm_engine_config *cfg, uint32_t nr_taps, uint32_t *nr_buffers);

int mfm_engine_create(struct mfm_engine **pe, const struct mfm_engine_config *cfg);
void mfm_engine_destroy(struct mfm_engine **pe);

/*
 * Register a channel the way demod_thread_new() does (multifm/demod.h:104-116): real low-pass taps
 * are rotated to offset_hz and quantised to Q14 (demod.c:204-269), the derotator increment is
 * derived from offset_hz and the decimation (direct_fir.c:72-79).  want_iq != 0 asks for the
 * filtered-IQ stream too (fir_debug_output).  Returns the channel index (>= 0) or MFM_E_*.
 * All channels of one engine share nr_taps (the reference shares lpfTaps, receiver.c:175-184).
 */
int mfm_engine_add_channel(struct mfm_engine *e, int32_t offset_hz, const double *lpf_taps, size_t nr_taps,
                           double channel_gain, int want_iq);

/* Same, from already-quantised Q14 taps and rotator increment (fixtures, tests). */
int mfm_engine_add_channel_q14(struct mfm_engine *e, const int16_t *coeff_re, const int16_t *coeff_im,
                               size_t nr_taps, int16_t rot_incr_re, int16_t rot_incr_im, int want_iq);

/* Read back what a channel was programmed with (Q14 taps, rotator increment as {re, im}). */
int mfm_engine_get_channel(struct mfm_engine *e, uint32_t chan, int16_t *coeff_re, int16_t *coeff_im,
                           int16_t rot_incr[2]);

/* Freeze the channel set: build tap/rotator tables, allocate device buffers. */
int mfm_engine_commit(struct mfm_engine *e);

/*
 * Zero-copy ingest.  acquire_input() returns device memory where the caller (an H2D copy, an RCCL
 * broadcast, a generator kernel) must write the next block; submit() then processes nr_samples of
 * it.  With wait_producer != 0, producer_stream is the hipStream_t the data was produced on (NULL is
 * the legacy default stream, which is what torch's default stream is): the engine's compute stream
 * waits for the work queued there so far (no host sync).  With wait_producer == 0 the caller
 * guarantees the data is already in place.  Blocks are processed in submit order.
 */
int mfm_engine_acquire_input(struct mfm_engine *e, void **d_dst, size_t *capacity_samples);
int mfm_engine_submit(struct mfm_engine *e, size_t nr_samples, void *producer_stream, int wait_producer);

/* Host ingest: copies nr_samples interleaved int16 IQ pairs (any count <= max_block_samples) and
 * submits them.  Returns as soon as the copy has been staged; never blocks longer than a memcpy. */
int mfm_engine_push(struct mfm_engine *e, const int16_t *iq, size_t nr_samples);

/*
 * Host ingest of 8-bit captures: the byte pairs are staged as they are (half the
 * PCIe bytes of mfm_engine_push).  Where a matrix-core kernel runs and no channel asked for its filtered IQ they stay
 * bytes in HBM and the kernel's GEMM takes them as its one sample plane (same bits as the widened path); otherwise - and
 * for a cu8 block of odd length, or behind a history of another format - they are widened to int16 on the device exactly
 * as the reference's front ends do on the host.  nr_samples IQ pairs = 2 * nr_samples bytes.
 */
#define MFM_IN_CS16 0       /* interleaved int16, same as mfm_engine_push (multifm/file_if.c:46-64) */
#define MFM_IN_CS8 1        /* signed bytes, sign-extended (multifm/file_if.c:66-111) */
#define MFM_IN_CU8 2        /* file_if's "cu8": bytes read as SIGNED, minus 127; after an odd number of samples the
                               last one is stored without the subtraction (multifm/file_if.c:113-157) */
#define MFM_IN_RTLSDR_U8 3  /* unsigned bytes, (b - 127) << 7 (multifm/rtl_sdr_if.c:146-158) */
int mfm_engine_push_bytes(struct mfm_engine *e, const void *bytes, size_t nr_samples, int format);
/*
 * Host ingest without the staging copy.  The reference's sample_bufs come from a fixed pool (frame_alloc_new,
 * multifm/receiver.c:154-157); when that pool is page-locked memory (mfm_host_alloc) the H2D copy reads data_buf where
 * the front end wrote it.  mfm_engine_push_pinned() is mfm_engine_push_bytes() (any MFM_IN_* format) on such memory: it
 * returns at once with a ticket, and the buffer must stay untouched until mfm_engine_copy_done(ticket) says 1 (or
 * mfm_engine_copy_wait returns) - that is when the reference would sample_buf_decref() it (filter/direct_fir.c:395).
 */
void *mfm_host_alloc(size_t bytes); /* page-locked host memory (hipHostMalloc); NULL on failure */
/* The host <-> device link by itself, as a yardstick for host-fed throughput (bench.py `link`): total_bytes leave an arena of
 * page-locked memory in pieces of piece_bytes (one hipMemcpyAsync each on one stream, as the engine's pinned pushes do per
 * sample_buf), while d2h_per_h2d bytes per input byte come back on a second stream (the PCM mirror; 0: none).  Rates in
 * GB/s over the second of two passes. */
int mfm_link_probe(int device, size_t piece_bytes, size_t total_bytes, double d2h_per_h2d, double *h2d_GBps, double *d2h_GBps);
/* the same with the pieces gap_bytes apart in the arena (a sample_buf's header between the data of two frames) and
 * pieces_per_command of them per copy command - one strided hipMemcpy2DAsync that packs them on the device, what
 * mfm_engine_push_pinned_run() issues for a run of adjacent frames */
int mfm_link_probe_runs(int device, size_t piece_bytes, size_t gap_bytes, size_t pieces_per_command, size_t total_bytes,
                        double d2h_per_h2d, double *h2d_GBps, double *d2h_GBps);
void mfm_host_free(void *p);
int mfm_engine_push_pinned(struct mfm_engine *e, const void *data, size_t nr_samples, int format, uint64_t *ticket);
int mfm_engine_copy_done(struct mfm_engine *e, uint64_t ticket); /* 1: read, 0: not yet, < 0: error */
int mfm_engine_copy_wait(struct mfm_engine *e, uint64_t ticket);
/* The same for a producer on the device (a collective, a capture card's DMA): where the next block's BYTES go, two per
 * sample, when the engine's kernel can read them as they are - the second-generation matrix kernel, and no history of
 * another format in front (MFM_E_STATE otherwise: widen the block as the reference does and use
 * mfm_engine_acquire_input).  Then mfm_engine_submit() as for int16 blocks; a cu8 block must be of even length. */
int mfm_engine_acquire_input_bytes(struct mfm_engine *e, int format, void **d_dst, size_t *capacity_samples);

/* Oldest finished block, in submit order (blocks with zero outputs are skipped).  Waits for the
 * device.  MFM_E_DONE when nothing is pending.  The block stays valid until release(). */
int mfm_engine_fetch(struct mfm_engine *e, struct mfm_block *blk);
int mfm_engine_release(struct mfm_engine *e);

/* Device-resident view of the most recent submit's outputs (MFM_F_DEVICE_ONLY users). */
int mfm_engine_last_output_device(struct mfm_engine *e, void **d_pcm, size_t *stride, size_t *nr_outputs,
                                  void **d_iq);

/* What the most recent launch read: device address of its [history tail | blocks] and the sample count (self-checks that
 * re-run a launch's input through a reference; valid until nbuf - 1 further launches have been queued). */
int mfm_engine_last_launch_input(struct mfm_engine *e, void **d_in, size_t *nr_samples, int *format);

/* coalesce_samples: launch what has been accepted and not yet launched (MFM_E_BUSY when every output slot holds an
 * unfetched block: fetch / release and call again).  Producer side: the thread that submits.  No-op otherwise. */
int mfm_engine_flush(struct mfm_engine *e);

/* Wait for everything submitted so far (launches pending samples first, as mfm_engine_flush, and returns its MFM_E_BUSY).
 * Threading: push / stage / acquire_input / submit / flush / sync / seek / reset are PRODUCER-side calls and belong to one
 * thread at a time; fetch / release (and get_stats, copy_done / copy_wait) may run on another.  A flush that finds nothing
 * gathered is safe from any thread (it tests under the engine's lock). */
int mfm_engine_sync(struct mfm_engine *e);

/*
 * A producer loop in C, for measurements: `nr_blocks` times { acquire_input(); submit(block_samples, no producer) } on
 * whatever the input buffers hold (the caller pre-fills them), exactly what a C host does per delivered sample_buf
 * without the per-call cost of a scripting language in between.  Stops at the first error and returns it.
 */
int mfm_engine_replay(struct mfm_engine *e, size_t block_samples, size_t nr_blocks);

/* Forget the stream: history tail, rotator phase and discriminator state go back to a fresh
 * stream (what restarting the reference does). Pending blocks are dropped. */
int mfm_engine_reset(struct mfm_engine *e);

/*
 * Resume a stream that had produced outputs_before outputs per channel (a receiver restarted from a checkpoint): as
 * mfm_engine_reset(), except that the derotators stand where outputs_before steps of their recurrence leave them
 * (filter/direct_fir.c:151-172 carries rot_phase across buffers; the recurrence is input independent) and
 * mfm_block::first_output goes on counting from there.  The filter history and the discriminator's last sample start
 * empty, as after a restart of the reference.
 */
int mfm_engine_seek(struct mfm_engine *e, uint64_t outputs_before);

int mfm_engine_get_stats(struct mfm_engine *e, struct mfm_stats *st);

/* MFM_F_TIMING: durations (ms, HIP events on the compute stream) of the most recent launches, oldest first; at most
 * `cap` and at most the last 4096.  Returns how many were written. */
size_t mfm_engine_get_launch_ms(struct mfm_engine *e, float *dst, size_t cap);
/* MFM_F_TIMING, second-generation kernels: the last launches' durations in the shader's own clocks, oldest first - the
 * longest workgroup's s_memtime (shader-clock ticks) and s_memrealtime (100 MHz reference ticks) difference, stamped by the
 * kernel itself.  shader / ref * 100 MHz is the clock the launch really ran at; shader ticks against the kernel's issue
 * cycles is how much of the launch the SIMDs were issuing (bench.py: roofline.issue_model).  Waits for the launches issued so
 * far and nothing else: samples that were accepted and not yet launched stay where they are (no flush - the call is read-only,
 * also on a device group's shard engines).  At most the last 512 launches.  0 entries for launches that left no stamp; returns the number of entries written (0 without MFM_F_TIMING or on the other
 * kernels).  Either array may be NULL. */
size_t mfm_engine_get_launch_cycles(struct mfm_engine *e, uint64_t *shader_ticks, uint64_t *ref_ticks, size_t cap);

/* The engine's compute stream (hipStream_t) - with MFM_F_OVERLAP the one the most recent launch went to - for callers that
 * order their own work after it. */
void *mfm_engine_stream(struct mfm_engine *e);

const char *mfm_strerror(int err);
const char *mfm_last_error(void); /* thread-local detail of the last failure */

/*
 * ---- one channel set on several GPUs of a node (SURVEY.md section 8b "set_devices", section 8e) -----------------
 * The reference fans every delivered sample_buf out to all channel threads (multifm/receiver.c:78-98).  A device
 * group does the same across GPUs: channels are cut into contiguous shards (mfm_shard_range), one engine per device;
 * mfm_group_push() stages a block on the first device and exchanges it with RCCL over xGMI (ncclBroadcast, or a scatter
 * plus ncclAllGather: MFM_X_*), in place into every other engine's input buffer, then every engine runs its shard - all
 * shards take a block or none does; a failure after the first shard has taken it makes every later call fail with
 * MFM_E_DEVICE rather than let the shards drift apart.  No other exchange.  Blocks come back per
 * shard: mfm_group_fetch() fills one mfm_block per shard, all for the same stream position; channel c of the group is
 * row c - first_channel of its shard's block (mfm_group_shard_info).  One host thread at a time may push, another one
 * fetch/release (as for a single engine).  RCCL (librccl.so) is loaded at run time, and only by groups that exchange.
 */
#define MFM_GROUP_MAX_DEVICES 16
#define MFM_X_AUTO 0u /* one device: direct staging, no RCCL; several: RCCL broadcast */
#define MFM_X_RCCL 1u /* always through the RCCL broadcast path (exercises the call sequence on a one-GPU box) */
#define MFM_X_RCCL_ALLGATHER 2u /* RCCL, large-block form for point-to-point xGMI: the root sends 1/S of the block to each of
                                   its S - 1 peers (ncclSend/ncclRecv, S - 1 links at once), then ncclAllGather in place -
                                   no single link carries the whole block, as it does along a broadcast's ring */

struct mfm_group_config {
    uint32_t abi_version;       /* MFM_ABI_VERSION */
    uint32_t nr_devices;        /* 1..MFM_GROUP_MAX_DEVICES; devices beyond the channel count stay idle */
    int32_t devices[MFM_GROUP_MAX_DEVICES]; /* HIP device ordinals; devices[0] ingests and is the broadcast root */
    uint32_t sample_rate_hz;
    uint32_t decimation;
    uint32_t max_block_samples;
    uint32_t flags;             /* MFM_F_* handed to every engine (MFM_F_DEVICE_ONLY: see mfm_group_submit) */
    uint32_t exchange;          /* MFM_X_* */
    uint32_t coalesce_samples;  /* as mfm_engine_config::coalesce_samples; the shards launch or defer together */
    uint32_t reserved;          /* 0 */
};

struct mfm_group; /* opaque */

/* channels [first, first + count) of nr_channels belong to shard `shard` of nr_shards: contiguous ranges whose sizes
 * differ by at most one (empty only when there are fewer channels than shards).  Pure function. */
void mfm_shard_range(uint32_t nr_channels, uint32_t nr_shards, uint32_t shard, uint32_t *first, uint32_t *count);

int mfm_group_create(struct mfm_group **pg, const struct mfm_group_config *cfg);
void mfm_group_destroy(struct mfm_group **pg);
/* as mfm_engine_add_channel(); returns the channel's index in the group */
int mfm_group_add_channel(struct mfm_group *g, int32_t offset_hz, const double *lpf_taps, size_t nr_taps, double channel_gain,
                          int want_iq);
/* cut the shards, create and commit one engine per non-empty shard, set up the RCCL communicators */
int mfm_group_commit(struct mfm_group *g);
int mfm_group_nr_shards(struct mfm_group *g); /* >= 1 after commit */
int mfm_group_shard_info(struct mfm_group *g, uint32_t shard, uint32_t *first_channel, uint32_t *nr_channels, int32_t *device);
/* Blocks that are in device memory already (a producer kernel, a peer copy, another library's collective wrote them): the
 * ROOT's input buffer is where they go - acquire_input() names the address, as mfm_engine_acquire_input() - and submit()
 * exchanges the nr_samples int16 samples there to the other shards and submits them on every shard, in the same order and
 * under the same all-or-nothing rules as a host block (mfm_group_push).  The caller has made sure the block is complete
 * before it submits.  With MFM_F_DEVICE_ONLY in the group's flags the shards keep their outputs in HBM (no host mirror, no
 * mfm_group_fetch): a throughput measurement, or a consumer that works on the device. */
int mfm_group_acquire_input(struct mfm_group *g, void **d_dst, size_t *capacity_samples);
int mfm_group_submit(struct mfm_group *g, size_t nr_samples);
/* shard `shard`'s engine, for the READ-ONLY engine calls (get_stats, get_launch_ms / _cycles, last_output_device,
 * last_launch_input, get_channel with the shard's own channel numbers); NULL when there is no such shard */
struct mfm_engine *mfm_group_shard_engine(struct mfm_group *g, uint32_t shard);
/* host ingest of one block in any MFM_IN_* format.  MFM_E_BUSY when a shard's output ring is full (nothing was
 * staged on any shard: fetch/release and retry). */
int mfm_group_push(struct mfm_group *g, const void *data, size_t nr_samples, int format);
/* the same out of page-locked memory (mfm_host_alloc), without the staging copy: as mfm_engine_push_pinned() */
int mfm_group_push_pinned(struct mfm_group *g, const void *data, size_t nr_samples, int format, uint64_t *ticket);
/* A RUN of page-locked buffers of nr_samples_each samples that lie stride_bytes apart in one arena - a pool that hands its
 * frames out in address order delivers neighbours one after the other (host/mfm_tsl.c frame_alloc, host/mfm_receiver.c) - goes
 * to the device as ONE strided copy command and is accepted as one block.  One command per 512 KiB sample_buf runs at half the
 * link's rate, one per 16 KiB file_if buffer at a ninth (bench.py end_to_end.link).  *accepted (>= 1 on MFM_OK) = how many
 * buffers of the run were taken: fewer than `count` when less fits the buffer being filled or the gathering policy would have
 * launched in between; the caller offers the rest again.  One ticket covers the accepted buffers. */
int mfm_engine_push_pinned_run(struct mfm_engine *e, const void *first, size_t stride_bytes, size_t nr_samples_each, size_t count,
                               int format, uint64_t *ticket, size_t *accepted);
int mfm_group_push_pinned_run(struct mfm_group *g, const void *first, size_t stride_bytes, size_t nr_samples_each, size_t count,
                              int format, uint64_t *ticket, size_t *accepted);
size_t mfm_engine_input_room(struct mfm_engine *e); /* samples the buffer being filled still takes */
/* mfm_group_replay_pinned() over one arena of buffers stride_bytes apart, in runs of up to max_run neighbours (measurements) */
int mfm_group_replay_arena(struct mfm_group *g, const void *arena, size_t stride_bytes, size_t nr_bufs, size_t buf_samples, int format,
                           size_t nr_pushes, size_t max_run, uint64_t *outputs_per_channel, uint64_t *copy_commands);
int mfm_group_copy_done(struct mfm_group *g, uint64_t ticket);
/* A host loop in C, for measurements (bench.py end_to_end): nr_pushes buffers of buf_samples samples, taken in turn from the
 * caller's nr_bufs page-locked buffers, pushed with mfm_group_push_pinned(); blocks are fetched and released whenever the
 * output rings are full, everything is flushed and drained at the end.  *outputs_per_channel = outputs fetched. */
int mfm_group_replay_pinned(struct mfm_group *g, const void *const *bufs, size_t nr_bufs, size_t buf_samples, int format,
                            size_t nr_pushes, uint64_t *outputs_per_channel);
int mfm_group_copy_wait(struct mfm_group *g, uint64_t ticket);
/* oldest finished block of every shard into blks[0 .. nr_shards); MFM_E_DONE when nothing is pending */
int mfm_group_fetch(struct mfm_group *g, struct mfm_block *blks);
int mfm_group_release(struct mfm_group *g);
/* coalesce_samples: launch, on every shard, what has been pushed and not yet launched (MFM_E_BUSY: fetch / release first) */
int mfm_group_flush(struct mfm_group *g);
/* flush, then wait for every shard (MFM_E_BUSY from the flush is returned as it is: it is not a failure).  Producer-side
 * like the pushes: a host with a submit thread lets THAT thread flush and waits for mfm_stats::pending_samples == 0 and
 * pending_blocks == 0 instead (host/mfm_receiver.c receiver_drain). */
int mfm_group_sync(struct mfm_group *g);
int mfm_group_get_stats(struct mfm_group *g, uint32_t shard, struct mfm_stats *st);
/* whether blocks travel through RCCL, how many blocks were pushed through it and how many bytes it moved to
 * non-root devices */
int mfm_group_exchange_info(struct mfm_group *g, int *uses_rccl, uint64_t *blocks, uint64_t *bytes_exchanged);
/* One shard of the exchange, as measured: what a scaling figure needs to be read (is a step bound by the exchange of the block -
 * multifm/receiver.c:89-95's fan-out, here over xGMI - or by the shard's kernel?).  With MFM_F_TIMING the group brackets the
 * RCCL calls of one block in four with an event pair on every shard's exchange stream. */
struct mfm_exchange_detail {
    int32_t device;            /* HIP device of the shard */
    int32_t rccl_ranks;        /* ncclCommCount of the shard's communicator; 0: the group does not exchange; -1: the library has no such call */
    char pci_bus_id[32];       /* hipDeviceGetPCIBusId of the device */
    uint64_t timed_exchanges;  /* exchanges whose duration is in exchange_ms */
    double exchange_ms;        /* sum of their durations on this shard's exchange stream */
    uint64_t timed_launches;   /* the shard engine's mfm_stats::timed_launches ... */
    double kernel_ms;          /* ... and ::kernel_ms */
    uint32_t bound;            /* MFM_BOUND_KERNEL / MFM_BOUND_EXCHANGE: the larger of the two means; MFM_BOUND_UNKNOWN without both */
    uint32_t reserved0;
};
#define MFM_BOUND_UNKNOWN 0u
#define MFM_BOUND_KERNEL 1u
#define MFM_BOUND_EXCHANGE 2u
int mfm_group_exchange_detail(struct mfm_group *g, uint32_t shard, struct mfm_exchange_detail *out);
/* Which RCCL a device group of more than one GPU uses: loads it as mfm_group_commit() would - the file the environment variable
 * MFM_RCCL_LIBRARY names, if it is set (nothing else is tried then); else a librccl that is mapped into the
 * process already (PyTorch brings its own), else the loader's search for the bare name (LD_LIBRARY_PATH, the cache), then
 * $ROCM_PATH/lib/librccl.so and /opt/rocm/lib/librccl.so - and writes the file's path.  MFM_E_DEVICE when none can be loaded. */
int mfm_group_rccl_library(char *path, size_t cap);

/*
 * ---- PCM stage behind the FIFO -------------------------------------------
 * The decoder / resampler processes read a channel's PCM FIFO and run it through a real-valued rational
 * resampler and an optional DC blocker before the protocol decoders (decoder/decoder.c:580-673):
 *
 *   polyphase_fir_new(&fir, nr_coeffs, q14_coeffs, interpolate, decimate)   filter/polyphase_fir.c:47-105
 *   polyphase_fir_push_sample_buf / polyphase_fir_process                    filter/polyphase_fir.c:162-233
 *   dc_blocker_init(pole) / dc_blocker_apply                                 filter/dc_blocker.h:45-93
 *
 * mfm_resampler does that for ALL channels of an engine at once, on PCM that is still in HBM (the output
 * of mfm_engine_submit) or handed in from the host.  Same arithmetic, bit for bit: int16 x int16 -> wrapping
 * int32 dot product of one phase filter with consecutive samples, Q14 rounding, phase walk
 * phase += D; consumed = phase / I; phase %= I, an output only while MORE than one phase length of
 * unconsumed samples exists (polyphase_fir.c:184).
 */
struct mfm_resampler; /* opaque */

struct mfm_resampler_config {
    uint32_t abi_version;    /* MFM_ABI_VERSION */
    int32_t device;
    uint32_t nr_channels;
    uint32_t interpolate;    /* I */
    uint32_t decimate;       /* D */
    uint32_t max_in_samples; /* most PCM samples per channel one process call may carry */
    uint32_t invert;         /* decoder -i: negate the input samples (decoder.c:621-626) */
    uint32_t dc_block;       /* decoder -b */
    double dc_pole;          /* decoder -p, only with dc_block */
    uint32_t flags;          /* MFM_RS_* */
    uint32_t reserved;       /* 0 */
};

#define MFM_RS_FORCE_DOT2 1u /* the v_dot2 kernel even where the matrix-core form applies (A/B timing; same bits out) */

/* coeffs are the Q14 int16 taps (decoder.c:530-533 quantises lpfCoeffs with (int16_t)(c * 16384)) */
int mfm_resampler_create(struct mfm_resampler **pr, const struct mfm_resampler_config *cfg, const int16_t *coeffs,
                         size_t nr_coeffs);
void mfm_resampler_destroy(struct mfm_resampler **pr);
/* Upper bound of outputs per channel one process call can produce. */
size_t mfm_resampler_max_out(const struct mfm_resampler *r);
/*
 * Consume nr_in PCM samples per channel from device memory laid out [channel][in_stride] (for instance the
 * pointer/stride of mfm_engine_last_output_device) and produce *nr_out resampled samples per channel at
 * *d_out, laid out [channel][*out_stride], valid until the next call.  Work is queued on `stream` (a
 * hipStream_t, NULL = legacy default stream); no host synchronisation.
 */
int mfm_resampler_process_device(struct mfm_resampler *r, const int16_t *d_pcm, size_t in_stride, size_t nr_in,
                                 void *stream, int16_t **d_out, size_t *out_stride, size_t *nr_out);
/* Host in, device out: the PCM is staged to the device on `stream` (what a decoder-shaped host reading FIFOs
 * uses, so that only the input crosses PCIe); otherwise as mfm_resampler_process_device. */
int mfm_resampler_process_host_to_device(struct mfm_resampler *r, const int16_t *pcm, size_t in_stride, size_t nr_in,
                                         void *stream, int16_t **d_out, size_t *out_stride, size_t *nr_out);
/* Host convenience (tests, harnesses): same, host in / host out, synchronous; out is [channel][out_stride]. */
int mfm_resampler_process_host(struct mfm_resampler *r, const int16_t *pcm, size_t in_stride, size_t nr_in,
                               int16_t *out, size_t out_stride, size_t *nr_out);

/*
 * ---- Pager stage: POCSAG slicer / sync / batch collection + BCH(31,21) -----
 * Replaces, for ALL channels at once and on PCM that is still in HBM (38 400 Hz, i.e. the resampler's output):
 *
 *   pager_pocsag_on_pcm            pager/pager_pocsag.c:434-543   state machine SEARCH -> SYNCHRONIZED ->
 *                                                                  BATCH_RECEIVE -> SEARCH_SYNCWORD
 *   _pager_pocsag_baud_on_sample   pager/pager_pocsag.c:81-117    three eye detectors (75 / 32 / 16 samples/bit)
 *   bch_code_decode                pager/bch_code.c:307-398       on the 16 words of every batch (:332-334)
 *
 * Output is an event list per channel (sync found, batch of 16 raw + corrected words with the BCH verdicts, sync
 * kept / lost) - everything _pager_pocsag_process_batch (:319-432) needs to assemble pages.  That last step is
 * a byte-serial walk over at most 16 words per batch and stays on the host (tsl-sdr_amd/host/mfm_pager_pocsag.c,
 * same callback signatures as pager/pager_pocsag.h:29-46).
 *
 * Conventions kept: bit = (sample < 0); sync = popcount(word ^ 0x7cd215d8) <= 4; eye open when more than
 * samples_per_bit/2 consecutive matches, sampling offset = matches/2; batch words filled LSB first; the word is
 * masked with 0x7fffffff before BCH; a word that fails BCH ends the batch for the message layer (nr_ok).
 */
#define MFM_POCSAG_EV_SYNC_FOUND 1u
#define MFM_POCSAG_EV_BATCH      2u
#define MFM_POCSAG_EV_SYNC_LOST  3u
#define MFM_POCSAG_EV_SYNC_KEPT  4u

struct mfm_pocsag_event {
    uint32_t type;          /* MFM_POCSAG_EV_* */
    uint32_t baud;          /* 512 / 1200 / 2400 */
    uint32_t channel;
    uint32_t aux;           /* SYNC_FOUND: eye matches; SYNC_LOST / SYNC_KEPT: the 32 bits seen in the sync slot */
    uint64_t sample;        /* index (per channel, since creation) of the PCM sample that completed the event */
    uint32_t nr_ok;         /* BATCH: words accepted before the first BCH failure (16 = whole batch) */
    uint32_t fail_mask;     /* BATCH: bit z set when word z is uncorrectable */
    uint32_t raw[16];       /* BATCH: words as collected */
    uint32_t corrected[16]; /* BATCH: (raw & 0x7fffffff) after BCH correction */
};

struct mfm_pocsag; /* opaque */

struct mfm_pocsag_config {
    uint32_t abi_version;    /* MFM_ABI_VERSION */
    int32_t device;
    uint32_t nr_channels;
    uint32_t max_in_samples; /* most PCM samples per channel one process call may carry */
    uint32_t max_events;     /* per channel and call; 0 = max_in_samples / 2048 + 16 (cannot overflow) */
    uint32_t flags;          /* 0 */
};

int mfm_pocsag_create(struct mfm_pocsag **pp, const struct mfm_pocsag_config *cfg);
void mfm_pocsag_destroy(struct mfm_pocsag **pp);
/*
 * Consume nr_in PCM samples per channel, laid out [channel][in_stride] in device memory (for instance the
 * output of mfm_resampler_process_device).  Work is queued on `stream`; no host synchronisation.  The events of
 * THIS call replace those of the previous one.
 */
int mfm_pocsag_process_device(struct mfm_pocsag *p, const int16_t *d_pcm, size_t in_stride, size_t nr_in,
                              void *stream);
/* Host convenience: same from host memory, synchronous. */
int mfm_pocsag_process_host(struct mfm_pocsag *p, const int16_t *pcm, size_t in_stride, size_t nr_in);
/*
 * Wait for the last process call and copy its events: channels ascending, stream order within a channel.
 * MFM_E_NOMEM when `max_events` is too small (nothing copied, *nr_events = needed), MFM_E_STATE when a channel
 * overflowed its device-side event list (only possible with a caller-chosen max_events).
 */
int mfm_pocsag_fetch_events(struct mfm_pocsag *p, struct mfm_pocsag_event *out, size_t max_events,
                            size_t *nr_events);

/*
 * ---- Pager stage: FLEX sync 1 / frame information word / sync 2 / block de-interleave -------
 * Replaces, for ALL channels at once and on PCM that is still in HBM (16 000 Hz, pager_flex_priv.h:231):
 *
 *   pager_flex_on_pcm              pager/pager_flex.c:1401-1455   SYNC_1 -> SYNC_2 -> BLOCK, sample skipping
 *   _pager_flex_sync_update        pager/pager_flex.c:295-458     ten-phase BS1 search, eye run, A / B / inverted A,
 *                                                                  frame information word, swing of the signal
 *   _pager_flex_handle_fiw         pager/pager_flex.c:1312-1345   BCH(31,21) + checksum of the FIW, cycle / frame
 *   _pager_flex_sync2_update       pager/pager_flex.c:460-525     counts the 25 ms of sync 2
 *   _pager_flex_block_update       pager/pager_flex.c:1224-1310   2- / 4-level slicer, phase split, block de-interleave
 *
 * Output per channel: an event list (frame collected / unknown A code / bad FIW) and, for every frame, the 88 words
 * of each phase exactly as _pager_flex_phase_append_bit (:1200-1222) leaves them.  What follows in the reference,
 * _pager_flex_phase_process (:1088-1198: BIW, addresses, vectors, message bodies), is a serial walk over at most
 * 88 words with in-place corrections; it stays on the host (tsl-sdr_amd/host/mfm_pager_flex.c, same callback
 * signatures as pager/pager_flex.h:16-87).
 *
 * Conventions kept: bit = (sample >= 0); BS1 = 0xaaaaaaaa seen by one of ten registers that take every tenth
 * sample; a run of three or more consecutive matching samples opens the eye and the sampling clock is set to half
 * the run length (the run counter is 8 bits wide, as in the reference); only the upper half of A is compared, with
 * fewer than 4 differing bits (the inverted-A comparison of :278 can never succeed and is not evaluated); swing =
 * mean of the positive minus mean of the non-positive samples over the 112 sync bits, all in int16 arithmetic;
 * registers are zero-filled after every reset, so no match is possible for the next 310 samples.
 * A sync run without a positive or without a non-positive sample (the reference divides by zero) is reported as
 * MFM_FLEX_EV_BAD_FIW with fiw_rc 3.
 */
#define MFM_FLEX_EV_FRAME    1u
#define MFM_FLEX_EV_BAD_BAUD 2u
#define MFM_FLEX_EV_BAD_FIW  3u
#define MFM_FLEX_PHASE_WORDS 88u /* pager_flex_priv.h:175 */

struct mfm_flex_event {
    uint32_t type;          /* MFM_FLEX_EV_* */
    uint32_t channel;
    uint64_t sample;        /* index (per channel, since creation) of the PCM sample that completed the event */
    uint64_t sync_sample;   /* FRAME: sample of the last FIW bit */
    uint32_t coding;        /* index into _pager_codings[] (pager_flex.c:46-96); 0xffffffff for BAD_BAUD */
    uint32_t baud;          /* 1600 / 3200 / 6400; 0 for BAD_BAUD */
    uint32_t eye;           /* length of the BS1 run (sync->bit_counter at :339) */
    uint32_t a, b, inv_a;   /* the sync words as collected */
    uint32_t fiw_raw;       /* the 32 FIW bits as collected */
    uint32_t fiw;           /* (fiw_raw & 0x7fffffff) after BCH correction */
    uint32_t fiw_rc;        /* 0 accepted, 1 uncorrectable, 2 checksum, 3 no swing */
    int32_t sample_range;   /* flex->sample_range / sample_delta (:441-442) */
    int32_t sample_delta;
    uint32_t cycle, frame;  /* FIW fields (:1337-1338) */
    uint32_t frame_index;   /* FRAME: index of this frame's words in the array the same fetch call fills */
    uint32_t nr_phases;     /* 1 / 2 / 4 */
    uint32_t reserved;
};

/* phase_words[] of phases A..D of one frame; phases the coding does not carry are zero */
struct mfm_flex_frame_words {
    uint32_t words[4][MFM_FLEX_PHASE_WORDS];
};

struct mfm_flex; /* opaque */

struct mfm_flex_config {
    uint32_t abi_version;    /* MFM_ABI_VERSION */
    int32_t device;
    uint32_t nr_channels;
    uint32_t max_in_samples; /* most PCM samples per channel one process call may carry (<= 2^26) */
    uint32_t max_events;     /* per channel and call; 0 = max_in_samples / 1024 + 8 (cannot overflow) */
    uint32_t flags;          /* 0 */
};

int mfm_flex_create(struct mfm_flex **pf, const struct mfm_flex_config *cfg);
void mfm_flex_destroy(struct mfm_flex **pf);
/*
 * Consume nr_in PCM samples per channel, laid out [channel][in_stride] in device memory (for instance the output of
 * mfm_resampler_process_device).  Work is queued on `stream`; no host synchronisation; d_pcm is read until the
 * queued work has run.  The events of THIS call replace those of the previous one.
 */
int mfm_flex_process_device(struct mfm_flex *f, const int16_t *d_pcm, size_t in_stride, size_t nr_in, void *stream);
/* Host convenience: same from host memory, synchronous. */
int mfm_flex_process_host(struct mfm_flex *f, const int16_t *pcm, size_t in_stride, size_t nr_in);
/*
 * Wait for the last process call and copy its events (channels ascending, stream order within a channel) and the
 * words of its frames.  MFM_E_NOMEM when either array is too small (nothing copied, *nr_events / *nr_frames =
 * needed), MFM_E_STATE when a channel overflowed its device-side lists (only with a caller-chosen max_events).
 */
int mfm_flex_fetch_events(struct mfm_flex *f, struct mfm_flex_event *events, size_t max_events, size_t *nr_events,
                          struct mfm_flex_frame_words *frames, size_t max_frames, size_t *nr_frames);

/*
 * ---- Mueller-Muller clock recovery (BASELINE.json configs[3]: "mueller_muller slicer") -------------------------
 *   mm_init      pager/mueller_muller.c:10-33
 *   mm_process   pager/mueller_muller.c:41-115
 * for all channels of a PCM block at once, one loop state per channel, same float operations in the same order
 * (decisions are bit-identical to a build of the reference without FP contraction).  The reference's live decoder
 * does not call it (pager_pocsag.c has its own eye detectors); its consumer is pager/test/test_mueller_muller.c:
 * decisions[i] > 0 ? 0 : 1 shifted into a 32-bit register and compared with the POCSAG sync word.
 * As in the reference (:66-67) the index of a decision can reach nr_in: when in_stride > nr_in that sample is read
 * (the next block's first one, if the caller slices a longer buffer as the reference's test does), otherwise the
 * last sample stands in for it.
 */
struct mfm_mm; /* opaque */

struct mfm_mm_config {
    uint32_t abi_version; /* MFM_ABI_VERSION */
    int32_t device;
    uint32_t nr_channels;
    uint32_t max_in_samples; /* < 2^22: the reference's float sample position stays a whole number it holds exactly */
    float kw, km, samples_per_bit, error_min, error_max; /* mm_init's arguments; MFM_E_INVAL unless kw is finite,
                                                          * error_min <= error_max, error_min - |km| * 32768 >= 1 (a
                                                          * step cannot reach zero) and error_max + |km| * 32768 < 2^20 */
};

int mfm_mm_create(struct mfm_mm **pm, const struct mfm_mm_config *cfg);
void mfm_mm_destroy(struct mfm_mm **pm);
size_t mfm_mm_max_decisions(const struct mfm_mm *m);
/* decisions: [channel][*dec_stride] int16 (the samples picked, :71), counts: [channel] decisions of this call;
 * both in device memory, valid until the next call; queued on `stream`.  A row holds mfm_mm_max_decisions()
 * decisions (sized so that no configuration create accepts can exceed it); what lies behind them in the row is
 * scratch. */
int mfm_mm_process_device(struct mfm_mm *m, const int16_t *d_pcm, size_t in_stride, size_t nr_in, void *stream,
                          int16_t **d_decisions, size_t *dec_stride, uint32_t **d_counts);
/* Host convenience: synchronous; MFM_E_NOMEM if a channel produced more than dec_stride decisions. */
int mfm_mm_process_host(struct mfm_mm *m, const int16_t *pcm, size_t in_stride, size_t nr_in, int16_t *decisions,
                        size_t dec_stride, uint32_t *counts);

/* bch_code_decode (pager/bch_code.c:307-398) on n words in place; rc[i] = its return value (0 or 1). */
int mfm_bch3121_decode_device(uint32_t *d_words, uint8_t *d_rc, size_t n, int device, void *stream);
int mfm_bch3121_decode_host(uint32_t *words, uint8_t *rc, size_t n, int device);

/*
 * ---- Floating-point IQ path (BASELINE.json configs[4] "fp32 vs int16 IQ path") ---------
 * The reference has no floating-point channel path.  This is multifm's per-channel loop
 *
 *   _demod_fir_prepare               multifm/demod.c:204-269       taps = (gain * cexp(j f_offs i)) * h[i], NOT quantised
 *   direct_fir_process               filter/direct_fir.c:328-453   complex FIR, decimation D, derotation by
 *                                                                  cexp(-j 2 pi off D n / fs) (closed form of :151-172)
 *   multifm_fm_demod_process         multifm/fm_demod.c:36-85      s = o[n] conj(o[n-1]), fast_atan2f, phi/pi*16384
 *
 * on float32 interleaved IQ (any scale) in fp32 arithmetic.  Parity target: oracle/f32_oracle.c (the same in fp64),
 * 1e-5 relative (tests/test_f32_path.py).  PCM comes out as float and, truncated like fm_demod.c:72, as int16 laid
 * out like the integer engine's ([channel][stride]) so that mfm_resampler_* / mfm_pocsag_* take it unchanged.
 * All channels of an engine share one filter length (>= the decimation).
 */
struct mfm_f32_engine; /* opaque */

#define MFM_F32_WANT_IQ 1u /* also keep the derotated filtered samples (signalDebugFile analogue) */
#define MFM_F32_PACKED_FMA 2u /* multiply with v_pk_fma_f32 instead of the fp32 matrix instructions (A/B timing) */
#define MFM_F32_TILE_KERNEL 4u /* the round-1 kernel, one workgroup per tile, instead of the persistent one (A/B timing) */

struct mfm_f32_config {
    uint32_t abi_version; /* MFM_ABI_VERSION */
    int32_t device;
    uint32_t sample_rate_hz;
    uint32_t decimation;
    uint32_t max_block_samples; /* most IQ samples one process call may carry */
    uint32_t flags;             /* MFM_F32_* */
};

struct mfm_f32_block {
    float *d_pcm_f32;     /* device, [channel][stride] */
    int16_t *d_pcm_i16;   /* device, [channel][stride] */
    float *d_iq_f32;      /* device, [channel][stride] (re, im) pairs, NULL without MFM_F32_WANT_IQ */
    size_t stride;        /* elements between channels */
    size_t nr_out;        /* outputs per channel of this call */
    uint32_t nr_channels;
    uint32_t reserved;
};

int mfm_f32_create(struct mfm_f32_engine **pe, const struct mfm_f32_config *cfg);
/* same arguments as demod_thread_new's offset / taps / gain (multifm/demod.h:104-110); returns the channel index */
int mfm_f32_add_channel(struct mfm_f32_engine *e, int32_t offset_hz, const double *lpf_taps, size_t nr_taps,
                        double gain);
int mfm_f32_commit(struct mfm_f32_engine *e);
void mfm_f32_destroy(struct mfm_f32_engine **pe);
size_t mfm_f32_max_out(const struct mfm_f32_engine *e);
/* nr_samples float IQ pairs in device memory; work is queued on `stream`, the block's pointers are valid until the
 * next call.  The stream position (unconsumed samples, output phase, discriminator history) carries over. */
int mfm_f32_process_device(struct mfm_f32_engine *e, const float *d_iq, size_t nr_samples, void *stream,
                           struct mfm_f32_block *out);
/* Host convenience (tests): synchronous, any of the three outputs may be NULL; out_stride in elements. */
int mfm_f32_process_host(struct mfm_f32_engine *e, const float *iq, size_t nr_samples, float *pcm_f32,
                         int16_t *pcm_i16, float *iq_f32, size_t out_stride, size_t *nr_out);

/*
 * Host twins of the kernel's scalar numerics (compiled from the same header the kernel uses).
 * They exist so the test-suite can check, on the CPU, that the device formulas reproduce the
 * reference's expressions bit for bit; they are not a compute path.
 */
int32_t mfm_hosttwin_discriminate(int32_t s_re, int32_t s_im);
void mfm_hosttwin_discriminate_batch(const int32_t *s_re, const int32_t *s_im, size_t n, int16_t *out);
int16_t mfm_hosttwin_r14(int32_t a);
/* out[i] = pcm of the non-negative angle whose float bit pattern is first_bits + i */
void mfm_hosttwin_pcm_range(uint32_t first_bits, uint32_t count, int16_t *out);
void mfm_hosttwin_atan_table(float tbl[257]);
int mfm_hosttwin_atan_table_ok(void); /* 1 if the generated table matches the pinned hash */
/* The discriminator (multifm/fm_demod.c:68-72 on fast_atan2f.c:101-174) exactly as the channel kernel of `variant`
 * (mfm_stats::kernel_variant: 0, 1, 2) computes it ON THE DEVICE, for caller-supplied products s = q conj(prev): the three
 * kernels carry three renderings of it (scalar, packed, four at a time), and tests run the hard cases of its division
 * through each (tools/div_proof.c).  Host arrays in and out; not a compute path. */
int mfm_devtest_discriminate(int variant, const int32_t *s_re, const int32_t *s_im, size_t n, int16_t *pcm_out, int device);
/* What v_rcp_f32 returns on `device` for all 2^23 binary32 significands, as a hash (+ how many reciprocals are one ulp low /
 * correctly rounded / one ulp high / anything else), and optionally the discriminator's division against the IEEE quotient
 * on 2^28 significand pairs.  mfm_engine_commit() compares the hash with MFM_RCP_TABLE_HASH_GFX950 - the table the
 * division's correctness proof (tools/div_proof.c) enumerated - once per device, and falls back to the sweep when it
 * differs: a device with another reciprocal table is accepted only if not one quotient is off. */
#define MFM_RCP_TABLE_HASH_GFX950 0x706d94bc005bcc1aull /* (read off an MI355X: tests/test_gpu_parity.py checks it there) */
int mfm_devtest_rcp_table(int device, uint64_t *hash, uint64_t counts[4], uint64_t *sweep_bad, uint64_t *sweep_tried);
/* the device's table-driven BCH(31,21) decode (syndrome bytes -> 1024-entry flip table), on the host */
int mfm_hosttwin_bch3121_decode(uint32_t *word);

#ifdef __cplusplus
}
#endif

#endif /* MULTIFM_HIP_H */
