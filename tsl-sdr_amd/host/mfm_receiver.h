/*
 * mfm_receiver.h - multifm's receiver / demod_thread / sample_buf interface on top of the MI355X engine.
 *
 * Same names, arguments and error behaviour as the reference's
 *     filter/sample_buf.h:59-104   struct sample_buf, sample_buf_decref()
 *     multifm/receiver.h:18-124    struct receiver, receiver_init/start/cleanup/set_mute/thread_running,
 *                                  receiver_sample_buf_alloc/deliver
 *     multifm/demod.h:104-116      demod_thread_new / demod_thread_delete
 * so a front end written against them (file_if, rtl_sdr_if, airspy_if, uhd_if) compiles and behaves the
 * same: allocate a sample_buf from the receiver's pool, fill data_buf with interleaved int16 I,Q, set
 * nr_samples, deliver.  What differs is behind the interface: a demod_thread is a channel REGISTRATION on
 * the receiver's device group (no pthread per channel; "gpuDevices": [..] in the receiver configuration shards the
 * channels over several GPUs of the node, mfm_group_*); deliver() only queues the buffer's pointer (refcount 1) and
 * returns - a submit thread copies it to the device(s) and gives it back to the pool, a drain thread writes every
 * channel's PCM to its FIFO.  The front end's thread never waits for the GPU: when the device falls behind, the pool
 * runs dry and receiver_sample_buf_alloc() drops + counts exactly as in the reference (multifm/receiver.c:57-63).
 */
#pragma once

#include <semaphore.h>
#include <stdatomic.h>

#include "mfm_config.h"
#include "mfm_tsl.h"

#include "../../include/multifm_hip.h"

/* ---- sample buffers (filter/sample_buf.h) ---- */

enum sample_type {
    UNKNOWN = 0,
    REAL_UINT_16 = 1,
    COMPLEX_UINT_16 = 2,
    COMPLEX_INT_16 = 3,
    REAL_UINT_32 = 4,
    COMPLEX_UINT_32 = 5,
    /* not in the reference: data_buf holds 8-bit IQ pairs exactly as a front end received them; the engine widens
     * them on the GPU the way that front end would have on the host (mfm_engine_push_bytes, MFM_IN_*) */
    RAW_COMPLEX_INT_8 = 101,       /* file_if "cs8" */
    RAW_COMPLEX_FILE_UINT_8 = 102, /* file_if "cu8" */
    RAW_COMPLEX_RTLSDR_UINT_8 = 103,
};

struct sample_buf;
typedef aresult_t (*sample_buf_release_func_t)(struct sample_buf *buf);

struct sample_buf {
    uint32_t refcount CAL_ALIGN(16);   /* atomically decremented; buffer is released at 0 */
    enum sample_type sample_type;
    uint32_t nr_samples;               /* complex samples in data_buf */
    uint32_t sample_buf_bytes;
    uint64_t start_time_ns;
    sample_buf_release_func_t release;
    void *priv;
    uint8_t data_buf[];                /* I,Q,I,Q... int16 */
};

aresult_t sample_buf_decref(struct sample_buf *buf);

/* ---- channels (multifm/demod.h) ---- */

/* A doorbell between two threads: the waiter announces that it is about to sleep, looks once more, and sleeps on the
 * semaphore; whoever produces work rings only when somebody announced (one atomic exchange, no lock, never waits - which
 * is what receiver_sample_buf_deliver() needs).  The reference hands buffers to its channel threads through a work queue
 * and a condition variable (multifm/demod.c:134-150); this is that hand-off with a producer side that cannot block. */
struct mfm_doorbell {
    sem_t sem;
    _Atomic int sleeping;
};

struct demod_thread {
    struct list_entry dt_node;
    int chan_index;          /* channel number inside the receiver's engine */
    int shard;               /* which shard of the device group holds it, and which row of that shard's blocks it is */
    size_t shard_row;        /*   (set at receiver_start(), when the group is committed) */
    int fifo_fd;             /* PCM sink (demod.c:331) */
    int debug_signal_fd;     /* filtered-IQ sink or -1 (demod.c:322-328) */
    size_t total_nr_demod_samples;
    size_t total_nr_pcm_samples;
    size_t nr_dropped_samples; /* EPIPE policy of demod.c:93-110 */
};

/*
 * demod_thread_new() keeps the reference signature.  core_id is ignored (there is no thread to pin).
 * The channel is registered on the device group made current with demod_thread_bind_group(), which
 * receiver_init() does around its channel loop.
 */
aresult_t demod_thread_new(struct demod_thread **pthr, unsigned core_id, int32_t offset_hz, uint32_t samp_hz,
                           const char *out_fifo, int decimation_factor, const double *lpf_taps, size_t lpf_nr_taps,
                           const char *fir_debug_output, double channel_gain);
aresult_t demod_thread_delete(struct demod_thread **pthr);
void demod_thread_bind_group(struct mfm_group *group);

/* ---- receiver (multifm/receiver.h) ---- */

struct receiver;
struct mfm_rx_lifeline;
typedef aresult_t (*receiver_cleanup_func_t)(struct receiver *rx);
typedef aresult_t (*receiver_rx_thread_func_t)(struct receiver *rx);

struct receiver {
    _Atomic bool muted; /* set by the main thread (receiver_set_mute, receiver_cleanup), read by the front end's callback: plain
                           `rx->muted` in a front end's source (multifm/rtl_sdr_if.c:94) is then an atomic load */
    struct list_entry demod_threads;
    size_t nr_demod_threads;
    _Atomic size_t nr_samp_buf_alloc_fails; /* written by the front end, read by whoever reports */
    struct frame_alloc *samp_alloc;
    struct worker_thread wthr;
    receiver_cleanup_func_t cleanup_func;
    receiver_rx_thread_func_t thread_func;

    /* MI355X build */
    struct mfm_group *group;      /* the channel set on one or more devices */
    int nr_shards;                /* after receiver_start() */
    struct worker_thread submit_thr, drain_thr;
    struct sample_buf **ring;     /* delivered, not yet submitted buffers: single producer (front end), single
                                     consumer (submit thread); as many slots as the pool has frames, so it never fills */
    size_t ring_slots;
    size_t frame_stride;          /* bytes between neighbouring sample_bufs of the pool's slab */
    _Atomic size_t nr_copy_commands; /* H2D copy commands issued: fewer than buffers when runs of neighbours went as one */
    /* buffers handed to the device group and not yet read by its H2D copy (the pool is page-locked memory: the copy
     * engine reads data_buf where the front end wrote it).  Only the submit thread touches these. */
    struct sample_buf **copying;
    uint64_t *copy_ticket;
    size_t copy_head, copy_tail;
    /* shared between the front end's thread, the submit thread and the drain thread: C11 atomics (sequentially
     * consistent by default; a slot of `ring` is published by the store to ring_head that follows it) */
    _Atomic size_t ring_head, ring_tail;
    _Atomic bool input_done;      /* front end reached end of input */
    _Atomic int closing;          /* receiver_cleanup() has begun: receiver_sample_buf_alloc() hands out nothing any more */
    _Atomic int front_end_holds;  /* buffers the front end got from receiver_sample_buf_alloc() and has not yet delivered or released */
    _Atomic int failed;           /* a device error stopped the submit or the drain thread (A_E_DEVICE from then on) */
    _Atomic size_t nr_bufs_delivered, nr_bufs_submitted;
    _Atomic size_t nr_blocks_drained;
    _Atomic uint64_t max_deliver_ns; /* longest receiver_sample_buf_deliver() call so far */
    struct mfm_doorbell ring_bell;   /* front end -> submit thread: a buffer is in the ring */
    struct mfm_doorbell room_bell;   /* drain thread -> submit thread: an output slot was released */
    struct mfm_doorbell block_bell;  /* submit thread -> drain thread: a block was pushed to the devices */
    struct mfm_doorbell idle_bell;   /* both -> receiver_drain(): a buffer was submitted / a block written out */
    struct mfm_rx_lifeline *life;    /* outlives this structure: see receiver_cleanup() */
};

aresult_t receiver_init(struct receiver *rx, struct config *cfg, receiver_rx_thread_func_t rx_func,
                        receiver_cleanup_func_t cleanup_func, size_t samples_per_buf);
aresult_t receiver_start(struct receiver *rx);
aresult_t receiver_cleanup(struct receiver **prx);
aresult_t receiver_sample_buf_alloc(struct receiver *rx, struct sample_buf **pbuf);
aresult_t receiver_set_mute(struct receiver *rx, bool mute);
aresult_t receiver_sample_buf_deliver(struct receiver *rx, struct sample_buf *buf);
bool receiver_thread_running(struct receiver *rx);

/* additions of this build: a finite input (file) can tell the receiver it is finished, and a driver can
 * wait until everything delivered so far has reached the FIFOs */
void receiver_mark_input_done(struct receiver *rx);
aresult_t receiver_drain(struct receiver *rx);
