/*
 * mfm_receiver.c - receiver / demod_thread / sample_buf plumbing of multifm on the MI355X engine.
 *
 * Follows the behaviour of the reference's multifm/receiver.c, multifm/demod.c (set-up and FIFO output
 * only; the per-buffer DSP loop demod.c:48-121 is the engine's kernel) and filter/sample_buf.c.
 */
#include "mfm_receiver.h"

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdatomic.h>
#include <time.h>
#include <unistd.h>

#define MFM_MSG(sev, sys, msg, ...) MESSAGE("MULTIFM", sev, sys, msg, ##__VA_ARGS__)

/* ---- doorbells (mfm_receiver.h) ---- */

static void _bell_init(struct mfm_doorbell *b)
{
    TSL_BUG_ON(0 != sem_init(&b->sem, 0, 0));
    atomic_store(&b->sleeping, 0);
}

/* producer side: never blocks, one atomic exchange when nobody sleeps */
static void _bell_ring(struct mfm_doorbell *b)
{
    if (1 == atomic_exchange(&b->sleeping, 0)) {
        (void)sem_post(&b->sem);
    }
}

/* waiter side: _bell_arm(), look for work once more, then _bell_sleep() - a ring between the two is not lost.  The sleep is
 * bounded so that a thread also notices a shutdown request that nobody rang for. */
static void _bell_arm(struct mfm_doorbell *b)
{
    atomic_store(&b->sleeping, 1);
}

static void _bell_disarm(struct mfm_doorbell *b)
{
    if (0 == atomic_exchange(&b->sleeping, 0)) {
        (void)sem_trywait(&b->sem); /* a ring slipped in: take its post so that the next sleep does not return at once */
    }
}

static void _bell_sleep(struct mfm_doorbell *b, unsigned max_ms)
{
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_nsec += (long)max_ms * 1000000L;
    ts.tv_sec += ts.tv_nsec / 1000000000L;
    ts.tv_nsec %= 1000000000L;
    while (0 != sem_timedwait(&b->sem, &ts) && EINTR == errno) {
    }
    atomic_store(&b->sleeping, 0);
}

#define MFM_RUN_MAX 64u /* buffers per strided copy command, at most (16 reach 94 % of the link's rate, bench.py end_to_end.link) */
#define MFM_IDLE_MS 50u /* how long a thread sleeps at most before it looks at its run flag again */

/* ---- sample buffers ---- */

aresult_t sample_buf_decref(struct sample_buf *buf)
{
    TSL_ASSERT_ARG(NULL != buf);
    /* filter/sample_buf.c:31-43: the holder that takes the count to zero releases the buffer */
    if (1 == atomic_fetch_sub((_Atomic uint32_t *)&buf->refcount, 1)) {
        TSL_BUG_ON(NULL == buf->release);
        TSL_BUG_IF_FAILED(buf->release(buf));
    }
    return A_OK;
}

static aresult_t _sample_buf_release(struct sample_buf *buf)
{
    TSL_ASSERT_ARG(NULL != buf);
    TSL_BUG_ON(atomic_load((_Atomic uint32_t *)&buf->refcount) != 0);
    struct receiver *rx = buf->priv;
    return frame_free(rx->samp_alloc, (void **)&buf);
}

/* ... of a buffer the front end got from receiver_sample_buf_alloc() and lets go of without delivering it (end of a file,
 * multifm/airspy_if.c:71-75): it no longer holds one */
static aresult_t _sample_buf_release_undelivered(struct sample_buf *buf)
{
    TSL_ASSERT_ARG(NULL != buf);
    struct receiver *rx = buf->priv;
    const aresult_t ret = _sample_buf_release(buf);
    atomic_fetch_sub(&rx->front_end_holds, 1);
    return ret;
}

/* ---- channels ---- */

static struct mfm_group *g_bound_group; /* set by receiver_init() around its channel loop */

void demod_thread_bind_group(struct mfm_group *group)
{
    g_bound_group = group;
}

aresult_t demod_thread_new(struct demod_thread **pthr, unsigned core_id, int32_t offset_hz, uint32_t samp_hz,
                           const char *out_fifo, int decimation_factor, const double *lpf_taps, size_t lpf_nr_taps,
                           const char *fir_debug_output, double channel_gain)
{
    aresult_t ret = A_OK;
    struct demod_thread *thr = NULL;

    TSL_ASSERT_ARG(NULL != pthr);
    TSL_ASSERT_ARG(NULL != out_fifo && '\0' != *out_fifo);
    TSL_ASSERT_ARG(0 != decimation_factor);
    TSL_ASSERT_ARG(NULL != lpf_taps);
    TSL_ASSERT_ARG(0 != lpf_nr_taps);
    (void)core_id;
    (void)samp_hz; /* the engine was created with the receiver's sample rate and decimation */

    *pthr = NULL;
    if (NULL == g_bound_group) {
        MFM_MSG(SEV_FATAL, "NO-ENGINE", "demod_thread_new() outside receiver_init(): no device group is bound");
        return A_E_INVAL;
    }
    if (FAILED(ret = TZAALLOC(thr, SYS_CACHE_LINE_LENGTH))) {
        return ret;
    }
    thr->fifo_fd = -1;
    thr->debug_signal_fd = -1;
    list_init(&thr->dt_node);

    const bool want_iq = NULL != fir_debug_output && '\0' != *fir_debug_output;
    int chan = mfm_group_add_channel(g_bound_group, offset_hz, lpf_taps, lpf_nr_taps, channel_gain, want_iq);
    if (chan < 0) {
        MFM_MSG(SEV_FATAL, "BAD-CHANNEL", "Channel at offset %d Hz rejected: %s", offset_hz, mfm_last_error());
        ret = A_E_INVAL;
        goto done;
    }
    thr->chan_index = chan;

    /* same order and flags as multifm/demod.c:322-331: debug file first, then the FIFO, both O_WRONLY
     * (opening a FIFO blocks until a reader shows up) */
    if (want_iq) {
        if (0 > (thr->debug_signal_fd = open(fir_debug_output, O_WRONLY))) {
            MFM_MSG(SEV_FATAL, "CANT-OPEN-SIGNAL-DEBUG", "Unable to open signal debug dump file '%s'", fir_debug_output);
            ret = A_E_INVAL;
            goto done;
        }
    }
    if (0 > (thr->fifo_fd = open(out_fifo, O_WRONLY))) {
        MFM_MSG(SEV_FATAL, "CANT-OPEN-FIFO", "Unable to open output fifo '%s'", out_fifo);
        ret = A_E_INVAL;
        goto done;
    }
    *pthr = thr;

done:
    if (FAILED(ret) && NULL != thr) {
        if (-1 != thr->fifo_fd) {
            close(thr->fifo_fd);
        }
        if (-1 != thr->debug_signal_fd) {
            close(thr->debug_signal_fd);
        }
        TFREE(thr);
    }
    return ret;
}

aresult_t demod_thread_delete(struct demod_thread **pthr)
{
    TSL_ASSERT_ARG(NULL != pthr);
    TSL_ASSERT_ARG(NULL != *pthr);
    struct demod_thread *thr = *pthr;
    if (-1 != thr->fifo_fd) {
        close(thr->fifo_fd);
        thr->fifo_fd = -1;
    }
    if (-1 != thr->debug_signal_fd) {
        close(thr->debug_signal_fd);
        thr->debug_signal_fd = -1;
    }
    TFREE(thr);
    *pthr = NULL;
    return A_OK;
}

/* write one channel's share of a finished block: the FIFO byte stream of demod.c:93, EPIPE policy :95-110 */
static void _demod_thread_emit(struct demod_thread *dthr, const struct mfm_block *blk, size_t row)
{
    const size_t n = blk->nr_outputs;
    const int16_t *pcm = blk->pcm + row * blk->stride;

    dthr->total_nr_demod_samples += n;
    if (-1 != dthr->debug_signal_fd && NULL != blk->iq) {
        const int16_t *iq = blk->iq + row * blk->stride * 2;
        if (0 > write(dthr->debug_signal_fd, iq, n * 2 * sizeof(int16_t))) {
            int errnum = errno;
            MFM_MSG(SEV_WARNING, "CANT-WRITE-DEBUG-FILE", "Unable to write %zu bytes to post-demod debug file. "
                    "Reason: %s (%d). Skipping.", n * 2 * sizeof(int16_t), strerror(errnum), errnum);
        }
    }

    size_t done = 0;
    while (done < n) {
        ssize_t w = write(dthr->fifo_fd, pcm + done, (n - done) * sizeof(int16_t));
        if (w < 0) {
            int errnum = errno;
            if (errnum == EINTR) {
                continue;
            }
            if (errnum == EPIPE) {
                if (0 == dthr->nr_dropped_samples) {
                    MFM_MSG(SEV_WARNING, "FIFO-REMOTE-END-DISCONNECTED", "Remote end of FIFO disconnected. "
                            "Until a process picks up the FIFO, we're dropping samples.");
                }
                dthr->nr_dropped_samples += n - done;
                return;
            }
            PANIC("Failed to write %zu bytes to the output fifo. Reason: %s (%d)", (n - done) * sizeof(int16_t),
                  strerror(errnum), errnum);
        }
        done += (size_t)w / sizeof(int16_t);
    }
    dthr->total_nr_pcm_samples += n;
    if (0 != dthr->nr_dropped_samples) {
        MFM_MSG(SEV_WARNING, "FIFO-RESUMED", "Remote FIFO end reconnected. Dropped %zu samples in the interim.",
                dthr->nr_dropped_samples);
        dthr->nr_dropped_samples = 0;
    }
}

/* ---- receiver ---- */

/* What the front end's thread and receiver_cleanup() share beyond the life of the receiver structure.  The reference's
 * front ends free the structure that embeds their `struct receiver` in their cleanup function (multifm/rtl_sdr_if.c:59-82),
 * and for a blocking reader (librtlsdr) that function is also what makes the thread function return - so the thread may
 * come back from its front end after the memory is gone, and must not touch it then. */
struct mfm_rx_lifeline {
    pthread_mutex_t mu;
    bool rx_gone; /* receiver_cleanup() has handed the structure to the front end's cleanup function */
    int refs;     /* the receiver and, once started, its front-end thread */
};

static void _lifeline_drop(struct mfm_rx_lifeline *life)
{
    pthread_mutex_lock(&life->mu);
    const int left = --life->refs;
    pthread_mutex_unlock(&life->mu);
    if (0 == left) {
        pthread_mutex_destroy(&life->mu);
        free(life);
    }
}


aresult_t receiver_sample_buf_alloc(struct receiver *rx, struct sample_buf **pbuf)
{
    aresult_t ret = A_OK;
    struct sample_buf *sbuf = NULL;

    TSL_ASSERT_ARG(NULL != rx);
    TSL_ASSERT_ARG(NULL != pbuf);
    *pbuf = NULL;

    /* receiver_cleanup() takes the pool away under a front end whose reader it cannot stop first (librtlsdr's is only cancelled
     * by the front end's own cleanup function, which frees `rx` and so runs last): a buffer the front end holds is counted from
     * here to its delivery, and once the receiver is closing there are no more - the callback drops its samples as if the pool
     * were empty.  Count first, then look at the flag; cleanup sets the flag first, then looks at the count. */
    atomic_fetch_add(&rx->front_end_holds, 1);
    if (atomic_load(&rx->closing)) {
        atomic_fetch_sub(&rx->front_end_holds, 1);
        return A_E_BUSY;
    }
    /* pool exhausted: drop and count, log once (multifm/receiver.c:57-63) */
    if (FAILED(ret = frame_alloc(rx->samp_alloc, (void **)&sbuf))) {
        atomic_fetch_sub(&rx->front_end_holds, 1);
        if (0 == rx->nr_samp_buf_alloc_fails) {
            MFM_MSG(SEV_INFO, "NO-SAMPLE-BUFFER", "There are no available sample buffers, dropping received samples.");
        }
        rx->nr_samp_buf_alloc_fails++;
        return ret;
    }
    sbuf->release = _sample_buf_release_undelivered;
    sbuf->priv = rx;
    sbuf->refcount = 0;
    sbuf->sample_type = COMPLEX_INT_16;
    *pbuf = sbuf;
    return A_OK;
}

aresult_t receiver_sample_buf_deliver(struct receiver *rx, struct sample_buf *buf)
{
    TSL_ASSERT_ARG(NULL != rx);
    TSL_ASSERT_ARG(NULL != buf);
    TSL_BUG_ON(0 == buf->nr_samples); /* multifm/receiver.c:84 */

    const uint64_t t0 = tsl_get_clock_monotonic();
    /* One consumer - the submit thread - so refcount 1 (multifm/receiver.c:86 sets it to the number of channel
     * threads).  Nothing is copied and nothing is waited for here: the pointer goes into a ring with as many slots as
     * the pool has frames.  A front end that outruns the GPU finds the pool empty at its next
     * receiver_sample_buf_alloc() and drops there, counted, as the reference does. */
    atomic_store((_Atomic uint32_t *)&buf->refcount, 1);
    if (rx->failed) {
        (void)sample_buf_decref(buf); /* (undelivered: the release counts it off the front end's holds) */
        return A_E_DEVICE;
    }
    buf->release = _sample_buf_release;
    const size_t head = rx->ring_head;
    TSL_BUG_ON(head - rx->ring_tail >= rx->ring_slots); /* more buffers in flight than the pool holds */
    rx->ring[head % rx->ring_slots] = buf;
    rx->ring_head = head + 1; /* publishes the slot */
    rx->nr_bufs_delivered++;
    _bell_ring(&rx->ring_bell); /* wakes the submit thread if it sleeps; never waits */
    atomic_fetch_sub(&rx->front_end_holds, 1); /* the last touch of the receiver's ring and bell on this path */
    const uint64_t dt = tsl_get_clock_monotonic() - t0;
    if (dt > rx->max_deliver_ns) {
        rx->max_deliver_ns = dt;
    }
    return A_OK;
}

static int _format_of(const struct sample_buf *buf)
{
    switch (buf->sample_type) {
    case RAW_COMPLEX_INT_8: return MFM_IN_CS8;
    case RAW_COMPLEX_FILE_UINT_8: return MFM_IN_CU8;
    case RAW_COMPLEX_RTLSDR_UINT_8: return MFM_IN_RTLSDR_U8;
    default: return MFM_IN_CS16;
    }
}

/* buffers whose H2D copy is through go back to the pool (filter/direct_fir.c:395 is where a channel thread of the reference
 * lets go of a buffer: when its last sample has been consumed); wait = block until every one of them is back */
static void _receiver_reap_copies(struct receiver *rx, bool wait)
{
    while (rx->copy_tail != rx->copy_head) {
        const size_t i = rx->copy_tail % rx->ring_slots;
        int done = rx->failed ? 1 : mfm_group_copy_done(rx->group, rx->copy_ticket[i]);
        if (0 == done && wait) {
            done = MFM_OK == mfm_group_copy_wait(rx->group, rx->copy_ticket[i]) ? 1 : -1;
        }
        if (0 == done) {
            return;
        }
        if (done < 0 && !rx->failed) {
            MFM_MSG(SEV_FATAL, "ENGINE-COPY", "waiting for a buffer's copy failed: %s", mfm_last_error());
            rx->failed = 1;
        }
        TSL_BUG_IF_FAILED(sample_buf_decref(rx->copying[i])); /* back to the pool */
        rx->copy_tail++;
    }
}

/* the submit thread: ring -> device group; the only place that waits for the GPU on the input side */
static aresult_t _receiver_submit_thread(struct worker_thread *wthr)
{
    struct receiver *rx = BL_CONTAINER_OF(wthr, struct receiver, submit_thr);
    while (worker_thread_is_running(wthr) || rx->ring_tail != rx->ring_head) {
        const size_t tail = rx->ring_tail;
        if (rx->copy_head - rx->copy_tail >= (rx->ring_slots + 3) / 4) {
            /* a quarter of the pool is with the copy engine: ask for it back (asking costs an event on the copy stream, so
             * not after every buffer) */
            _receiver_reap_copies(rx, false);
        }
        if (tail == rx->ring_head) {
            _receiver_reap_copies(rx, true); /* nothing else to do: the front end gets its buffers back as soon as possible */
            /* the backlog is through: what the device group accepted without launching goes out now, as the reference's
             * channel thread runs what its queue held before it sleeps again (multifm/demod.c:134-150) */
            while (!rx->failed) {
                _bell_arm(&rx->room_bell);
                const int frc = mfm_group_flush(rx->group);
                if (MFM_E_BUSY == frc) {
                    _bell_sleep(&rx->room_bell, MFM_IDLE_MS);
                    continue;
                }
                _bell_disarm(&rx->room_bell);
                if (MFM_OK != frc) {
                    MFM_MSG(SEV_FATAL, "ENGINE-FLUSH", "mfm_group_flush failed: %s", mfm_last_error());
                    rx->failed = 1;
                }
                break;
            }
            _bell_ring(&rx->block_bell);
            _bell_ring(&rx->idle_bell); /* receiver_drain() waits for "nothing gathered and not launched" */
            _bell_arm(&rx->ring_bell);
            if (tail == rx->ring_head && worker_thread_is_running(wthr)) {
                _bell_sleep(&rx->ring_bell, MFM_IDLE_MS);
            } else {
                _bell_disarm(&rx->ring_bell);
            }
            continue;
        }
        struct sample_buf *buf = rx->ring[tail % rx->ring_slots]; /* the load of ring_head above acquired it */
        size_t run = 1;
        if (!rx->failed) {
            /* a backlog: how many of the buffers behind this one are its neighbours in the pool's slab (the pool hands frames out
             * in address order), with as many samples and the same format - they go to the device as one strided copy command */
            const size_t have = rx->ring_head - tail;
            while (run < have && run < MFM_RUN_MAX) {
                const struct sample_buf *nb = rx->ring[(tail + run) % rx->ring_slots];
                if ((const uint8_t *)nb != (const uint8_t *)buf + run * rx->frame_stride || nb->nr_samples != buf->nr_samples ||
                    nb->sample_type != buf->sample_type) {
                    break;
                }
                run++;
            }
            for (;;) {
                /* armed before the attempt: a slot released between a refused push and the sleep still rings */
                _bell_arm(&rx->room_bell);
                uint64_t ticket = 0;
                size_t took = 0;
                const int rc = mfm_group_push_pinned_run(rx->group, buf->data_buf, rx->frame_stride, buf->nr_samples, run, _format_of(buf),
                                                         &ticket, &took);
                if (MFM_OK == rc) {
                    for (size_t k = 0; k < took; k++) {
                        rx->copying[rx->copy_head % rx->ring_slots] = rx->ring[(tail + k) % rx->ring_slots];
                        rx->copy_ticket[rx->copy_head % rx->ring_slots] = ticket;
                        rx->copy_head++;
                    }
                    run = took;
                    buf = NULL; /* the copy engine is reading them: _receiver_reap_copies() returns them to the pool */
                }
                if (MFM_E_BUSY == rc && !rx->failed) {
                    /* every output slot holds a block the drain thread has not written out yet */
                    _bell_sleep(&rx->room_bell, MFM_IDLE_MS);
                    continue;
                }
                _bell_disarm(&rx->room_bell);
                if (MFM_OK == rc) {
                    break;
                }
                MFM_MSG(SEV_FATAL, "ENGINE-PUSH", "mfm_group_push failed: %s", mfm_last_error());
                rx->failed = 1;
                break;
            }
        }
        if (NULL != buf) {
            run = 1;
            TSL_BUG_IF_FAILED(sample_buf_decref(buf)); /* not handed over (a failed receiver): back to the pool */
        }
        rx->ring_tail = tail + run;
        rx->nr_bufs_submitted += run;
        rx->nr_copy_commands++;
        _bell_ring(&rx->block_bell);
        _bell_ring(&rx->idle_bell);
    }
    _receiver_reap_copies(rx, true);
    return rx->failed ? A_E_DEVICE : A_OK;
}

static aresult_t _receiver_drain_once(struct receiver *rx, bool *got)
{
    struct mfm_block blks[MFM_GROUP_MAX_DEVICES];
    struct demod_thread *dthr = NULL;

    *got = false;
    int rc = mfm_group_fetch(rx->group, blks);
    if (MFM_E_DONE == rc) {
        return A_OK;
    }
    if (MFM_OK != rc) {
        MFM_MSG(SEV_FATAL, "ENGINE-FETCH", "mfm_group_fetch failed: %s", mfm_last_error());
        return A_E_DEVICE;
    }
    if (!rx->muted) {
        list_for_each_type(dthr, &rx->demod_threads, dt_node) {
            /* which shard holds this channel, and which row of the shard's block it is: looked up once, at receiver_start() */
            _demod_thread_emit(dthr, &blks[dthr->shard], dthr->shard_row);
        }
    }
    rx->nr_blocks_drained++;
    *got = true;
    if (MFM_OK != mfm_group_release(rx->group)) {
        return A_E_DEVICE;
    }
    _bell_ring(&rx->room_bell);
    _bell_ring(&rx->idle_bell);
    return A_OK;
}

static aresult_t _receiver_drain_thread(struct worker_thread *wthr)
{
    struct receiver *rx = BL_CONTAINER_OF(wthr, struct receiver, drain_thr);
    while (worker_thread_is_running(wthr)) {
        bool got = false;
        if (FAILED(_receiver_drain_once(rx, &got))) {
            rx->failed = 1; /* deliver(), the submit thread and receiver_drain() stop waiting for us */
            _bell_ring(&rx->idle_bell);
            return A_E_DEVICE;
        }
        if (!got) {
            /* nothing finished or in flight: sleep until the submit thread has pushed a buffer */
            _bell_arm(&rx->block_bell);
            if (FAILED(_receiver_drain_once(rx, &got))) {
                rx->failed = 1;
                _bell_ring(&rx->idle_bell);
                return A_E_DEVICE;
            }
            if (!got && worker_thread_is_running(wthr)) {
                _bell_sleep(&rx->block_bell, MFM_IDLE_MS);
            } else {
                _bell_disarm(&rx->block_bell);
            }
        }
    }
    return A_OK;
}

/* pending work of the device group: samples accepted and not launched, blocks launched and not written out */
static aresult_t _receiver_pending(struct receiver *rx, bool *samples, bool *blocks)
{
    *samples = *blocks = false;
    for (int s = 0; s < rx->nr_shards; s++) {
        struct mfm_stats st;
        if (MFM_OK != mfm_group_get_stats(rx->group, (uint32_t)s, &st)) {
            return A_E_DEVICE;
        }
        *samples = *samples || 0 != st.pending_samples;
        *blocks = *blocks || 0 != st.pending_blocks;
    }
    return A_OK;
}

aresult_t receiver_drain(struct receiver *rx)
{
    TSL_ASSERT_ARG(NULL != rx);
    /* everything delivered has been submitted, the devices are done, every finished block has been written out */
    while (rx->submit_thr.started && rx->nr_bufs_submitted != rx->nr_bufs_delivered) {
        if (rx->failed) {
            return A_E_DEVICE;
        }
        _bell_arm(&rx->idle_bell);
        if (rx->nr_bufs_submitted != rx->nr_bufs_delivered && !rx->failed) {
            _bell_sleep(&rx->idle_bell, MFM_IDLE_MS);
        } else {
            _bell_disarm(&rx->idle_bell);
        }
    }
    if (rx->failed) {
        return A_E_DEVICE;
    }
    for (;;) {
        bool samples = false, blocks = false;
        if (FAILED(_receiver_pending(rx, &samples, &blocks))) {
            return A_E_DEVICE;
        }
        if (!samples && !blocks) {
            return A_OK;
        }
        if (rx->failed) {
            return A_E_DEVICE;
        }
        if (samples && !rx->submit_thr.started) {
            /* no submit thread (a caller that pushes by itself): launch what was gathered from here.  MFM_E_BUSY = every
             * output slot holds an unfetched block: not an error, it goes away as blocks are written out below. */
            const int frc = mfm_group_flush(rx->group);
            if (MFM_OK != frc && MFM_E_BUSY != frc) {
                return A_E_DEVICE;
            }
        } else if (samples) {
            /* The submit thread is the group's one producer: it launches what it gathered when its ring runs empty
             * (_receiver_submit_thread) and waits there for an output slot if it has to.  Flushing from this thread as well
             * would run two producer-side calls at once, and a full output ring (MFM_E_BUSY) would look like a device
             * failure and cut the end of the stream. */
            _bell_ring(&rx->ring_bell);
        }
        if (rx->drain_thr.started) {
            /* the drain thread owns fetch/release and rings after every block */
            _bell_arm(&rx->idle_bell);
            _bell_sleep(&rx->idle_bell, 5u);
        } else {
            bool got = false;
            if (FAILED(_receiver_drain_once(rx, &got))) {
                return A_E_DEVICE;
            }
        }
    }
}

void receiver_mark_input_done(struct receiver *rx)
{
    rx->input_done = true;
}

/* what receiver_init() reads from the configuration before it builds anything (keys, defaults and messages of
 * multifm/receiver.c:133-184; the MI355X keys gpuDevice / gpuDevices / gpuExchange are this build's) */
struct receiver_settings {
    int nr_samp_bufs, sample_rate, center_freq, decimation;
    double *lpf_taps;
    size_t lpf_nr_taps;
    struct config channels;
    struct mfm_group_config group;
};

static aresult_t _settings_devices(struct config *cfg, struct mfm_group_config *gc)
{
    struct config devs = CONFIG_INIT_EMPTY, dev = CONFIG_INIT_EMPTY;
    const char *xchg = NULL;
    aresult_t it = A_OK;
    size_t ctr = 0;
    int gpu = 0;

    memset(gc, 0, sizeof(*gc));
    gc->abi_version = MFM_ABI_VERSION;
    /* "gpuDevices": [d0, d1, ...] shards the channels over several GPUs of the node (d0 ingests and is the root of the
     * exchange); "gpuDevice": d is the one-GPU form (default device 0) */
    if (!FAILED(config_get(cfg, &devs, "gpuDevices"))) {
        CONFIG_ARRAY_FOR_EACH(dev, &devs, it, ctr) {
            int d = -1;
            if (gc->nr_devices >= MFM_GROUP_MAX_DEVICES || FAILED(config_get_integer(&dev, &d, NULL)) || d < 0) {
                MFM_MSG(SEV_ERROR, "BAD-GPU-DEVICES", "'gpuDevices' must be an array of at most %d device numbers.",
                        MFM_GROUP_MAX_DEVICES);
                return A_E_INVAL;
            }
            gc->devices[gc->nr_devices++] = d;
        }
        (void)it;
    }
    {
        /* test aid: "gpuTestSharedDevice": true lets 'gpuDevices' list a device several times (MFM_F_GROUP_SHARED_DEVICE),
         * which is how the multi-shard paths of this file are run on a one-GPU box against a test double of RCCL */
        bool shared = false;
        if (!FAILED(config_get_boolean(cfg, &shared, "gpuTestSharedDevice")) && shared) {
            gc->flags |= MFM_F_GROUP_SHARED_DEVICE;
        }
    }
    if (0 == gc->nr_devices) {
        (void)config_get_integer(cfg, &gpu, "gpuDevice");
        gc->devices[0] = gpu;
        gc->nr_devices = 1;
    }
    /* "gpuExchange": "rccl" sends the blocks through the RCCL broadcast path even on one device, "allgather" through the
     * scatter + all-gather form (include/multifm_hip.h, MFM_X_*) */
    if (!FAILED(config_get_string(cfg, &xchg, "gpuExchange"))) {
        gc->exchange = 0 == strcmp(xchg, "rccl") ? MFM_X_RCCL : 0 == strcmp(xchg, "allgather") ? MFM_X_RCCL_ALLGATHER : MFM_X_AUTO;
    }
    return A_OK;
}

static aresult_t _settings_read(struct config *cfg, size_t samples_per_buf, struct receiver_settings *st)
{
    memset(st, 0, sizeof(*st));
    if (FAILED(config_get_integer(cfg, &st->nr_samp_bufs, "nrSampBufs"))) {
        MFM_MSG(SEV_INFO, "DEFAULT-SAMP-BUFS", "Setting sample buffer count to 64");
        st->nr_samp_bufs = 64;
    }
    if (FAILED(config_get_integer(cfg, &st->sample_rate, "sampleRateHz"))) {
        MFM_MSG(SEV_INFO, "NO-SAMPLE-RATE", "Need to specify a sample rate, in Hertz.");
        return A_E_INVAL;
    }
    if (FAILED(config_get_integer(cfg, &st->center_freq, "centerFreqHz"))) {
        MFM_MSG(SEV_INFO, "NO-CENTER-FREQ", "You forgot to specify a center frequency, in Hz.");
        return A_E_INVAL;
    }
    MFM_MSG(SEV_INFO, "SAMPLE-RATE", "Sample rate is set to %u Hz", st->sample_rate);
    MFM_MSG(SEV_INFO, "CENTER-FREQ", "Center Frequency is %u Hz", st->center_freq);
    if (FAILED(config_get_integer(cfg, &st->decimation, "decimationFactor"))) {
        MFM_MSG(SEV_INFO, "NO-DECIMATION", "Not decimating the output signal: using full bandwidth.");
        return A_E_INVAL;
    }
    if (0 >= st->decimation) {
        MFM_MSG(SEV_ERROR, "BAD-DECIMATION-FACTOR", "Decimation factor of '%d' is not valid.", st->decimation);
        return A_E_INVAL;
    }
    if (FAILED(config_get_float_array(cfg, &st->lpf_taps, &st->lpf_nr_taps, "lpfTaps"))) {
        MFM_MSG(SEV_ERROR, "BAD-FILTER-TAPS", "Need to provide a baseband filter with at least two filter taps as 'lpfTaps'.");
        return A_E_INVAL;
    }
    if (1 >= st->lpf_nr_taps) {
        MFM_MSG(SEV_ERROR, "INSUFF-FILTER-TAPS", "Not enough filter taps for the low-pass filter.");
        return A_E_INVAL;
    }
    if (FAILED(config_get(cfg, &st->channels, "channels"))) {
        MFM_MSG(SEV_ERROR, "MISSING-CHANNELS", "Need to specify at least one channel to demodulate.");
        return A_E_INVAL;
    }
    aresult_t ret = _settings_devices(cfg, &st->group);
    st->group.sample_rate_hz = (uint32_t)st->sample_rate;
    st->group.decimation = (uint32_t)st->decimation;
    st->group.max_block_samples = (uint32_t)samples_per_buf;
    {
        /* A channel thread of the reference runs whatever its queue holds - up to 128 sample_bufs, multifm/demod.c:297 - back
         * to back.  The device group does that with launches: up to a pool's worth of delivered buffers (all that can be
         * outstanding at once) share one, "gpuCoalesceSamples" overrides (0: one launch per buffer). */
        int co = -1;
        uint64_t want = (uint64_t)st->nr_samp_bufs * samples_per_buf;
        if (!FAILED(config_get_integer(cfg, &co, "gpuCoalesceSamples")) && co >= 0) {
            want = (uint64_t)co;
        }
        st->group.coalesce_samples = (uint32_t)(want > (1u << 26) ? (1u << 26) : want);
        if (st->group.coalesce_samples <= samples_per_buf) {
            st->group.coalesce_samples = 0; /* nothing to gather */
        }
    }
    return ret;
}

/* one "channels" entry -> one demod_thread on the bound group (keys and messages of multifm/receiver.c:195-244) */
static aresult_t _receiver_add_channel(struct receiver *rx, struct config *channel, const struct receiver_settings *st)
{
    const char *fifo_name = NULL, *signal_debug = NULL;
    int nb_center_freq = -1;
    struct demod_thread *dmt = NULL;
    double gain = 1.0, gain_db = 0.0;

    if (FAILED(config_get_string(channel, &fifo_name, "outFifo"))) {
        MFM_MSG(SEV_ERROR, "MISSING-FIFO-ID", "Missing output FIFO filename, aborting.");
        return A_E_INVAL;
    }
    if (FAILED(config_get_integer(channel, &nb_center_freq, "chanCenterFreq"))) {
        MFM_MSG(SEV_ERROR, "MISSING-CENTER-FREQ", "Missing output channel center frequency.");
        return A_E_INVAL;
    }
    if (!FAILED(config_get_string(channel, &signal_debug, "signalDebugFile"))) {
        MFM_MSG(SEV_INFO, "WRITING-SIGNAL-DEBUG", "The channel at frequency %d will have raw I/Q written to '%s'",
                nb_center_freq, signal_debug);
    }
    /* the key is case sensitive ("dbGain" in etc/pocsag_rtlsdr.json:19 is silently ignored) and the conversion is
     * 10^(dB/10) applied to amplitude taps (multifm/receiver.c:218-220) */
    if (!FAILED(config_get_float(channel, &gain_db, "dBGain"))) {
        gain = pow(10.0, gain_db / 10.0);
    }
    if (FAILED(demod_thread_new(&dmt, (unsigned)-1, (int32_t)nb_center_freq - st->center_freq, (uint32_t)st->sample_rate,
                                fifo_name, st->decimation, st->lpf_taps, st->lpf_nr_taps, signal_debug, gain))) {
        MFM_MSG(SEV_ERROR, "FAILED-DEMOD-THREAD", "Failed to create demodulator thread, aborting.");
        return A_E_INVAL;
    }
    list_append(&rx->demod_threads, &dmt->dt_node);
    rx->nr_demod_threads++;
    MFM_MSG(SEV_INFO, "CHANNEL", "[%zu]: %4.5f MHz Gain: %f dB -> [%s]%s%s", rx->nr_demod_threads, (double)nb_center_freq / 1e6,
            gain_db, fifo_name, (NULL != signal_debug ? " DEBUG: " : ""), (NULL != signal_debug ? signal_debug : ""));
    return A_OK;
}

static void _receiver_zero(struct receiver *rx, receiver_rx_thread_func_t rx_func, receiver_cleanup_func_t cleanup_func)
{
    memset(&rx->wthr, 0, sizeof(rx->wthr));
    memset(&rx->submit_thr, 0, sizeof(rx->submit_thr));
    memset(&rx->drain_thr, 0, sizeof(rx->drain_thr));
    rx->muted = true;
    rx->samp_alloc = NULL;
    rx->cleanup_func = cleanup_func;
    rx->thread_func = rx_func;
    rx->group = NULL;
    rx->nr_shards = 0;
    rx->ring = NULL;
    rx->ring_slots = 0;
    rx->copying = NULL;
    rx->copy_ticket = NULL;
    rx->copy_head = rx->copy_tail = 0;
    rx->nr_demod_threads = 0;
    rx->life = calloc(1, sizeof(*rx->life));
    TSL_BUG_ON(NULL == rx->life);
    pthread_mutex_init(&rx->life->mu, NULL);
    rx->life->refs = 1;
    atomic_store(&rx->nr_samp_buf_alloc_fails, 0);
    atomic_store(&rx->closing, 0);
    atomic_store(&rx->front_end_holds, 0);
    atomic_store(&rx->input_done, false);
    atomic_store(&rx->nr_blocks_drained, 0);
    atomic_store(&rx->ring_head, 0);
    atomic_store(&rx->ring_tail, 0);
    atomic_store(&rx->failed, 0);
    atomic_store(&rx->nr_bufs_delivered, 0);
    atomic_store(&rx->nr_bufs_submitted, 0);
    atomic_store(&rx->nr_copy_commands, 0);
    rx->frame_stride = 0;
    atomic_store(&rx->max_deliver_ns, 0);
    list_init(&rx->demod_threads);
    _bell_init(&rx->ring_bell);
    _bell_init(&rx->room_bell);
    _bell_init(&rx->block_bell);
    _bell_init(&rx->idle_bell);
}

/* everything receiver_init() built inside `rx`, except the lifeline */
static void _receiver_teardown(struct receiver *rx)
{
    struct demod_thread *cur = NULL, *tmp = NULL;
    list_for_each_type_safe(cur, tmp, &rx->demod_threads, dt_node) {
        list_del(&cur->dt_node);
        TSL_BUG_IF_FAILED(demod_thread_delete(&cur));
    }
    rx->nr_demod_threads = 0;
    mfm_group_destroy(&rx->group);
    if (NULL != rx->ring) {
        TFREE(rx->ring);
    }
    if (NULL != rx->copying) {
        TFREE(rx->copying);
    }
    if (NULL != rx->copy_ticket) {
        TFREE(rx->copy_ticket);
    }
    TSL_BUG_IF_FAILED(frame_alloc_delete(&rx->samp_alloc));
    sem_destroy(&rx->ring_bell.sem);
    sem_destroy(&rx->room_bell.sem);
    sem_destroy(&rx->block_bell.sem);
    sem_destroy(&rx->idle_bell.sem);
}

aresult_t receiver_init(struct receiver *rx, struct config *cfg, receiver_rx_thread_func_t rx_func,
                        receiver_cleanup_func_t cleanup_func, size_t samples_per_buf)
{
    struct receiver_settings st;
    struct config channel = CONFIG_INIT_EMPTY;
    aresult_t ret = A_OK, it = A_OK;
    size_t ctr = 0;

    TSL_ASSERT_ARG(NULL != rx);
    TSL_ASSERT_ARG(NULL != cfg);
    TSL_ASSERT_ARG(NULL != rx_func);
    TSL_ASSERT_ARG(NULL != cleanup_func);
    TSL_ASSERT_ARG(0 != samples_per_buf);

    _receiver_zero(rx, rx_func, cleanup_func);
    if (FAILED(ret = _settings_read(cfg, samples_per_buf, &st))) {
        goto done;
    }
    /* the pool of multifm/receiver.c:154-157, in page-locked memory: the H2D copy reads a delivered buffer's data_buf where
     * the front end wrote it, no staging copy in between */
    if (FAILED(frame_alloc_new_on(&rx->samp_alloc, sizeof(struct sample_buf) + samples_per_buf * sizeof(int16_t) * 2,
                                  (size_t)st.nr_samp_bufs, mfm_host_alloc, mfm_host_free))) {
        MFM_MSG(SEV_FATAL, "NO-PINNED-POOL", "Unable to allocate %d page-locked sample buffers of %zu samples (no usable GPU?).",
                st.nr_samp_bufs, samples_per_buf);
        ret = A_E_NOMEM;
        goto done;
    }

    /* one device group for the whole channel set, and a pointer ring with a slot per pool frame in front of it */
    if (MFM_OK != mfm_group_create(&rx->group, &st.group)) {
        MFM_MSG(SEV_FATAL, "ENGINE-CREATE", "Unable to create the channel engine: %s", mfm_last_error());
        ret = A_E_DEVICE;
        goto done;
    }
    MFM_MSG(SEV_INFO, "GPU-DEVICES", "Channels are sharded over %u GPU(s), first device %d%s", st.group.nr_devices,
            st.group.devices[0], st.group.exchange == MFM_X_RCCL ? " (RCCL exchange forced)" :
            st.group.exchange == MFM_X_RCCL_ALLGATHER ? " (RCCL scatter + all-gather exchange)" : "");
    rx->ring_slots = (size_t)st.nr_samp_bufs;
    rx->frame_stride = frame_alloc_frame_bytes(rx->samp_alloc);
    if (FAILED(ret = TACALLOC(&rx->ring, rx->ring_slots, sizeof(*rx->ring), SYS_CACHE_LINE_LENGTH)) ||
        FAILED(ret = TACALLOC(&rx->copying, rx->ring_slots, sizeof(*rx->copying), SYS_CACHE_LINE_LENGTH)) ||
        FAILED(ret = TACALLOC(&rx->copy_ticket, rx->ring_slots, sizeof(*rx->copy_ticket), SYS_CACHE_LINE_LENGTH))) {
        goto done;
    }

    demod_thread_bind_group(rx->group);
    CONFIG_ARRAY_FOR_EACH(channel, &st.channels, it, ctr) {
        if (FAILED(ret = _receiver_add_channel(rx, &channel, &st))) {
            break;
        }
    }
    demod_thread_bind_group(NULL);
    if (FAILED(ret) || FAILED(it)) {
        MFM_MSG(SEV_ERROR, "CHANNEL-SETUP-FAILURE", "Error reading array of channels, aborting.");
        ret = FAILED(ret) ? ret : it;
    } else if (0 == rx->nr_demod_threads) {
        MFM_MSG(SEV_ERROR, "MISSING-CHANNELS", "Need to specify at least one channel to demodulate.");
        ret = A_E_INVAL;
    }

done:
    if (NULL != st.lpf_taps) {
        TFREE(st.lpf_taps);
    }
    if (FAILED(ret)) {
        /* the caller frees its own structure and never calls receiver_cleanup() (multifm/rtl_sdr_if.c:463-477) */
        _receiver_teardown(rx);
        _lifeline_drop(rx->life);
        rx->life = NULL;
    }
    return ret;
}

static aresult_t _receiver_worker_thread(struct worker_thread *wthr)
{
    struct receiver *rx = BL_CONTAINER_OF(wthr, struct receiver, wthr);
    struct mfm_rx_lifeline *life = rx->life; /* referenced for this thread by receiver_start() */
    TSL_BUG_ON(NULL == rx->thread_func);
    const aresult_t ret = rx->thread_func(rx);
    /* a front end whose thread function has returned delivers nothing further (the reference's file front end leaves its
     * loop when a read comes back short, multifm/file_if.c:181-184): a driver may stop waiting for input */
    pthread_mutex_lock(&life->mu);
    if (!life->rx_gone) {
        receiver_mark_input_done(rx);
        wthr->running = false;
    }
    pthread_mutex_unlock(&life->mu);
    _lifeline_drop(life);
    return ret;
}

aresult_t receiver_start(struct receiver *rx)
{
    aresult_t ret = A_OK;
    TSL_ASSERT_ARG(NULL != rx);

    /* the channel set is complete: build the tables and go to the device */
    if (MFM_OK != mfm_group_commit(rx->group)) {
        MFM_MSG(SEV_ERROR, "ENGINE-COMMIT", "Unable to start the channel engine: %s", mfm_last_error());
        return A_E_DEVICE;
    }
    rx->nr_shards = mfm_group_nr_shards(rx->group);
    {
        /* channel -> (shard, row of the shard's block): fixed once the group is committed */
        struct demod_thread *dthr = NULL;
        list_for_each_type(dthr, &rx->demod_threads, dt_node) {
            bool found = false;
            for (int s = 0; s < rx->nr_shards && !found; s++) {
                uint32_t first = 0, count = 0;
                if (MFM_OK == mfm_group_shard_info(rx->group, (uint32_t)s, &first, &count, NULL) &&
                    (uint32_t)dthr->chan_index >= first && (uint32_t)dthr->chan_index < first + count) {
                    dthr->shard = s;
                    dthr->shard_row = (size_t)dthr->chan_index - first;
                    found = true;
                }
            }
            if (!found) {
                MFM_MSG(SEV_ERROR, "ENGINE-COMMIT", "Channel %d is on no shard of the device group.", dthr->chan_index);
                return A_E_DEVICE;
            }
        }
    }
    if (FAILED(ret = worker_thread_new(&rx->drain_thr, _receiver_drain_thread, WORKER_THREAD_CPU_MASK_ANY))) {
        return ret;
    }
    if (FAILED(ret = worker_thread_new(&rx->submit_thr, _receiver_submit_thread, WORKER_THREAD_CPU_MASK_ANY))) {
        return ret;
    }
    pthread_mutex_lock(&rx->life->mu);
    rx->life->refs++; /* the front end's thread */
    pthread_mutex_unlock(&rx->life->mu);
    if (FAILED(ret = worker_thread_new(&rx->wthr, _receiver_worker_thread, WORKER_THREAD_CPU_MASK_ANY))) {
        MFM_MSG(SEV_ERROR, "THREAD-START-FAIL", "Failed to start worker thread, aborting.");
        _lifeline_drop(rx->life);
    }
    return ret;
}

/* Wait up to `ms` for the front end's thread to end by itself.  true = joined. */
static bool _receiver_join_front_end(struct receiver *rx, unsigned ms)
{
    struct timespec ts;
    if (!rx->wthr.started) {
        return true;
    }
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_nsec += (long)(ms % 1000u) * 1000000L;
    ts.tv_sec += (time_t)(ms / 1000u) + ts.tv_nsec / 1000000000L;
    ts.tv_nsec %= 1000000000L;
    if (0 != pthread_timedjoin_np(rx->wthr.thr, NULL, &ts)) {
        return false;
    }
    rx->wthr.started = false;
    return true;
}

/*
 * The reference's order (multifm/receiver.c:281-315) is: the front end's cleanup function, then stop and join the front
 * end's thread, then the channel threads, then the pool.  Two properties of the reference's front ends shape the order
 * here: their cleanup function frees the structure that embeds `*rx` (rtl_sdr_if.c:81, so nothing of `rx` may be touched
 * after it), and for librtlsdr it is also what ends the blocking reader (rtlsdr_cancel_async, :70), so the thread cannot be
 * joined before it.  So: ask the thread to stop and give it a moment (a file reader or any polling front end ends by
 * itself: what it delivered then reaches the FIFOs completely); a reader that is still blocked is muted instead (its
 * callback delivers nothing when muted, rtl_sdr_if.c:94-98); drain and tear down everything that lives in `rx`; the front
 * end's cleanup function comes LAST, and the thread - by then on its way out - is joined through a copy of its handle.
 */
aresult_t receiver_cleanup(struct receiver **prx)
{
    struct receiver *rx = NULL;

    TSL_ASSERT_ARG(NULL != prx);
    TSL_ASSERT_ARG(NULL != *prx);
    rx = *prx;

    bool leak = false;
    TSL_BUG_IF_FAILED(worker_thread_request_shutdown(&rx->wthr));
    const bool joined = _receiver_join_front_end(rx, 500u);
    if (!joined) {
        /* A reader this function cannot stop (see above): no new buffers from here on, and wait for the one a callback may be
         * holding - it is in receiver_sample_buf_alloc(), filling the buffer, or in receiver_sample_buf_deliver() - to be
         * delivered.  A callback that was past its muted check but not yet in receiver_sample_buf_alloc() gets A_E_BUSY there and
         * touches nothing of what is torn down below. */
        rx->muted = true;
        atomic_store(&rx->closing, 1);
        unsigned waited_ms = 0;
        while (atomic_load(&rx->front_end_holds) > 0 && waited_ms < 5000u) {
            usleep(1000);
            waited_ms++;
        }
        if (atomic_load(&rx->front_end_holds) > 0) {
            /* a callback stuck for seconds with a buffer in its hands (a write to iqDumpFile that does not return): the pool, the
             * ring and the device group stay behind rather than being freed under it */
            MFM_MSG(SEV_WARNING, "FRONT-END-STUCK", "The front end still holds a sample buffer after %u ms; leaving its pool behind.", waited_ms);
            leak = true;
        }
    }
    if (NULL != rx->group && rx->drain_thr.started) {
        (void)receiver_drain(rx); /* returns A_E_DEVICE instead of waiting for a thread that has given up */
        if (rx->submit_thr.started) {
            TSL_BUG_IF_FAILED(worker_thread_request_shutdown(&rx->submit_thr));
            _bell_ring(&rx->ring_bell);
            _bell_ring(&rx->room_bell);
            TSL_BUG_IF_FAILED(worker_thread_delete(&rx->submit_thr));
        }
        TSL_BUG_IF_FAILED(worker_thread_request_shutdown(&rx->drain_thr));
        _bell_ring(&rx->block_bell);
        TSL_BUG_IF_FAILED(worker_thread_delete(&rx->drain_thr));
    }

    if (rx->nr_bufs_submitted) {
        MFM_MSG(SEV_INFO, "INGEST-SUMMARY", "%zu sample buffers delivered, %zu submitted in %zu copy commands, %zu requests found the pool empty, "
                "%zu blocks written out", (size_t)rx->nr_bufs_delivered, (size_t)rx->nr_bufs_submitted, (size_t)rx->nr_copy_commands,
                (size_t)rx->nr_samp_buf_alloc_fails, (size_t)rx->nr_blocks_drained);
    }
    if (!leak) {
        _receiver_teardown(rx);
    }

    /* from here on `rx` belongs to the front end */
    struct mfm_rx_lifeline *life = rx->life;
    const bool join_after = rx->wthr.started;
    const pthread_t front_end = rx->wthr.thr;
    const receiver_cleanup_func_t cleanup = rx->cleanup_func;
    pthread_mutex_lock(&life->mu);
    life->rx_gone = true;
    pthread_mutex_unlock(&life->mu);
    *prx = NULL;
    TSL_BUG_IF_FAILED(cleanup(rx));
    if (join_after) {
        /* a thread that is still stuck in its front end (a read on a pipe nobody writes to) is left behind, not waited for */
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        ts.tv_sec += 2;
        if (0 != pthread_timedjoin_np(front_end, NULL, &ts)) {
            pthread_detach(front_end);
        }
    }
    _lifeline_drop(life);
    return A_OK;
}

aresult_t receiver_set_mute(struct receiver *rx, bool mute)
{
    TSL_ASSERT_ARG(NULL != rx);
    rx->muted = mute;
    return A_OK;
}

bool receiver_thread_running(struct receiver *rx)
{
    TSL_BUG_ON(NULL == rx);
    return worker_thread_is_running(&rx->wthr);
}
