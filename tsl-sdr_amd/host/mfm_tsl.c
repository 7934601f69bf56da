/* mfm_tsl.c - worker threads, frame pool and clock behind mfm_tsl.h. */
#include "mfm_tsl.h"

#include <signal.h>
#include <time.h>

/* ---- worker thread ---- */

static void *worker_trampoline(void *arg)
{
    struct worker_thread *thr = arg;
    (void)thr->fn(thr);
    return NULL;
}

aresult_t worker_thread_new(struct worker_thread *thr, worker_thread_func_t fn, unsigned cpu)
{
    TSL_ASSERT_ARG(NULL != thr);
    TSL_ASSERT_ARG(NULL != fn);
    (void)cpu; /* no pinning: WORKER_THREAD_CPU_MASK_ANY is the only mask multifm passes for receivers */
    thr->fn = fn;
    thr->running = true;
    if (0 != pthread_create(&thr->thr, NULL, worker_trampoline, thr)) {
        thr->running = false;
        return A_E_INVAL;
    }
    thr->started = true;
    return A_OK;
}

aresult_t worker_thread_request_shutdown(struct worker_thread *thr)
{
    TSL_ASSERT_ARG(NULL != thr);
    thr->running = false;
    return A_OK;
}

aresult_t worker_thread_delete(struct worker_thread *thr)
{
    TSL_ASSERT_ARG(NULL != thr);
    if (thr->started) {
        thr->running = false;
        pthread_join(thr->thr, NULL);
        thr->started = false;
    }
    return A_OK;
}

/* ---- work queue ---- */

aresult_t work_queue_new(struct work_queue *wq, unsigned depth)
{
    TSL_ASSERT_ARG(NULL != wq);
    TSL_ASSERT_ARG(0 != depth);
    wq->slots = calloc(depth, sizeof(void *));
    if (NULL == wq->slots) {
        return A_E_NOMEM;
    }
    wq->depth = depth;
    wq->head = wq->count = 0;
    return A_OK;
}

aresult_t work_queue_release(struct work_queue *wq)
{
    TSL_ASSERT_ARG(NULL != wq);
    free(wq->slots);
    wq->slots = NULL;
    wq->depth = wq->head = wq->count = 0;
    return A_OK;
}

aresult_t work_queue_push(struct work_queue *wq, void *value)
{
    TSL_ASSERT_ARG(NULL != wq && NULL != wq->slots);
    if (wq->count == wq->depth) {
        return A_E_BUSY;
    }
    wq->slots[(wq->head + wq->count) % wq->depth] = value;
    wq->count++;
    return A_OK;
}

aresult_t work_queue_pop(struct work_queue *wq, void **value)
{
    TSL_ASSERT_ARG(NULL != wq && NULL != wq->slots);
    TSL_ASSERT_ARG(NULL != value);
    *value = NULL;
    if (wq->count) {
        *value = wq->slots[wq->head];
        wq->head = (wq->head + 1) % wq->depth;
        wq->count--;
    }
    return A_OK;
}

aresult_t work_queue_size(struct work_queue *wq, unsigned *count)
{
    TSL_ASSERT_ARG(NULL != wq);
    TSL_ASSERT_ARG(NULL != count);
    *count = wq->count;
    return A_OK;
}

/* ---- application scaffolding ---- */

static volatile sig_atomic_t app_stop_requested;
static app_sigint_handler_t app_user_handler;

static void app_on_sigint(int sig)
{
    (void)sig;
    app_stop_requested = 1;
    if (NULL != app_user_handler) {
        app_user_handler();
    }
}

/* The reference's app_init() daemonises / logs according to an optional "app" stanza; multifm and decoder pass a name and
 * (possibly NULL) configuration and use nothing of it afterwards.  Here: a broken pipe is an errno for the writer
 * (EPIPE is handled per channel, multifm/demod.c:95-105), not a signal that ends the process. */
aresult_t app_init(const char *app_name, struct config *cfg)
{
    TSL_ASSERT_ARG(NULL != app_name && '\0' != *app_name);
    (void)cfg;
    app_stop_requested = 0;
    signal(SIGPIPE, SIG_IGN);
    return A_OK;
}

aresult_t app_sigint_catch(app_sigint_handler_t handler)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    app_user_handler = handler;
    sa.sa_handler = app_on_sigint;
    sigemptyset(&sa.sa_mask);
    if (0 != sigaction(SIGINT, &sa, NULL) || 0 != sigaction(SIGTERM, &sa, NULL)) {
        return A_E_INVAL;
    }
    return A_OK;
}

int app_running(void)
{
    return !app_stop_requested;
}

void app_request_stop(void)
{
    app_stop_requested = 1;
}

/* ---- frame pool: nr_frames equally sized, 64-byte aligned frames of one slab on a lock-protected free QUEUE.  First in,
 *      first out: frames are handed out in the order they came back, which - with one producer and buffers that return in
 *      the order they were delivered - is address order, round and round the slab.  Buffers delivered one after the other
 *      are then neighbours in memory, and the receiver copies runs of them to the device with one strided command
 *      (mfm_group_push_pinned_run). ---- */

struct frame_alloc {
    uint8_t *slab;
    void (*slab_free)(void *);
    void **free_stack; /* ring of nr_frames slots: [head, head + nr_free) hold free frames */
    size_t frame_bytes, nr_frames, nr_free, head;
    pthread_mutex_t lock;
};

static void *_slab_malloc(size_t bytes)
{
    void *p = NULL;
    return 0 == posix_memalign(&p, 64, bytes) ? p : NULL;
}

aresult_t frame_alloc_new(struct frame_alloc **pfa, size_t frame_bytes, size_t nr_frames)
{
    return frame_alloc_new_on(pfa, frame_bytes, nr_frames, _slab_malloc, free);
}

/* the same pool on memory from the caller's allocator: the receiver puts its sample_bufs into page-locked memory
 * (mfm_host_alloc) so that the H2D copy reads data_buf where the front end wrote it */
aresult_t frame_alloc_new_on(struct frame_alloc **pfa, size_t frame_bytes, size_t nr_frames, void *(*slab_alloc)(size_t),
                             void (*slab_free)(void *))
{
    TSL_ASSERT_ARG(NULL != slab_alloc && NULL != slab_free);
    TSL_ASSERT_ARG(NULL != pfa);
    TSL_ASSERT_ARG(0 != frame_bytes);
    TSL_ASSERT_ARG(0 != nr_frames);
    struct frame_alloc *fa = calloc(1, sizeof(*fa));
    if (!fa) {
        return A_E_NOMEM;
    }
    fa->frame_bytes = (frame_bytes + 63u) & ~(size_t)63u;
    fa->nr_frames = nr_frames;
    fa->slab_free = slab_free;
    fa->slab = slab_alloc(fa->frame_bytes * nr_frames);
    if (NULL == fa->slab) {
        free(fa);
        return A_E_NOMEM;
    }
    fa->free_stack = malloc(nr_frames * sizeof(void *));
    if (!fa->free_stack) {
        slab_free(fa->slab);
        free(fa);
        return A_E_NOMEM;
    }
    for (size_t i = 0; i < nr_frames; i++) {
        fa->free_stack[i] = fa->slab + i * fa->frame_bytes;
    }
    fa->nr_free = nr_frames;
    fa->head = 0;
    pthread_mutex_init(&fa->lock, NULL);
    *pfa = fa;
    return A_OK;
}

aresult_t frame_alloc(struct frame_alloc *fa, void **pframe)
{
    TSL_ASSERT_ARG(NULL != fa);
    TSL_ASSERT_ARG(NULL != pframe);
    aresult_t ret = A_E_NOMEM;
    *pframe = NULL;
    pthread_mutex_lock(&fa->lock);
    if (fa->nr_free) {
        *pframe = fa->free_stack[fa->head];
        fa->head = (fa->head + 1) % fa->nr_frames;
        fa->nr_free--;
        ret = A_OK;
    }
    pthread_mutex_unlock(&fa->lock);
    return ret;
}

aresult_t frame_free(struct frame_alloc *fa, void **pframe)
{
    TSL_ASSERT_ARG(NULL != fa);
    TSL_ASSERT_ARG(NULL != pframe && NULL != *pframe);
    pthread_mutex_lock(&fa->lock);
    TSL_BUG_ON(fa->nr_free == fa->nr_frames);
    fa->free_stack[(fa->head + fa->nr_free) % fa->nr_frames] = *pframe;
    fa->nr_free++;
    pthread_mutex_unlock(&fa->lock);
    *pframe = NULL;
    return A_OK;
}

size_t frame_alloc_frame_bytes(struct frame_alloc *fa)
{
    return fa->frame_bytes; /* bytes from one frame of the slab to the next */
}

size_t frame_alloc_nr_free(struct frame_alloc *fa)
{
    pthread_mutex_lock(&fa->lock);
    size_t n = fa->nr_free;
    pthread_mutex_unlock(&fa->lock);
    return n;
}

aresult_t frame_alloc_delete(struct frame_alloc **pfa)
{
    TSL_ASSERT_ARG(NULL != pfa);
    if (*pfa) {
        pthread_mutex_destroy(&(*pfa)->lock);
        free((*pfa)->free_stack);
        (*pfa)->slab_free((*pfa)->slab);
        free(*pfa);
        *pfa = NULL;
    }
    return A_OK;
}

uint64_t tsl_get_clock_monotonic(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
