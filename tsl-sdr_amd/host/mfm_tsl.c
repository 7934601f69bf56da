/* mfm_tsl.c - worker threads, frame pool and clock behind mfm_tsl.h. */
#include "mfm_tsl.h"

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

/* ---- frame pool: nr_frames equally sized, 64-byte aligned frames on a lock-protected free stack ---- */

struct frame_alloc {
    uint8_t *slab;
    void (*slab_free)(void *);
    void **free_stack;
    size_t frame_bytes, nr_frames, nr_free;
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
        *pframe = fa->free_stack[--fa->nr_free];
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
    fa->free_stack[fa->nr_free++] = *pframe;
    pthread_mutex_unlock(&fa->lock);
    *pframe = NULL;
    return A_OK;
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
