/*
 * mfm_tsl.h - the small slice of the external TSL support library that multifm's receiver-side code
 * is written against (result codes, argument/bug checks, logging, aligned allocation, container_of,
 * worker threads, a fixed frame pool), re-provided from scratch.  The reference links pvachon/tsl
 * (CMakeLists.txt:84); it is not vendored there and not installed here.  Names match the ones the
 * reference's front ends use so file_if/rtl_sdr_if-style code reads the same;
 * the numeric values of the A_E_* codes are this library's own.
 */
#pragma once

#include <stdatomic.h>
#include <pthread.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int aresult_t;

#define A_OK 0
#define A_E_INVAL (-1)
#define A_E_NOMEM (-2)
#define A_E_BUSY (-3)
#define A_E_DONE (-6)
#define A_E_NOTFOUND (-7)
#define A_E_DEVICE (-4)

#define FAILED(x) ((x) < 0)
#define FAILED_UNLIKELY(x) (__builtin_expect(((x) < 0), 0))

#define SEV_INFO "I"
#define SEV_WARNING "W"
#define SEV_ERROR "E"
#define SEV_FATAL "F"

/* MESSAGE(subsystem, severity, ident, fmt, ...) -> one line on stderr */
#define MESSAGE(subsys, sev, ident, fmt, ...)                                                                \
    fprintf(stderr, "%s:%s:%s " fmt "\n", (subsys), (sev), (ident), ##__VA_ARGS__)

#ifdef _TSL_DEBUG
#define DIAG(fmt, ...) fprintf(stderr, "DIAG %s:%d " fmt "\n", __FILE__, __LINE__, ##__VA_ARGS__)
#else
#define DIAG(fmt, ...)                                                                                       \
    do {                                                                                                     \
    } while (0)
#endif

#define PANIC(fmt, ...)                                                                                      \
    do {                                                                                                     \
        fprintf(stderr, "PANIC %s:%d " fmt "\n", __FILE__, __LINE__, ##__VA_ARGS__);                         \
        abort();                                                                                             \
    } while (0)

/* argument check: returns an error to the caller */
#define TSL_ASSERT_ARG(cond)                                                                                 \
    do {                                                                                                     \
        if (!(cond)) {                                                                                       \
            fprintf(stderr, "ASSERT-ARG %s:%d %s\n", __FILE__, __LINE__, #cond);                             \
            return A_E_INVAL;                                                                                \
        }                                                                                                    \
    } while (0)
#define TSL_ASSERT_ARG_DEBUG(cond) TSL_ASSERT_ARG(cond)

/* invariant check: aborts */
#define TSL_BUG_ON(cond)                                                                                     \
    do {                                                                                                     \
        if (cond) {                                                                                          \
            PANIC("BUG: %s", #cond);                                                                         \
        }                                                                                                    \
    } while (0)
#define TSL_BUG_IF_FAILED(expr)                                                                              \
    do {                                                                                                     \
        aresult_t bug_ret_ = (expr);                                                                         \
        if (FAILED(bug_ret_)) {                                                                              \
            PANIC("BUG: %s failed (%d)", #expr, bug_ret_);                                                   \
        }                                                                                                    \
    } while (0)

#define BL_CONTAINER_OF(ptr, type, member) ((type *)((char *)(ptr) - offsetof(type, member)))
#define BL_MIN2(a, b) ((a) < (b) ? (a) : (b))
#define CAL_ALIGN(n) __attribute__((aligned(n)))
#define CAL_CACHE_ALIGNED __attribute__((aligned(64)))
#ifndef SYS_CACHE_LINE_LENGTH
#define SYS_CACHE_LINE_LENGTH 64
#endif

static inline aresult_t tsl_aligned_zalloc(void **p, size_t bytes, size_t align)
{
    void *m = NULL;
    if (align < sizeof(void *)) {
        align = sizeof(void *);
    }
    if (0 != posix_memalign(&m, align, bytes ? bytes : align)) {
        *p = NULL;
        return A_E_NOMEM;
    }
    memset(m, 0, bytes);
    *p = m;
    return A_OK;
}
#define TACALLOC(pptr, n, size, align) tsl_aligned_zalloc((void **)(pptr), (size_t)(n) * (size_t)(size), (align))
#define TZAALLOC(ptr, align) tsl_aligned_zalloc((void **)&(ptr), sizeof(*(ptr)), (align))
#define TFREE(ptr)                                                                                           \
    do {                                                                                                     \
        free(ptr);                                                                                           \
        (ptr) = NULL;                                                                                        \
    } while (0)

/* ---- worker thread ---- */
struct worker_thread;
typedef aresult_t (*worker_thread_func_t)(struct worker_thread *thr);
#define WORKER_THREAD_CPU_MASK_ANY (~0u)

struct worker_thread {
    pthread_t thr;
    worker_thread_func_t fn;
    _Atomic bool running; /* cleared by whoever asks the thread to stop */
    bool started;
};

aresult_t worker_thread_new(struct worker_thread *thr, worker_thread_func_t fn, unsigned cpu);
aresult_t worker_thread_request_shutdown(struct worker_thread *thr);
aresult_t worker_thread_delete(struct worker_thread *thr);
static inline bool worker_thread_is_running(struct worker_thread *thr)
{
    return thr->running;
}

/* ---- fixed pool of equally sized frames (sample buffers) ---- */
struct frame_alloc;
aresult_t frame_alloc_new(struct frame_alloc **pfa, size_t frame_bytes, size_t nr_frames);
aresult_t frame_alloc_new_on(struct frame_alloc **pfa, size_t frame_bytes, size_t nr_frames, void *(*slab_alloc)(size_t),
                             void (*slab_free)(void *));
aresult_t frame_alloc(struct frame_alloc *fa, void **pframe); /* A_E_NOMEM when the pool is empty */
aresult_t frame_free(struct frame_alloc *fa, void **pframe);
aresult_t frame_alloc_delete(struct frame_alloc **pfa);
size_t frame_alloc_nr_free(struct frame_alloc *fa);

uint64_t tsl_get_clock_monotonic(void); /* ns */
