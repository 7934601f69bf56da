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
/* a pointer handed over by reference: neither the reference nor what it refers to may be NULL
 * (filter/polyphase_fir.c:120) */
#define TSL_ASSERT_PTR_BY_REF(pptr)                                                                          \
    do {                                                                                                     \
        TSL_ASSERT_ARG(NULL != (pptr));                                                                      \
        TSL_ASSERT_ARG(NULL != *(pptr));                                                                     \
    } while (0)

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

/* TCALLOC(&ptr, a, b): a * b zeroed bytes (multifm/rtl_sdr_if.c:248, decoder/decoder.c:528,560) */
static inline aresult_t tsl_zalloc(void **p, size_t a, size_t b)
{
    if (NULL == p) {
        return A_E_INVAL;
    }
    *p = calloc(a ? a : 1, b ? b : 1);
    return NULL == *p ? A_E_NOMEM : A_OK;
}
#define TCALLOC(pptr, a, b) tsl_zalloc((void **)(pptr), (size_t)(a), (size_t)(b))

/* `type *x CAL_CLEANUP(free_xxx) = NULL;` releases x when it goes out of scope (multifm/rtl_sdr_if.c:230,
 * multifm/receiver.c:107-115, multifm/multifm.c:92) */
#define CAL_CLEANUP(fn) __attribute__((cleanup(fn)))
static inline void free_memory(void **p)
{
    if (NULL != p && NULL != *p) {
        free(*p);
        *p = NULL;
    }
}
static inline void free_double_array(double **p)
{
    free_memory((void **)p);
}
static inline void free_i16_array(int16_t **p)
{
    free_memory((void **)p);
}
static inline void free_u32_array(uint32_t **p)
{
    free_memory((void **)p);
}
static inline void free_string(char **p)
{
    free_memory((void **)p);
}

/* ---- intrusive circular list (<tsl/list.h>): multifm/receiver.c:89,186,236-237,303-304, demod.c:337 ---- */
struct list_entry {
    struct list_entry *prev, *next;
};
#define LIST_INIT(name) { &(name), &(name) }

static inline void list_init(struct list_entry *e)
{
    e->prev = e->next = e;
}

/* at the tail: a walk sees entries in the order they were appended */
static inline void list_append(struct list_entry *head, struct list_entry *e)
{
    e->prev = head->prev;
    e->next = head;
    head->prev->next = e;
    head->prev = e;
}

static inline void list_del(struct list_entry *e)
{
    e->prev->next = e->next;
    e->next->prev = e->prev;
    e->prev = e->next = e;
}

static inline bool list_empty(const struct list_entry *head)
{
    return head->next == head;
}

#define list_for_each_type(pos, head, member)                                                                \
    for (pos = BL_CONTAINER_OF((head)->next, __typeof__(*pos), member); &pos->member != (head);              \
         pos = BL_CONTAINER_OF(pos->member.next, __typeof__(*pos), member))
/* `pos` may be unlinked (and freed) inside the body */
#define list_for_each_type_safe(pos, tmp, head, member)                                                      \
    for (pos = BL_CONTAINER_OF((head)->next, __typeof__(*pos), member),                                      \
        tmp = BL_CONTAINER_OF(pos->member.next, __typeof__(*pos), member);                                   \
         &pos->member != (head); pos = tmp, tmp = BL_CONTAINER_OF(tmp->member.next, __typeof__(*pos), member))

/* ---- bounded FIFO of pointers (<tsl/work_queue.h>): multifm/demod.c:134,176,297, receiver.c:91.  Not locked: the
 * reference takes its own mutex around every call (receiver.c:90-92, demod.c:133-135).  Popping an empty queue is A_OK
 * with *value = NULL (demod.c:134-136 relies on it); pushing into a full one is A_E_BUSY. ---- */
struct work_queue {
    void **slots;
    unsigned depth, head, count;
};
aresult_t work_queue_new(struct work_queue *wq, unsigned depth);
aresult_t work_queue_release(struct work_queue *wq);
aresult_t work_queue_push(struct work_queue *wq, void *value);
aresult_t work_queue_pop(struct work_queue *wq, void **value);
aresult_t work_queue_size(struct work_queue *wq, unsigned *count);

/* ---- application scaffolding (<app/app.h>): multifm/multifm.c:114-115,163, decoder/decoder.c:668,679-680 ---- */
struct config;
typedef void (*app_sigint_handler_t)(void);
aresult_t app_init(const char *app_name, struct config *cfg);
aresult_t app_sigint_catch(app_sigint_handler_t handler); /* NULL: just stop app_running() */
int app_running(void);
/* not in TSL: a finite input (file front end at end of file) ends the application's main loop the way SIGINT would */
void app_request_stop(void);

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
size_t frame_alloc_frame_bytes(struct frame_alloc *fa); /* distance between neighbouring frames of the pool's slab */

uint64_t tsl_get_clock_monotonic(void); /* ns */
