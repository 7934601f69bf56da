/* TEST DOUBLE of the device-group entry points of include/multifm_hip.h, for the CPU-only test of the receiver's
 * ingest policy (tests/test_host.py::test_deliver_never_blocks...): no GPU, no arithmetic - push() can be stalled
 * (returns MFM_E_BUSY while stub_set_busy(1)), everything pushed is counted.  Test infrastructure only. */
#include "../../include/multifm_hip.h"

#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

struct mfm_group {
    uint32_t nr_channels;
    int committed;
};

static _Atomic int g_busy;
static _Atomic unsigned long g_pushed, g_samples, g_busy_returns, g_order_errors, g_next_seq;

void stub_set_busy(int busy) { g_busy = busy; }
unsigned long stub_pushed(void) { return g_pushed; }
unsigned long stub_samples(void) { return g_samples; }
unsigned long stub_busy_returns(void) { return g_busy_returns; }
/* buffers whose first word was not the next sequence number (the test's front end numbers what it delivers): a wrong
 * stride or offset in the receiver's run detection, a buffer pushed twice or out of order */
unsigned long stub_order_errors(void) { return g_order_errors; }

const char *mfm_last_error(void) { return "stub"; }
const char *mfm_strerror(int e) { (void)e; return "stub"; }

int mfm_group_create(struct mfm_group **pg, const struct mfm_group_config *cfg)
{
    (void)cfg;
    *pg = calloc(1, sizeof(**pg));
    return *pg ? MFM_OK : MFM_E_NOMEM;
}
void mfm_group_destroy(struct mfm_group **pg) { free(*pg); *pg = NULL; }
int mfm_group_add_channel(struct mfm_group *g, int32_t o, const double *t, size_t n, double gain, int iq)
{
    (void)o, (void)t, (void)n, (void)gain, (void)iq;
    return (int)g->nr_channels++;
}
int mfm_group_commit(struct mfm_group *g) { g->committed = 1; return MFM_OK; }
int mfm_group_nr_shards(struct mfm_group *g) { (void)g; return 1; }
int mfm_group_shard_info(struct mfm_group *g, uint32_t s, uint32_t *first, uint32_t *count, int32_t *dev)
{
    (void)s;
    if (first) *first = 0;
    if (count) *count = g->nr_channels;
    if (dev) *dev = 0;
    return MFM_OK;
}
int mfm_group_push(struct mfm_group *g, const void *data, size_t nr_samples, int format)
{
    (void)g, (void)data, (void)format;
    if (g_busy) {
        g_busy_returns++;
        return MFM_E_BUSY;
    }
    uint32_t seq = 0;
    memcpy(&seq, data, sizeof(seq));
    if (seq != (uint32_t)g_next_seq) {
        g_order_errors++;
    }
    g_next_seq = seq + 1u;
    g_pushed++;
    g_samples += nr_samples;
    return MFM_OK;
}
/* the pinned form: the stand-in "copies" at once, so every ticket is done when it is asked for */
static _Atomic unsigned long g_tickets;
int mfm_group_push_pinned(struct mfm_group *g, const void *data, size_t nr_samples, int format, uint64_t *ticket)
{
    const int rc = mfm_group_push(g, data, nr_samples, format);
    if (MFM_OK == rc && ticket) {
        *ticket = ++g_tickets;
    }
    return rc;
}
/* a run of neighbouring buffers: the stand-in takes up to three of them at a time (so that "fewer than offered" is exercised) */
int mfm_group_push_pinned_run(struct mfm_group *g, const void *first, size_t stride_bytes, size_t nr_samples_each, size_t count,
                              int format, uint64_t *ticket, size_t *accepted)
{
    size_t k = count > 3 ? 3 : count;
    if (accepted) {
        *accepted = 0;
    }
    if (g_busy) {
        g_busy_returns++;
        return MFM_E_BUSY;
    }
    for (size_t i = 0; i < k; i++) {
        (void)mfm_group_push(g, (const uint8_t *)first + i * stride_bytes, nr_samples_each, format);
    }
    if (ticket) {
        *ticket = ++g_tickets;
    }
    if (accepted) {
        *accepted = k;
    }
    return MFM_OK;
}
int mfm_group_copy_done(struct mfm_group *g, uint64_t ticket) { (void)g, (void)ticket; return 1; }
int mfm_group_copy_wait(struct mfm_group *g, uint64_t ticket) { (void)g, (void)ticket; return MFM_OK; }
void *mfm_host_alloc(size_t bytes) { void *p = NULL; return 0 == posix_memalign(&p, 64, bytes) ? p : NULL; }
void mfm_host_free(void *p) { free(p); }
int mfm_group_fetch(struct mfm_group *g, struct mfm_block *blks) { (void)g, (void)blks; return MFM_E_DONE; }
int mfm_group_release(struct mfm_group *g) { (void)g; return MFM_OK; }
int mfm_group_sync(struct mfm_group *g) { (void)g; return MFM_OK; }
int mfm_group_flush(struct mfm_group *g) { (void)g; return MFM_OK; }
int mfm_group_get_stats(struct mfm_group *g, uint32_t s, struct mfm_stats *st)
{
    (void)g, (void)s;
    memset(st, 0, sizeof(*st));
    return MFM_OK;
}
