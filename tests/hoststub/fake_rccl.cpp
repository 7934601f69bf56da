/* TEST DOUBLE of the slice of RCCL that mfm_group.hip uses (ncclCommInitAll, ncclGroupStart/End, ncclBroadcast,
 * ncclSend/ncclRecv, ncclAllGather, ncclCommDestroy, ncclGetErrorString), for the one-GPU box: every "rank" lives on the same
 * device and the collectives are device-to-device copies ordered by events across the ranks' streams.  With it - built as
 * librccl.so into a temporary directory that a test process puts first in LD_LIBRARY_PATH - a device group of SEVERAL shards
 * runs on one GPU (MFM_F_GROUP_SHARED_DEVICE), so the shard arithmetic, the exchange's pointer arithmetic (which part goes
 * where, in-place all-gather) and the per-shard fetch are checked against the oracle without a multi-GPU node.  It says
 * nothing about links or speed.  Test infrastructure only. */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

struct FakeComm {
    int rank, size;
};
struct Op {
    int kind; /* 0 bcast, 1 send, 2 recv, 3 allgather */
    const void *src;
    void *dst;
    size_t bytes;
    int peer_or_root;
    FakeComm *comm;
    hipStream_t stream;
};
static std::vector<Op> g_ops;
static int g_depth = 0;
static int g_fail = 0;

static void after(hipStream_t producer, hipStream_t consumer)
{
    if (producer == consumer) {
        return;
    }
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, producer) != hipSuccess ||
        hipStreamWaitEvent(consumer, ev, 0) != hipSuccess) {
        g_fail = 1;
    }
    (void)hipEventDestroy(ev); /* destroyed when the recorded work completes */
}

static void copy(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    if (dst != src && bytes && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        g_fail = 1;
    }
}

static int flush()
{
    /* everything of one group: the ranks' calls arrive rank by rank; data moves on the RECEIVER's stream, behind what the
     * sender's stream has queued so far */
    for (const Op &o : g_ops) {
        if (o.kind == 0) {
            const Op *root = nullptr;
            for (const Op &r : g_ops) {
                if (r.kind == 0 && r.comm->rank == o.peer_or_root) {
                    root = &r;
                }
            }
            if (!root) {
                return 3;
            }
            after(root->stream, o.stream);
            copy(o.dst, root->src, o.bytes, o.stream);
        } else if (o.kind == 2) {
            const Op *snd = nullptr;
            for (const Op &r : g_ops) {
                if (r.kind == 1 && r.comm->rank == o.peer_or_root && r.peer_or_root == o.comm->rank) {
                    snd = &r;
                }
            }
            if (!snd || snd->bytes != o.bytes) {
                return 3;
            }
            after(snd->stream, o.stream);
            copy(o.dst, snd->src, o.bytes, o.stream);
        } else if (o.kind == 3) {
            for (const Op &r : g_ops) {
                if (r.kind == 3) {
                    after(r.stream, o.stream);
                    copy(static_cast<char *>(o.dst) + (size_t)r.comm->rank * o.bytes, r.src, o.bytes, o.stream);
                }
            }
        }
    }
    /* an all-gather reads every rank's part: nobody may go on (and overwrite it) before every reader has queued its copy -
     * later work of any rank waits for all streams of the group */
    for (const Op &a : g_ops) {
        for (const Op &b : g_ops) {
            after(a.stream, b.stream);
        }
    }
    g_ops.clear();
    return g_fail ? 1 : 0;
}

static int post(const Op &o)
{
    g_ops.push_back(o);
    return g_depth ? 0 : flush();
}

extern "C" {

int ncclCommInitAll(void **comms, int n, const int *devs)
{
    (void)devs;
    for (int i = 0; i < n; i++) {
        comms[i] = new FakeComm{ i, n };
    }
    return 0;
}
int ncclCommDestroy(void *c)
{
    delete static_cast<FakeComm *>(c);
    return 0;
}
int ncclCommCount(void *c, int *count)
{
    *count = static_cast<FakeComm *>(c)->size;
    return 0;
}
int ncclGroupStart() { g_depth++; return 0; }
int ncclGroupEnd() { return --g_depth ? 0 : flush(); }
int ncclBroadcast(const void *s, void *d, size_t count, int dtype, int root, void *c, hipStream_t st)
{
    (void)dtype;
    return post(Op{ 0, s, d, count, root, static_cast<FakeComm *>(c), st });
}
int ncclSend(const void *s, size_t count, int dtype, int peer, void *c, hipStream_t st)
{
    (void)dtype;
    return post(Op{ 1, s, nullptr, count, peer, static_cast<FakeComm *>(c), st });
}
int ncclRecv(void *d, size_t count, int dtype, int peer, void *c, hipStream_t st)
{
    (void)dtype;
    return post(Op{ 2, nullptr, d, count, peer, static_cast<FakeComm *>(c), st });
}
int ncclAllGather(const void *s, void *d, size_t count, int dtype, void *c, hipStream_t st)
{
    (void)dtype;
    return post(Op{ 3, s, d, count, 0, static_cast<FakeComm *>(c), st });
}
const char *ncclGetErrorString(int e) { return e ? "fake rccl: unmatched or failed operation" : "ok"; }
}
