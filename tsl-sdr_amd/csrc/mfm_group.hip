/*
 * mfm_group.hip - one channel set on several MI355X GPUs of one node, behind the C ABI (include/multifm_hip.h,
 * mfm_group_*).
 *
 * The reference fans every delivered sample_buf out to all of its channel threads (multifm/receiver.c:78-98): the
 * channels are independent given the same wideband input.  Here the channel set is cut into contiguous shards, one
 * mfm_engine per device; a delivered block is staged on the first device (H2D; an 8-bit block stays bytes when every
 * member's kernel can read it so, and is widened there otherwise) and
 * exchanged with RCCL over xGMI, in place into every other engine's input buffer - and then every engine submits it.
 * Two forms of the exchange (mfm_group_config::exchange):
 *   MFM_X_RCCL            ncclBroadcast from the root.  A broadcast is a pipeline along a ring: every byte crosses one
 *                         link per hop, so the block arrives at the rate of ONE xGMI link (~153 GB/s) whatever the number
 *                         of GPUs.
 *   MFM_X_RCCL_ALLGATHER  the root sends 1/S of the block to each of its S - 1 peers (ncclSend / ncclRecv inside one
 *                         group: S - 1 different links at once), then ncclAllGather in place: every GPU ends up with the
 *                         whole block, and every GPU both sends and receives on all its links.  The volume per GPU is the
 *                         same; what changes is that no single link carries the whole block (DESIGN.md section 7 has the
 *                         numbers: the break-even channel count per GPU drops by the number of links in use).
 * No other exchange: each device copies its own PCM back, the caller demultiplexes by shard.
 * One process, one host thread drives all devices (ncclCommInitAll + ncclGroupStart/End).  The ORDER of a push - what
 * is checked before anything changes, what happens under the lock that fetch takes - is mfm_group_seq.h.
 *
 * RCCL is loaded at run time (dlopen "librccl.so"), and only when a group really exchanges: a single-device group
 * stages and submits directly and never touches it.  MFM_X_RCCL forces the exchange path for a single device too, so
 * that the RCCL call sequence can be exercised on a one-GPU box.
 */
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <link.h>

#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <cstdlib>
#include <vector>

#include "../../include/multifm_hip.h"
#include "mfm_engine_internal.h"
#include "mfm_group_seq.h"

extern "C" __attribute__((visibility("hidden"))) void mfm_internal_set_error(const char *msg);

namespace {

int gfail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    mfm_internal_set_error(buf);
    return code;
}

/* the slice of the RCCL API used here (rccl.h: ncclResult_t / ncclComm_t / ncclDataType_t are int / pointer / int) */
typedef void *nccl_comm_t;
struct RcclApi {
    void *lib = nullptr;
    int (*CommInitAll)(nccl_comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(nccl_comm_t, int *) = nullptr;
};
constexpr int kNcclInt8 = 0; /* rccl.h: ncclInt8 = ncclChar = 0 */

/* a librccl that is mapped into the process already: its path (dl_iterate_phdr) */
int find_mapped_rccl(struct dl_phdr_info *info, size_t, void *out)
{
    const char *name = info->dlpi_name;
    if (name && *name) {
        const char *base = strrchr(name, '/');
        base = base ? base + 1 : name;
        if (0 == strncmp(base, "librccl.so", 10)) {
            snprintf(static_cast<char *>(out), 1024, "%s", name);
            return 1;
        }
    }
    return 0;
}

/* One process may carry several copies of RCCL (PyTorch ships its own beside /opt/rocm's); two of them in one address space
 * each bring their own device state.  So: the file MFM_RCCL_LIBRARY names, if set; else whatever is mapped already; else the loader's own search for the bare name
 * (LD_LIBRARY_PATH, the cache), and where that finds nothing the ROCm tree the environment names and the usual place. */
std::mutex g_rccl_mu;
RcclApi g_rccl;
char g_rccl_path[1024];

int load_rccl(RcclApi *api)
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl.lib) {
        void *h = nullptr;
        std::string tried;
        char mapped[1024] = "";
        /* MFM_RCCL_LIBRARY: the operator names the file (a site build of RCCL; the tests' transport double under a launcher that
         * has mapped PyTorch's copy already) - taken as given, nothing else is tried behind it */
        if (const char *named = getenv("MFM_RCCL_LIBRARY")) {
            if (*named) {
                h = dlopen(named, RTLD_NOW | RTLD_LOCAL);
                if (!h) {
                    return gfail(MFM_E_DEVICE, "cannot load MFM_RCCL_LIBRARY=%s: %s", named, dlerror());
                }
            }
        }
        if (!h && dl_iterate_phdr(find_mapped_rccl, mapped) && mapped[0]) {
            h = dlopen(mapped, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            tried += std::string(mapped) + " (mapped); ";
        }
        std::vector<std::string> cands;
        cands.push_back("librccl.so"); /* the loader's search: LD_LIBRARY_PATH, the cache */
        cands.push_back("librccl.so.1");
        if (const char *rocm = getenv("ROCM_PATH")) {
            if (*rocm) {
                cands.push_back(std::string(rocm) + "/lib/librccl.so");
            }
        }
        cands.push_back("/opt/rocm/lib/librccl.so");
        for (size_t i = 0; i < cands.size() && !h; i++) {
            h = dlopen(cands[i].c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!h) {
                tried += cands[i] + "; ";
            }
        }
        if (!h) {
            return gfail(MFM_E_DEVICE, "cannot load librccl.so (tried: %s last error: %s): a device group of more than one GPU needs RCCL",
                         tried.c_str(), dlerror());
        }
        RcclApi a;
        a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(dlsym(h, "ncclBroadcast"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
        a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(h, "ncclSend"));
        a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(h, "ncclRecv"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(h, "ncclCommCount")); /* optional */
        if (!a.CommInitAll || !a.CommDestroy || !a.Broadcast || !a.GroupStart || !a.GroupEnd || !a.AllGather || !a.Send || !a.Recv) {
            dlclose(h);
            return gfail(MFM_E_DEVICE, "librccl.so lacks a required entry point");
        }
        Dl_info di;
        if (dladdr(reinterpret_cast<void *>(a.CommInitAll), &di) && di.dli_fname) {
            snprintf(g_rccl_path, sizeof(g_rccl_path), "%s", di.dli_fname);
        }
        a.lib = h;
        g_rccl = a;
    }
    *api = g_rccl;
    return MFM_OK;
}

struct ChanDesc {
    int32_t offset_hz;
    std::vector<double> taps;
    double gain;
    int want_iq;
};

} /* namespace */

struct mfm_group {
    mfm_group_config cfg{};
    std::vector<ChanDesc> chans;
    std::vector<mfm_engine *> eng; /* one per shard (shards with channels only) */
    std::vector<int> dev;
    std::vector<uint32_t> first, count;
    bool committed = false;
    bool exchange = false; /* blocks travel through RCCL */
    RcclApi rccl;
    std::vector<nccl_comm_t> comm;
    std::vector<hipStream_t> xs; /* exchange stream per shard (shard 0: the root engine's copy stream) */
    uint64_t blocks = 0, bytes_exchanged = 0;
    /* MFM_F_TIMING: one exchange in four is bracketed by an event pair on every shard's exchange stream */
    std::vector<hipEvent_t> x0, x1;
    std::vector<double> x_ms;
    std::vector<uint64_t> x_n;
    bool x_open = false;      /* a pair has been recorded and not yet folded */
    uint64_t x_seq = 0;       /* exchanges so far */
    std::mutex x_mu;          /* the sums, between the push thread and whoever asks (mfm_group_exchange_detail) */
    std::atomic<bool> broken{ false }; /* a push failed after the first shard had taken the block (mfm_group_seq.h); the push
                                          thread writes it, the fetch thread reads it */
    std::mutex mu;       /* push's submit loop against fetch's "does every shard hold a block" */
};

extern "C" {

void mfm_shard_range(uint32_t nr_channels, uint32_t nr_shards, uint32_t shard, uint32_t *first, uint32_t *count)
{
    /* contiguous ranges whose sizes differ by at most one (any contiguous partition is valid);
     * the same rule as tsl-sdr_amd/dist.py shard_range(), which bench.py's ranks use.  Shards are empty only when
     * there are fewer channels than shards. */
    uint32_t lo = 0, hi = 0;
    if (nr_shards && shard < nr_shards) {
        const uint32_t base = nr_channels / nr_shards, extra = nr_channels % nr_shards;
        lo = shard * base + (shard < extra ? shard : extra);
        hi = lo + base + (shard < extra ? 1u : 0u);
    }
    if (first) {
        *first = lo;
    }
    if (count) {
        *count = hi - lo;
    }
}

int mfm_group_create(struct mfm_group **pg, const struct mfm_group_config *cfg)
{
    if (!pg || !cfg) {
        return gfail(MFM_E_INVAL, "NULL argument");
    }
    *pg = nullptr;
    if (cfg->abi_version != MFM_ABI_VERSION) {
        return gfail(MFM_E_INVAL, "ABI version %u, library is %u", cfg->abi_version, MFM_ABI_VERSION);
    }
    if (cfg->nr_devices < 1 || cfg->nr_devices > MFM_GROUP_MAX_DEVICES) {
        return gfail(MFM_E_INVAL, "a device group has 1..%d devices, not %u", MFM_GROUP_MAX_DEVICES, cfg->nr_devices);
    }
    for (uint32_t i = 0; i < cfg->nr_devices && !(cfg->flags & MFM_F_GROUP_SHARED_DEVICE); i++) {
        for (uint32_t j = 0; j < i; j++) {
            if (cfg->devices[i] == cfg->devices[j]) {
                return gfail(MFM_E_INVAL, "device %d listed twice", cfg->devices[i]);
            }
        }
    }
    if (0 == cfg->decimation || 0 == cfg->sample_rate_hz || 0 == cfg->max_block_samples) {
        return gfail(MFM_E_INVAL, "decimation, sample rate and max block must be non-zero");
    }
    /* (MFM_F_DEVICE_ONLY: the shards keep their outputs in HBM - mfm_group_fetch() is not available then; blocks are handed
     * over in device memory with mfm_group_acquire_input() / mfm_group_submit(), outputs are read through
     * mfm_group_shard_engine()) */
    mfm_group *g = new (std::nothrow) mfm_group();
    if (!g) {
        return gfail(MFM_E_NOMEM, "group allocation failed");
    }
    g->cfg = *cfg;
    *pg = g;
    return MFM_OK;
}

static void group_release(mfm_group *g)
{
    for (size_t i = 0; i < g->eng.size(); i++) {
        if (g->eng[i]) {
            (void)mfm_engine_sync(g->eng[i]);
        }
    }
    for (size_t i = 0; i < g->comm.size(); i++) {
        if (g->comm[i] && g->rccl.CommDestroy) {
            (void)hipSetDevice(g->dev[i]);
            (void)g->rccl.CommDestroy(g->comm[i]);
        }
    }
    g->comm.clear();
    for (size_t i = 0; i < g->x0.size(); i++) {
        (void)hipSetDevice(g->dev[i]);
        (void)hipEventDestroy(g->x0[i]);
        (void)hipEventDestroy(g->x1[i]);
    }
    g->x0.clear();
    g->x1.clear();
    g->x_ms.clear();
    g->x_n.clear();
    g->x_open = false;
    for (size_t i = 1; i < g->xs.size(); i++) { /* xs[0] belongs to the root engine */
        if (g->xs[i]) {
            (void)hipSetDevice(g->dev[i]);
            (void)hipStreamDestroy(g->xs[i]);
        }
    }
    g->xs.clear();
    for (size_t i = 0; i < g->eng.size(); i++) {
        mfm_engine_destroy(&g->eng[i]);
    }
    g->eng.clear();
    /* a group whose commit failed may be committed again: nothing of the failed attempt may survive */
    g->dev.clear();
    g->first.clear();
    g->count.clear();
    g->broken = false;
    g->committed = false;
}

void mfm_group_destroy(struct mfm_group **pg)
{
    if (!pg || !*pg) {
        return;
    }
    group_release(*pg);
    /* the RCCL library stays loaded: unloading it under a process that may hold other communicators is not safe */
    delete *pg;
    *pg = nullptr;
}

int mfm_group_add_channel(struct mfm_group *g, int32_t offset_hz, const double *lpf_taps, size_t nr_taps, double channel_gain,
                          int want_iq)
{
    if (!g || !lpf_taps || 0 == nr_taps) {
        return gfail(MFM_E_INVAL, "NULL or empty taps");
    }
    if (g->committed) {
        return gfail(MFM_E_STATE, "channel set is frozen after commit");
    }
    if (!g->chans.empty() && g->chans[0].taps.size() != nr_taps) {
        return gfail(MFM_E_INVAL, "all channels share one tap count (%zu), got %zu", g->chans[0].taps.size(), nr_taps);
    }
    if (nr_taps < g->cfg.decimation) {
        return gfail(MFM_E_INVAL, "taps (%zu) < decimation (%u) is not a valid multifm configuration", nr_taps,
                     g->cfg.decimation);
    }
    ChanDesc c;
    c.offset_hz = offset_hz;
    c.taps.assign(lpf_taps, lpf_taps + nr_taps);
    c.gain = channel_gain;
    c.want_iq = want_iq;
    g->chans.push_back(std::move(c));
    return (int)g->chans.size() - 1;
}

int mfm_group_commit(struct mfm_group *g)
{
    if (!g) {
        return gfail(MFM_E_INVAL, "NULL group");
    }
    if (g->committed) {
        return gfail(MFM_E_STATE, "already committed");
    }
    if (g->chans.empty()) {
        return gfail(MFM_E_INVAL, "no channels");
    }
    const uint32_t C = (uint32_t)g->chans.size(), G = g->cfg.nr_devices;
    int rc = MFM_OK;
    for (uint32_t s = 0; s < G && rc == MFM_OK; s++) {
        uint32_t lo = 0, n = 0;
        mfm_shard_range(C, G, s, &lo, &n);
        if (0 == n) {
            continue; /* fewer channels than devices: the trailing devices stay idle */
        }
        mfm_engine_config ec;
        memset(&ec, 0, sizeof(ec));
        ec.abi_version = MFM_ABI_VERSION;
        ec.device = g->cfg.devices[s];
        ec.sample_rate_hz = g->cfg.sample_rate_hz;
        ec.decimation = g->cfg.decimation;
        ec.max_block_samples = g->cfg.max_block_samples;
        ec.flags = g->cfg.flags;
        ec.coalesce_samples = g->cfg.coalesce_samples;
        mfm_engine *e = nullptr;
        rc = mfm_engine_create(&e, &ec);
        if (rc != MFM_OK) {
            break;
        }
        g->eng.push_back(e);
        g->dev.push_back(ec.device);
        g->first.push_back(lo);
        g->count.push_back(n);
        for (uint32_t c = lo; c < lo + n && rc >= 0; c++) {
            const ChanDesc &d = g->chans[c];
            rc = mfm_engine_add_channel(e, d.offset_hz, d.taps.data(), d.taps.size(), d.gain, d.want_iq);
        }
        rc = rc < 0 ? rc : mfm_engine_commit(e);
    }
    if (rc != MFM_OK) {
        group_release(g);
        return rc; /* the failing call left its message */
    }
    const size_t S = g->eng.size();
    if (g->cfg.exchange > MFM_X_RCCL_ALLGATHER) {
        group_release(g);
        return gfail(MFM_E_INVAL, "unknown exchange mode %u", g->cfg.exchange);
    }
    g->exchange = S > 1 || g->cfg.exchange != MFM_X_AUTO;
    if (g->exchange) {
        rc = load_rccl(&g->rccl);
        if (rc != MFM_OK) {
            group_release(g);
            return rc;
        }
        g->comm.assign(S, nullptr);
        const int nrc = g->rccl.CommInitAll(g->comm.data(), (int)S, g->dev.data());
        if (nrc != 0) {
            const char *why = g->rccl.GetErrorString ? g->rccl.GetErrorString(nrc) : "?";
            g->comm.clear();
            group_release(g);
            return gfail(MFM_E_DEVICE, "ncclCommInitAll over %zu devices failed: %s", S, why);
        }
        g->xs.assign(S, nullptr);
        g->xs[0] = static_cast<hipStream_t>(mfm_engine_copy_stream(g->eng[0]));
        for (size_t i = 1; i < S; i++) {
            if (hipSetDevice(g->dev[i]) != hipSuccess ||
                hipStreamCreateWithFlags(&g->xs[i], hipStreamNonBlocking) != hipSuccess) {
                group_release(g);
                return gfail(MFM_E_DEVICE, "cannot create the exchange stream on device %d", g->dev[i]);
            }
        }
    }
    if (g->exchange && (g->cfg.flags & MFM_F_TIMING)) {
        g->x_ms.assign(S, 0.0);
        g->x_n.assign(S, 0);
        for (size_t i = 0; i < S; i++) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipSetDevice(g->dev[i]) != hipSuccess || hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
                group_release(g);
                return gfail(MFM_E_DEVICE, "cannot create the exchange's events on device %d", g->dev[i]);
            }
            g->x0.push_back(a);
            g->x1.push_back(b);
        }
    }
    g->committed = true;
    return MFM_OK;
}

int mfm_group_nr_shards(struct mfm_group *g)
{
    return (g && g->committed) ? (int)g->eng.size() : gfail(MFM_E_STATE, "commit first");
}

int mfm_group_shard_info(struct mfm_group *g, uint32_t shard, uint32_t *first_channel, uint32_t *nr_channels, int32_t *device)
{
    if (!g || !g->committed || shard >= g->eng.size()) {
        return gfail(MFM_E_INVAL, "no such shard");
    }
    if (first_channel) {
        *first_channel = g->first[shard];
    }
    if (nr_channels) {
        *nr_channels = g->count[shard];
    }
    if (device) {
        *device = g->dev[shard];
    }
    return MFM_OK;
}

int mfm_group_exchange_detail(struct mfm_group *g, uint32_t shard, struct mfm_exchange_detail *out)
{
    if (!g || !out) {
        return gfail(MFM_E_INVAL, "NULL argument");
    }
    if (!g->committed || shard >= g->eng.size()) {
        return gfail(MFM_E_INVAL, "no such shard");
    }
    memset(out, 0, sizeof(*out));
    out->device = g->dev[shard];
    if (hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof(out->pci_bus_id), g->dev[shard]) != hipSuccess) {
        out->pci_bus_id[0] = 0;
    }
    if (g->exchange) {
        int n = -1;
        if (g->rccl.CommCount && shard < g->comm.size() && g->comm[shard]) {
            if (g->rccl.CommCount(g->comm[shard], &n) != 0) {
                n = -1;
            }
        }
        out->rccl_ranks = n;
    }
    {
        std::lock_guard<std::mutex> lk(g->x_mu);
        if (shard < g->x_n.size()) {
            out->timed_exchanges = g->x_n[shard];
            out->exchange_ms = g->x_ms[shard];
        }
    }
    mfm_stats st;
    memset(&st, 0, sizeof(st));
    if (mfm_engine_get_stats(g->eng[shard], &st) == MFM_OK) {
        out->timed_launches = st.timed_launches;
        out->kernel_ms = st.kernel_ms;
    }
    out->bound = MFM_BOUND_UNKNOWN;
    if (out->timed_exchanges && out->timed_launches) {
        const double x = out->exchange_ms / (double)out->timed_exchanges, k = out->kernel_ms / (double)out->timed_launches;
        out->bound = x > k ? MFM_BOUND_EXCHANGE : MFM_BOUND_KERNEL;
    }
    return MFM_OK;
}

int mfm_group_rccl_library(char *path, size_t cap)
{
    RcclApi api;
    const int rc = load_rccl(&api);
    if (rc != MFM_OK) {
        return rc;
    }
    if (path && cap) {
        snprintf(path, cap, "%s", g_rccl_path);
    }
    return MFM_OK;
}

} /* extern "C" */

namespace {

/* the group's shards behind the operations mfm_group_seq.h sequences */
struct GroupOps {
    mfm_group *g;
    bool pinned = false; /* the block lies in page-locked memory: the root's H2D reads it where it is */
    bool resident = false; /* the block already lies in the root's input buffer (mfm_group_acquire_input): nothing to stage */
    size_t shards() { return g->eng.size(); }
    int plan(size_t i, size_t n, bool *must, bool *may, bool *want)
    {
        int m = 0, y = 0, w = 0;
        const int rc = mfm_engine_plan(g->eng[i], n, &m, &y, &w);
        *must = m != 0;
        *may = y != 0;
        *want = w != 0;
        return rc;
    }
    bool conflict(size_t i, int fmt) { return 0 != mfm_engine_format_conflict(g->eng[i], fmt); }
    int unlaunched(size_t i) { return mfm_engine_pending_samples(g->eng[i]); }
    int flush(size_t i) { return mfm_engine_flush(g->eng[i]); }
    bool takes_bytes(size_t i, int fmt, size_t n) { return 0 != mfm_engine_can_take_bytes(g->eng[i], fmt, n); }
    int acquire(size_t i, bool raw, int fmt, void **dst, size_t *cap)
    {
        return raw ? mfm_engine_acquire_input_bytes(g->eng[i], fmt, dst, cap) : mfm_engine_acquire_input(g->eng[i], dst, cap);
    }
    int stage_root(const void *data, size_t n, int fmt, bool raw, void **d_root)
    {
        if (resident) {
            size_t cap = 0;
            const int rc = mfm_engine_acquire_input(g->eng[0], d_root, &cap);
            return rc != MFM_OK ? rc : cap < n ? gfail(MFM_E_INVAL, "the root's input buffer has room for %zu samples, the block has %zu", cap, n) : MFM_OK;
        }
        return mfm_engine_stage(g->eng[0], data, n, fmt, (raw ? MFM_STAGE_ALLOW_RAW : 0) | (pinned ? MFM_STAGE_PINNED : 0), d_root);
    }
    int nccl_fail(int nrc, const char *what, size_t bytes)
    {
        return gfail(MFM_E_DEVICE, "%s of a %zu-byte block failed: %s", what, bytes,
                     g->rccl.GetErrorString ? g->rccl.GetErrorString(nrc) : "?");
    }
    /* the wideband block, as bytes, from the root's input buffer into every member's input buffer.  One thread drives
     * all devices: the calls of every step sit in one RCCL group. */
    int exchange(void *d_root, void *const *dst, size_t bytes)
    {
        const size_t S = g->eng.size();
        const RcclApi &r = g->rccl;
        /* all-gather form: equal parts of whole 16-byte units; what does not divide goes by broadcast behind it */
        const size_t part = g->cfg.exchange == MFM_X_RCCL_ALLGATHER ? (bytes / S) & ~(size_t)15 : 0;
        const size_t even = part * S;
        int nrc = 0, nre = 0;
        const bool timed = fold_and_open_timing();
        struct Close {
            GroupOps *o;
            bool on;
            ~Close()
            {
                for (size_t i = 0; on && i < o->g->x1.size(); i++) {
                    (void)hipSetDevice(o->g->dev[i]);
                    (void)hipEventRecord(o->g->x1[i], o->g->xs[i]);
                }
            }
        } close_timing{ this, timed };
        if (part) {
            /* 1. part i of the block to peer i, over S - 1 different links at once (the root keeps part 0 where it is) */
            nrc = r.GroupStart();
            for (size_t i = 1; i < S && nrc == 0; i++) {
                nrc = r.Send(static_cast<const uint8_t *>(d_root) + i * part, part, kNcclInt8, (int)i, g->comm[0], g->xs[0]);
                nrc = nrc ? nrc : r.Recv(static_cast<uint8_t *>(dst[i]) + i * part, part, kNcclInt8, 0, g->comm[i], g->xs[i]);
            }
            nre = r.GroupEnd();
            nrc = nrc ? nrc : nre;
            if (nrc != 0) {
                return nccl_fail(nrc, "the scatter (ncclSend/ncclRecv)", bytes);
            }
            /* 2. all-gather in place: rank i contributes its part i, every rank's buffer ends up whole */
            nrc = r.GroupStart();
            for (size_t i = 0; i < S && nrc == 0; i++) {
                nrc = r.AllGather(static_cast<const uint8_t *>(dst[i]) + i * part, dst[i], part, kNcclInt8, g->comm[i], g->xs[i]);
            }
            nre = r.GroupEnd();
            nrc = nrc ? nrc : nre;
            if (nrc != 0) {
                return nccl_fail(nrc, "ncclAllGather", bytes);
            }
        }
        if (even < bytes) {
            nrc = r.GroupStart();
            for (size_t i = 0; i < S && nrc == 0; i++) {
                nrc = r.Broadcast(static_cast<const uint8_t *>(d_root) + even, static_cast<uint8_t *>(dst[i]) + even, bytes - even,
                                  kNcclInt8, 0, g->comm[i], g->xs[i]);
            }
            nre = r.GroupEnd();
            nrc = nrc ? nrc : nre;
            if (nrc != 0) {
                return nccl_fail(nrc, "ncclBroadcast", bytes);
            }
        }
        return MFM_OK;
    }
    /* the pair of the last bracketed exchange, once it is through, into the sums; every fourth exchange opens a new one */
    bool fold_and_open_timing()
    {
        if (g->x0.empty()) {
            return false;
        }
        if (g->x_open) {
            bool done = true;
            for (size_t i = 0; i < g->x1.size(); i++) {
                done = done && hipEventQuery(g->x1[i]) == hipSuccess;
            }
            if (!done) {
                g->x_seq++;
                return false; /* (still in flight: this exchange goes unbracketed) */
            }
            std::lock_guard<std::mutex> lk(g->x_mu);
            for (size_t i = 0; i < g->x1.size(); i++) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, g->x0[i], g->x1[i]) == hipSuccess) {
                    g->x_ms[i] += ms;
                    g->x_n[i]++;
                }
            }
            g->x_open = false;
        }
        if ((g->x_seq++ & 3u) != 0u) {
            return false;
        }
        for (size_t i = 0; i < g->x0.size(); i++) {
            (void)hipSetDevice(g->dev[i]);
            (void)hipEventRecord(g->x0[i], g->xs[i]);
        }
        g->x_open = true;
        return true;
    }
    int submit(size_t i, size_t n, bool launch)
    {
        return mfm_engine_submit_mode(g->eng[i], n, g->xs[i], 1, launch ? MFM_SUBMIT_LAUNCH : MFM_SUBMIT_DEFER);
    }
    int pending(size_t i) { return mfm_engine_pending_blocks(g->eng[i]); }
    int fetch(size_t i, mfm_block *blk) { return mfm_engine_fetch(g->eng[i], blk); }
    uint64_t first_output(const mfm_block &b) { return b.first_output; }
    size_t nr_outputs(const mfm_block &b) { return b.nr_outputs; }
    void lock() { g->mu.lock(); }
    void unlock() { g->mu.unlock(); }
    int fail(int code, const char *what, size_t shard) { return gfail(code, "%s (shard %zu)", what, shard); }
};

} /* namespace */

extern "C" {

/* ---- blocks that are in device memory already: the root's input buffer is where they are written (by a producer kernel, a
 *      peer-to-peer copy, another library's collective), the exchange and the submits are the same as for a host block ---- */

int mfm_group_acquire_input(struct mfm_group *g, void **d_dst, size_t *capacity_samples)
{
    if (!g || !d_dst) {
        return gfail(MFM_E_INVAL, "NULL argument");
    }
    if (!g->committed) {
        return gfail(MFM_E_STATE, "commit first");
    }
    return mfm_engine_acquire_input(g->eng[0], d_dst, capacity_samples);
}

int mfm_group_submit(struct mfm_group *g, size_t nr_samples)
{
    if (!g) {
        return gfail(MFM_E_INVAL, "NULL group");
    }
    if (!g->committed) {
        return gfail(MFM_E_STATE, "commit first");
    }
    if (!g->exchange) {
        return mfm_engine_submit(g->eng[0], nr_samples, nullptr, 0);
    }
    GroupOps ops{ g };
    ops.resident = true;
    size_t bytes = 0;
    const int rc = mfm_group_push_seq(ops, &g->broken, nullptr, nr_samples, MFM_IN_CS16, false, &bytes);
    if (rc == MFM_OK) {
        g->blocks++;
        g->bytes_exchanged += (uint64_t)bytes * (g->eng.size() - 1u);
    }
    return rc;
}

struct mfm_engine *mfm_group_shard_engine(struct mfm_group *g, uint32_t shard)
{
    return (g && g->committed && shard < g->eng.size()) ? g->eng[shard] : nullptr;
}

int mfm_group_push(struct mfm_group *g, const void *data, size_t nr_samples, int format)
{
    if (!g || !data) {
        return gfail(MFM_E_INVAL, "NULL argument");
    }
    if (!g->committed) {
        return gfail(MFM_E_STATE, "commit first");
    }
    if (!g->exchange) {
        return format == MFM_IN_CS16 ? mfm_engine_push(g->eng[0], static_cast<const int16_t *>(data), nr_samples)
                                     : mfm_engine_push_bytes(g->eng[0], data, nr_samples, format);
    }
    GroupOps ops{ g };
    size_t bytes = 0;
    const int rc = mfm_group_push_seq(ops, &g->broken, data, nr_samples, format, format != MFM_IN_CS16, &bytes);
    if (rc == MFM_OK) {
        g->blocks++;
        g->bytes_exchanged += bytes * (g->eng.size() - 1);
    }
    return rc;
}

int mfm_group_push_pinned(struct mfm_group *g, const void *data, size_t nr_samples, int format, uint64_t *ticket)
{
    if (!g || !data) {
        return gfail(MFM_E_INVAL, "NULL argument");
    }
    if (!g->committed) {
        return gfail(MFM_E_STATE, "commit first");
    }
    if (!g->exchange) {
        return mfm_engine_push_pinned(g->eng[0], data, nr_samples, format, ticket);
    }
    GroupOps ops{ g };
    ops.pinned = true;
    size_t bytes = 0;
    const int rc = mfm_group_push_seq(ops, &g->broken, data, nr_samples, format, format != MFM_IN_CS16, &bytes);
    if (rc == MFM_OK) {
        g->blocks++;
        g->bytes_exchanged += bytes * (g->eng.size() - 1);
        if (ticket) {
            *ticket = mfm_engine_copy_ticket(g->eng[0]);
        }
    }
    return rc;
}

/* a run of page-locked buffers that lie stride_bytes apart (mfm_engine_push_pinned_run): one copy command on a group of one
 * device; a group that exchanges takes the first buffer of the run the usual way (*accepted = 1) */
int mfm_group_push_pinned_run(struct mfm_group *g, const void *first, size_t stride_bytes, size_t nr_samples_each, size_t count,
                              int format, uint64_t *ticket, size_t *accepted)
{
    if (accepted) {
        *accepted = 0;
    }
    if (!g || !first || 0 == count) {
        return gfail(MFM_E_INVAL, "NULL or empty run");
    }
    if (!g->committed) {
        return gfail(MFM_E_STATE, "commit first");
    }
    if (!g->exchange) {
        return mfm_engine_push_pinned_run(g->eng[0], first, stride_bytes, nr_samples_each, count, format, ticket, accepted);
    }
    const int rc = mfm_group_push_pinned(g, first, nr_samples_each, format, ticket);
    if (rc == MFM_OK && accepted) {
        *accepted = 1;
    }
    return rc;
}

/* the producer loop of mfm_group_replay_pinned() over ONE arena of nr_bufs buffers stride_bytes apart, handing the group runs of
 * up to max_run neighbours at a time (what host/mfm_receiver.c's submit thread does with a backlog) */
int mfm_group_replay_arena(struct mfm_group *g, const void *arena, size_t stride_bytes, size_t nr_bufs, size_t buf_samples, int format,
                           size_t nr_pushes, size_t max_run, uint64_t *outputs_per_channel, uint64_t *copy_commands)
{
    if (!g || !arena || 0 == nr_bufs || 0 == max_run || !g->committed) {
        return gfail(MFM_E_INVAL, "bad argument");
    }
    std::vector<mfm_block> blks(g->eng.size());
    uint64_t outs = 0, cmds = 0;
    auto drain_one = [&]() -> int {
        const int rc = mfm_group_fetch(g, blks.data());
        if (rc != MFM_OK) {
            return rc;
        }
        outs += blks[0].nr_outputs;
        return mfm_group_release(g);
    };
    size_t i = 0;
    while (i < nr_pushes) {
        const size_t at = i % nr_bufs;
        const size_t want = std::min(std::min(max_run, nr_bufs - at), nr_pushes - i); /* neighbours up to the arena's end */
        size_t took = 0;
        const int rc = mfm_group_push_pinned_run(g, static_cast<const uint8_t *>(arena) + at * stride_bytes, stride_bytes, buf_samples,
                                                 want, format, nullptr, &took);
        if (rc == MFM_OK) {
            i += took;
            cmds++;
            continue;
        }
        if (rc != MFM_E_BUSY) {
            return rc;
        }
        const int dr = drain_one();
        if (dr != MFM_OK) {
            return dr == MFM_E_DONE ? gfail(MFM_E_STATE, "output rings full and nothing to fetch") : dr;
        }
    }
    for (;;) {
        const int frc = mfm_group_flush(g);
        if (frc != MFM_OK && frc != MFM_E_BUSY) {
            return frc;
        }
        int dr;
        while ((dr = drain_one()) == MFM_OK) {
        }
        if (dr != MFM_E_DONE) {
            return dr;
        }
        if (frc == MFM_OK) {
            break;
        }
    }
    if (outputs_per_channel) {
        *outputs_per_channel = outs;
    }
    if (copy_commands) {
        *copy_commands = cmds;
    }
    return MFM_OK;
}

/* A host loop in C, for measurements: nr_pushes buffers of buf_samples samples each, taken in turn from the caller's nr_bufs
 * page-locked buffers, through mfm_group_push_pinned(); whenever a push finds the output rings full a block is fetched and
 * released (its PCM has reached host memory by then), and at the end everything is flushed and drained.  What this repo's C
 * host does with two threads, in one - without a scripting language's call overhead between the buffers. */
int mfm_group_replay_pinned(struct mfm_group *g, const void *const *bufs, size_t nr_bufs, size_t buf_samples, int format,
                            size_t nr_pushes, uint64_t *outputs_per_channel)
{
    if (!g || !bufs || 0 == nr_bufs || !g->committed) {
        return gfail(MFM_E_INVAL, "bad argument");
    }
    std::vector<mfm_block> blks(g->eng.size());
    uint64_t outs = 0;
    auto drain_one = [&]() -> int {
        const int rc = mfm_group_fetch(g, blks.data());
        if (rc != MFM_OK) {
            return rc;
        }
        outs += blks[0].nr_outputs;
        return mfm_group_release(g);
    };
    for (size_t i = 0; i < nr_pushes; i++) {
        for (;;) {
            const int rc = mfm_group_push_pinned(g, bufs[i % nr_bufs], buf_samples, format, nullptr);
            if (rc == MFM_OK) {
                break;
            }
            if (rc != MFM_E_BUSY) {
                return rc;
            }
            const int dr = drain_one();
            if (dr != MFM_OK) {
                return dr == MFM_E_DONE ? gfail(MFM_E_STATE, "output rings full and nothing to fetch") : dr;
            }
        }
    }
    for (;;) {
        const int frc = mfm_group_flush(g);
        if (frc != MFM_OK && frc != MFM_E_BUSY) {
            return frc;
        }
        int dr;
        while ((dr = drain_one()) == MFM_OK) {
        }
        if (dr != MFM_E_DONE) {
            return dr;
        }
        if (frc == MFM_OK) {
            break;
        }
    }
    if (outputs_per_channel) {
        *outputs_per_channel = outs;
    }
    return MFM_OK;
}

int mfm_group_copy_done(struct mfm_group *g, uint64_t ticket)
{
    return (g && g->committed) ? mfm_engine_copy_done(g->eng[0], ticket) : gfail(MFM_E_STATE, "commit first");
}

int mfm_group_copy_wait(struct mfm_group *g, uint64_t ticket)
{
    return (g && g->committed) ? mfm_engine_copy_wait(g->eng[0], ticket) : gfail(MFM_E_STATE, "commit first");
}

int mfm_group_fetch(struct mfm_group *g, struct mfm_block *blks)
{
    if (!g || !blks) {
        return gfail(MFM_E_INVAL, "NULL argument");
    }
    if (!g->committed) {
        return gfail(MFM_E_STATE, "commit first");
    }
    GroupOps ops{ g };
    return mfm_group_fetch_seq(ops, &g->broken, blks);
}

int mfm_group_release(struct mfm_group *g)
{
    if (!g || !g->committed) {
        return gfail(MFM_E_STATE, "commit first");
    }
    int rc = MFM_OK;
    for (size_t i = 0; i < g->eng.size(); i++) {
        const int r = mfm_engine_release(g->eng[i]);
        rc = rc == MFM_OK ? r : rc;
    }
    return rc;
}

int mfm_group_flush(struct mfm_group *g)
{
    if (!g || !g->committed) {
        return gfail(MFM_E_STATE, "commit first");
    }
    if (!g->exchange) {
        return mfm_engine_flush(g->eng[0]);
    }
    GroupOps ops{ g };
    return mfm_group_flush_seq(ops, &g->broken);
}

int mfm_group_sync(struct mfm_group *g)
{
    if (!g || !g->committed) {
        return gfail(MFM_E_STATE, "commit first");
    }
    {
        const int rc = mfm_group_flush(g);
        if (rc != MFM_OK) {
            return rc;
        }
    }
    for (size_t i = 0; i < g->xs.size(); i++) {
        if (i > 0 && (hipSetDevice(g->dev[i]) != hipSuccess || hipStreamSynchronize(g->xs[i]) != hipSuccess)) {
            return gfail(MFM_E_DEVICE, "exchange stream of device %d failed", g->dev[i]);
        }
    }
    for (size_t i = 0; i < g->eng.size(); i++) {
        const int rc = mfm_engine_sync(g->eng[i]);
        if (rc != MFM_OK) {
            return rc;
        }
    }
    return MFM_OK;
}

int mfm_group_get_stats(struct mfm_group *g, uint32_t shard, struct mfm_stats *st)
{
    if (!g || !g->committed || shard >= g->eng.size()) {
        return gfail(MFM_E_INVAL, "no such shard");
    }
    return mfm_engine_get_stats(g->eng[shard], st);
}

int mfm_group_exchange_info(struct mfm_group *g, int *uses_rccl, uint64_t *blocks, uint64_t *bytes_exchanged)
{
    if (!g || !g->committed) {
        return gfail(MFM_E_STATE, "commit first");
    }
    if (uses_rccl) {
        *uses_rccl = g->exchange ? 1 : 0;
    }
    if (blocks) {
        *blocks = g->blocks;
    }
    if (bytes_exchanged) {
        *bytes_exchanged = g->bytes_exchanged;
    }
    return MFM_OK;
}

} /* extern "C" */
