/* CPU test of the device group's push / fetch ORDER (tsl-sdr_amd/csrc/mfm_group_seq.h) over fake shards that refuse or fail
 * at chosen steps, and of the fetch-against-push race with two threads.  Test infrastructure only; no GPU, no arithmetic.
 * Exit code 0 = every check held; prints the first failed check otherwise. */
#include "../../tsl-sdr_amd/csrc/mfm_group_seq.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>


struct Blk {
    uint64_t first;
    size_t n;
};

struct Fake {
    size_t S = 4;
    /* where to fail: step name -> (shard, code) */
    int room_fail_shard = -1, acquire_fail_shard = -1, submit_fail_shard = -1;
    int stage_fail = 0, exchange_fail = 0;
    /* what happened */
    int rooms = 0, acquires = 0, stages = 0, exchanges = 0;
    std::vector<std::atomic<int>> submitted, released;
    std::mutex mu;
    int slow_submit_us = 0;
    Fake() : submitted(16), released(16)
    {
        for (auto &a : submitted) a = 0;
        for (auto &a : released) a = 0;
    }
    size_t shards() { return S; }
    /* a shard whose output ring is full may not launch; `coalesce` = the shards need not launch with every block */
    bool coalesce = false, want_launch = false;
    std::vector<int> unl = std::vector<int>(16, 0);
    int launches = 0, flushes = 0;
    int plan(size_t i, size_t, bool *must, bool *may, bool *want)
    {
        rooms++;
        *must = !coalesce;
        *may = (int)i != room_fail_shard;
        *want = want_launch;
        return MFM_OK;
    }
    int conflict_fmt = -1;
    bool conflict(size_t i, int fmt) { return unl[i] > 0 && conflict_fmt >= 0 && fmt != conflict_fmt; }
    int unlaunched(size_t i) { return unl[i]; }
    int flush(size_t i)
    {
        if (unl[i] > 0) { unl[i] = 0; submitted[i]++; flushes++; }
        return MFM_OK;
    }
    bool takes_bytes(size_t i, int, size_t) { return i != 2; } /* shard 2 cannot: the block must travel widened */
    bool last_raw = false;
    int acquire(size_t i, bool raw, int, void **dst, size_t *cap)
    {
        acquires++;
        last_raw = raw;
        if ((int)i == acquire_fail_shard) return MFM_E_DEVICE;
        *dst = (void *)(uintptr_t)(0x1000 * (i + 1));
        *cap = 1 << 20;
        return MFM_OK;
    }
    int stage_root(const void *, size_t, int, bool, void **d)
    {
        stages++;
        if (stage_fail) return stage_fail;
        *d = (void *)(uintptr_t)0x1000;
        return MFM_OK;
    }
    int exchange(void *, void *const *, size_t) { exchanges++; return exchange_fail; }
    int submit(size_t i, size_t n, bool launch)
    {
        if ((int)i == submit_fail_shard) return MFM_E_DEVICE;
        if (slow_submit_us) std::this_thread::sleep_for(std::chrono::microseconds(slow_submit_us));
        if (launch) {
            unl[i] = 0;
            submitted[i]++; /* one finished-or-running block per LAUNCH */
            launches++;
        } else {
            unl[i] += (int)n;
        }
        return MFM_OK;
    }
    int pending(size_t i) { return submitted[i] - released[i]; }
    int fetch(size_t i, Blk *b)
    {
        if (pending(i) <= 0) return MFM_E_DONE;
        b->first = (uint64_t)released[i] * 100;
        b->n = 100;
        return MFM_OK;
    }
    uint64_t first_output(const Blk &b) { return b.first; }
    size_t nr_outputs(const Blk &b) { return b.n; }
    void lock() { mu.lock(); }
    void unlock() { mu.unlock(); }
    int fail(int code, const char *, size_t) { return code; }
    int total_submits() { int t = 0; for (size_t i = 0; i < S; i++) t += submitted[i]; return t; }
};

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main()
{
    char data[16] = { 0 };
    size_t bytes = 0;
    { /* the good path; an 8-bit block travels widened because one shard cannot read bytes */
        Fake f; bool broken = false;
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 3, true, &bytes) == MFM_OK);
        CHECK(!broken && f.total_submits() == 4 && f.stages == 1 && f.exchanges == 1 && bytes == 4000 && !f.last_raw);
        Fake h; h.S = 2; /* shards 0, 1 both read bytes */
        CHECK(mfm_group_push_seq(h, &broken, data, 1000, 3, true, &bytes) == MFM_OK && bytes == 2000 && h.last_raw);
    }
    { /* a shard without room: MFM_E_BUSY before anything was named, staged or submitted - and the group is fine */
        Fake f; bool broken = false; f.room_fail_shard = 3;
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_E_BUSY);
        CHECK(!broken && f.acquires == 0 && f.stages == 0 && f.exchanges == 0 && f.total_submits() == 0);
        f.room_fail_shard = -1;
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_OK && f.total_submits() == 4);
    }
    { /* a shard that refuses its input buffer mid-push: the root has not staged, nobody has submitted */
        Fake f; bool broken = false; f.acquire_fail_shard = 2;
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_E_DEVICE);
        CHECK(!broken && f.stages == 0 && f.exchanges == 0 && f.total_submits() == 0);
    }
    { /* the root's copy fails: nothing exchanged, nothing submitted, not broken */
        Fake f; bool broken = false; f.stage_fail = MFM_E_DEVICE;
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_E_DEVICE);
        CHECK(!broken && f.exchanges == 0 && f.total_submits() == 0);
    }
    { /* the collective fails: nothing submitted; the group refuses from then on */
        Fake f; bool broken = false; f.exchange_fail = MFM_E_DEVICE;
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_E_DEVICE);
        CHECK(broken && f.total_submits() == 0);
        f.exchange_fail = 0;
        const int before = f.rooms;
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_E_DEVICE && f.rooms == before);
        Blk b[4];
        CHECK(mfm_group_fetch_seq(f, &broken, b) == MFM_E_DEVICE);
    }
    { /* a submit fails after two shards took the block: broken, every later call refuses (no drifting apart) */
        Fake f; bool broken = false; f.submit_fail_shard = 2;
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_E_DEVICE);
        CHECK(broken && f.total_submits() == 2);
        f.submit_fail_shard = -1;
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_E_DEVICE && f.total_submits() == 2);
    }
    { /* coalescing shards: a block that need not be launched is accepted by all and launched by none; the root's wish
       * launches it on all; a full ring on ONE shard defers it on all (nobody must) - and refuses it when one must */
        Fake f; bool broken = false; f.coalesce = true; Blk b[4];
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_OK);
        CHECK(f.launches == 0 && f.unl[0] == 1000 && f.unl[3] == 1000 && mfm_group_fetch_seq(f, &broken, b) == MFM_E_DONE);
        f.want_launch = true;
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_OK && f.launches == 4 && f.unl[2] == 0);
        CHECK(mfm_group_fetch_seq(f, &broken, b) == MFM_OK);
        f.room_fail_shard = 1; /* shard 1's ring is full */
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_OK && f.launches == 4 && f.unl[1] == 1000);
        CHECK(mfm_group_flush_seq(f, &broken) == MFM_E_BUSY && f.flushes == 0 && !broken);
        f.coalesce = false; /* now every shard must launch with the next block: refused, nothing accepted */
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_E_BUSY && f.unl[0] == 1000 && f.stages == 3);
        f.room_fail_shard = -1;
        CHECK(mfm_group_flush_seq(f, &broken) == MFM_OK && f.flushes == 4 && f.unl[0] == 0);
        /* a block of another sample format behind accepted ones: those go out first, on every shard */
        f.coalesce = true; f.want_launch = false;
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_OK && f.unl[0] == 1000);
        f.conflict_fmt = 0; f.S = 2; /* shards 0, 1 read bytes: the 8-bit block would stay bytes */
        CHECK(mfm_group_push_seq(f, &broken, data, 500, 3, true, &bytes) == MFM_OK && f.flushes == 6 && f.unl[0] == 500);
    }
    { /* fetch: nothing pending = MFM_E_DONE; a block = one per shard, same position */
        Fake f; bool broken = false; Blk b[4];
        CHECK(mfm_group_fetch_seq(f, &broken, b) == MFM_E_DONE);
        CHECK(mfm_group_push_seq(f, &broken, data, 1000, 0, false, &bytes) == MFM_OK);
        CHECK(mfm_group_fetch_seq(f, &broken, b) == MFM_OK && b[3].first == b[0].first);
    }
    { /* two threads, small blocks: the consumer polls fetch while the producer sits between two submits (slow submits).
       * It must only ever see MFM_E_DONE or a complete block - never "shards out of step". */
        Fake f; bool broken = false; f.S = 2; f.slow_submit_us = 50;
        std::atomic<bool> stop{ false };
        std::atomic<int> fetched{ 0 }, bad{ 0 };
        std::thread consumer([&] {
            Blk b[2];
            while (!stop || fetched < 300) {
                const int rc = mfm_group_fetch_seq(f, &broken, b);
                if (rc == MFM_OK) {
                    if (b[0].first != b[1].first) bad++;
                    f.released[0]++;
                    f.released[1]++;
                    fetched++;
                } else if (rc != MFM_E_DONE) {
                    bad++;
                    break;
                }
                if (fetched >= 300) break;
            }
        });
        for (int k = 0; k < 300; k++) {
            if (mfm_group_push_seq(f, &broken, data, 64, 0, false, &bytes) != MFM_OK) bad++;
        }
        stop = true;
        consumer.join();
        CHECK(bad == 0 && fetched == 300 && !broken);
    }
    { /* `broken` is written by the push thread (a collective that fails, a submit that fails half-way) while
       * the fetch thread reads it - an atomic flag, exercised here under ThreadSanitizer: a consumer polling fetch while the
       * producer's exchange starts to fail must see blocks, "nothing yet", and from some point on only MFM_E_DEVICE */
        Fake f; std::atomic<bool> broken{ false }; f.S = 2;
        std::atomic<bool> stop{ false };
        std::atomic<int> ok{ 0 }, dev{ 0 }, bad{ 0 };
        std::thread consumer([&] {
            Blk b[2];
            bool seen_broken = false;
            while (!stop) {
                const int rc = mfm_group_fetch_seq(f, &broken, b);
                if (rc == MFM_OK) {
                    if (seen_broken) bad++;
                    f.released[0]++;
                    f.released[1]++;
                    ok++;
                } else if (rc == MFM_E_DEVICE) {
                    seen_broken = true;
                    dev++;
                } else if (rc != MFM_E_DONE) {
                    bad++;
                }
            }
        });
        for (int k = 0; k < 200; k++) {
            if (k == 120) f.exchange_fail = MFM_E_DEVICE;
            const int rc = mfm_group_push_seq(f, &broken, data, 64, 0, false, &bytes);
            if ((k < 120) != (rc == MFM_OK)) bad++;
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
        stop = true;
        consumer.join();
        CHECK(bad == 0 && broken && dev > 0 && f.total_submits() == 240);
    }
    printf("group sequence: all checks held\n");
    return 0;
}
