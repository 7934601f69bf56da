/*
 * mfm_engine.hip - host side of the MI355X multifm channel engine behind include/multifm_hip.h.
 *
 * Owns: the channel set (Q14 taps, rotator tables), two device input staging buffers holding
 * [history tail | new block] contiguously, a small ring of output slots (device + pinned host
 * mirror), the per-channel carry state (rotator index, last filtered sample) and three HIP
 * streams (copy-in, compute, copy-out) so H2D, the fused kernel and D2H of consecutive blocks
 * overlap.  What the reference keeps per channel thread in struct direct_fir / struct
 * multifm_fm_demod (filter/direct_fir.h:9-74, multifm/fm_demod.c:14-18) lives here per engine.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/multifm_hip.h"
#include "mfm_kernel.h"
#include "mfm_numerics.h"
#include "mfm_taps.h"
#include "mfm_engine_internal.h"

/* every kernel file exports a pair: select (which template instance runs a launch description; asked at commit, where the
 * instance's LDS limit is raised once) and launch (through that pointer) */
extern "C" hipError_t mfm_select_channel_kernel(int opl, int dbg_iq, const void **kfn_out);
extern "C" hipError_t mfm_launch_channel_kernel(const void *kfn, const mfm_launch *L, uint32_t lds_bytes, hipStream_t stream);
extern "C" hipError_t mfm_select_channel_kernel_mfma(const mfm_launch_mfma *L, int dbg_iq, const void **kfn_out,
                                                     uint32_t *waves_per_simd_out);
extern "C" hipError_t mfm_launch_channel_kernel_mfma(const void *kfn, const mfm_launch_mfma *L, uint32_t lds_bytes,
                                                     uint32_t grid, hipStream_t stream);
extern "C" hipError_t mfm_disc_test_dot2(const int *s_re, const int *s_im, int *pcm, uint32_t n, const float2 *lut, hipStream_t stream);
extern "C" hipError_t mfm_disc_test_mfma(const int *s_re, const int *s_im, int *pcm, uint32_t n, const float2 *lut, hipStream_t stream);
extern "C" hipError_t mfm_disc_test_v3(const int *s_re, const int *s_im, int *pcm, uint32_t n, const float2 *lut, hipStream_t stream);
extern "C" hipError_t mfm_select_channel_kernel_v3(const mfm_launch_v3 *L, int dbg_iq, const void **kfn_out);
extern "C" uint32_t mfm_rot_entry_bytes_v3(void);
extern "C" uint32_t mfm_sp_pitch_v3(void);
extern "C" uint32_t mfm_v3l_wg_per_cu(const mfm_launch_v3 *L);
extern "C" hipError_t mfm_launch_channel_kernel_v3(const void *kfn, const mfm_launch_v3 *L, uint32_t lds_bytes, uint32_t grid,
                                                   hipStream_t stream);


namespace {

thread_local char g_last_error[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t err_ = (expr);                                                                            \
        if (err_ != hipSuccess) {                                                                            \
            return fail(MFM_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err_), __FILE__,     \
                        __LINE__);                                                                           \
        }                                                                                                    \
    } while (0)

/* FNV-1a of the 257 float bit patterns of the fast_atan2f table */
constexpr uint64_t MFM_ATAN_TABLE_FNV1A = 0x674d1aab1787b44bull;

constexpr int kOutSlots = 4;      /* output ring depth (2 in device-only mode) */
constexpr int kMaxInBufs = 3;     /* input buffers: 2, or 3 when the engine coalesces (mfm_engine_config::coalesce_samples) */
constexpr int kTimingPairs = 256; /* event pairs kept before the oldest is folded into the total */
constexpr uint64_t kSparseTiming = 4; /* MFM_F_TIMING_SPARSE: one launch in this many is bracketed */
constexpr uint64_t kCycleRing = 1024; /* launches whose shader-clock stamps are kept (mfm_engine_get_launch_cycles) */
constexpr size_t kLaunchRing = 4096; /* per-launch durations kept for mfm_engine_get_launch_ms() */
constexpr uint32_t kMaxOutputsPerTile = 128;
/* 128-tap filters: slices of 128 channels from this many channels on (below, slices of 64: MFM_F_SLICE_128 / _64 override) */
constexpr uint32_t kSlice128MinChannels = 512; /* the measured crossover (profiles/r06_slice128_ab.txt): +1.8 % at 128, +1.7 % at 256,
                                                 -0.6 % at 512, -0.8 % at 768, -1.4 % at 1024 channels */
/* second-generation kernels: PCM stores with system scope from this many channels per launch on (profiles/r05_store_policy.txt,
 * r06_hbm_traffic_1024ch.json: L2-miss traffic 1.27 -> 1.16 x algorithmic at 1024 channels at unchanged time; +0.4 % time at 256
 * channels and 1.5-5 % at 64, where there is nothing to gain: profiles/r06_ab_store_policy.txt) */
constexpr uint32_t kPcmWriteThroughMinChannels = 512;
constexpr uint64_t kMaxRotEntries = 1ull << 26; /* per distinct increment: 512 MiB of table */

struct Channel {
    std::vector<int16_t> cre, cim;
    int16_t incr_re = 0, incr_im = 0;
    bool want_iq = false;
    /* rotator table placement */
    uint64_t rot_base = 0;
    uint32_t mu = 0, lam = 1;
};

inline void rot_step(int16_t &rr, int16_t &ri, int16_t ir, int16_t ii)
{
    /* filter/direct_fir.c:166-167 -> filter/complex.h:51-62 */
    const int32_t a = rr, b = ri;
    const int16_t nr = (int16_t)mfm_r14_wide(a * ir - b * ii);
    const int16_t ni = (int16_t)mfm_r14_wide(a * ii + b * ir);
    rr = nr;
    ri = ni;
}

/* Brent's cycle finder on the rotator recurrence started at (16384, 0) (direct_fir.c:78-79). */
bool rot_cycle(int16_t ir, int16_t ii, uint64_t limit, uint32_t *mu_out, uint32_t *lam_out)
{
    int16_t tr = 16384, ti = 0, hr = 16384, hi = 0;
    uint64_t power = 1, lam = 1;
    rot_step(hr, hi, ir, ii);
    while (!(tr == hr && ti == hi)) {
        if (power == lam) {
            tr = hr;
            ti = hi;
            power *= 2;
            lam = 0;
        }
        rot_step(hr, hi, ir, ii);
        lam++;
        if (lam > limit) {
            return false;
        }
    }
    tr = 16384, ti = 0, hr = 16384, hi = 0;
    for (uint64_t i = 0; i < lam; i++) {
        rot_step(hr, hi, ir, ii);
    }
    uint64_t mu = 0;
    while (!(tr == hr && ti == hi)) {
        rot_step(tr, ti, ir, ii);
        rot_step(hr, hi, ir, ii);
        mu++;
        if (mu > limit) {
            return false;
        }
    }
    *mu_out = (uint32_t)mu;
    *lam_out = (uint32_t)lam;
    return true;
}

struct OutSlot {
    int16_t *d_pcm = nullptr;
    uint32_t *d_iq = nullptr;
    int16_t *h_pcm = nullptr;
    uint32_t *h_iq = nullptr;
    hipEvent_t ready = nullptr; /* D2H (or, device-only, the kernel) finished */
    uint64_t first_output = 0;
    uint32_t nr_outputs = 0;
    enum { FREE, INFLIGHT, FETCHED } state = FREE;
};

} /* namespace */

struct mfm_engine {
    mfm_engine_config cfg{};
    std::vector<Channel> chans;
    uint32_t nr_taps = 0;
    bool committed = false;
    bool any_iq = false;

    /* geometry */
    int opl = 2;
    uint32_t rs2 = 0, lds_bytes = 0, lut_off = 0, nchunks = 0, ngroups = 0, gpw = 0, nslices = 0;
    uint32_t cap_in = 0;     /* samples per input buffer */
    uint32_t out_stride = 0; /* outputs per channel one submit can produce (even) */

    /* matrix-core path (mfm_kernel_mfma.hip) */
    bool use_mfma = false;
    uint32_t m_row_bytes = 0, m_nstage = 0;
    bool slice128 = false;        /* 128-tap filters on 128-channel slices (layout 3, two row blocks per wave) */
    uint32_t m_ks = 0, m_ot = 0, m_rs = 0, m_plane_bytes = 0, m_lut_off = 0, m_krow_off = 0, m_nrb = 0,
             m_nslices = 0, m_lds_bytes = 0, m_wg_per_cu = 1;
    uint32_t m_wg_fmt[4] = { 1, 1, 1, 1 }; /* workgroups per CU of the instance each input format runs (select_kernels) */
    bool m_resident_taps = false;
    bool m_fixed_planes = false;
    uint32_t m_ah_mask = 0; /* k-steps whose high-byte tap plane is not all zero */
    uint32_t m_kq_used = 0; /* k-steps that hold taps at all */
    /* second-generation matrix kernel (mfm_kernel_v3.hip): same tap fragments, its own LDS image */
    bool use_v3 = false;
    uint32_t v_rs = 0, v_sp_pitch = 0, v_nstage4 = 0, v_lds_bytes = 0, v_wg_per_cu = 1, v_cross[4] = { 0, 0, 0, 0 },
             v_within[4] = { 0, 0, 0, 0 };
    uint32_t v_layout = 0, v_t_per = 0, v_t_pitch = 0; /* mfm_launch_v3::layout: chunk rows for decimations % 32 != 0 */
    /* layout 3, long filters (mfm_kernel_v3l.hip): plain rows; v_rs = row stride, v_plane = bytes of one byte plane */
    uint32_t v_plane = 0, v_ng = 0, v_nstage_p = 0, v_sta_bytes = 0, v_rb = 1;
    uint32_t v_shift = 0, v_copy_pitch = 0; /* decimations 1, 2, 4: 8 / D shifted copies of the image, this many bytes apart */
    uint32_t v_nslices = 0; /* channel slices of the second-generation launches: of 64 channels, or of 128 (v_rb = 2) */
    uint32_t v_kq = 0, v_nh = 0, v_kperm[4] = { 0, 0, 0, 0 }; /* the instance's k-step count, k-steps with a high-byte tap plane,
                                                                 and the order the k-steps are laid out in (mfm_launch_v3::kperm) */
    uint32_t *d_afrag = nullptr;
    int32_t *d_krow = nullptr;
    int32_t *d_krow8[4] = { nullptr, nullptr, nullptr, nullptr }; /* [MFM_IN_*]: row constants of the 8-bit input forms */
    bool raw8_ok = false; /* the matrix kernels can read 8-bit input as it is (IN8 forms of mfm_kernel_v3.hip, mfm_kernel_mfma.hip) */
    uint32_t v_rc = 0; /* the lowest rotator class (MFM_RC_*) among the channels: selects the kernel instance */
    uint32_t rot_exact_channels = 0, rot_fast_slices = 0;
    /* the kernel instance that runs a block of input format MFM_IN_* (selected, and its LDS limit raised, at commit) */
    const void *kfn[4] = { nullptr, nullptr, nullptr, nullptr };

    /* device tables */
    uint32_t *d_coef = nullptr, *d_tapoff = nullptr;
    mfm_chan_info *d_info = nullptr;
    uint2 *d_rot = nullptr;
    float2 *d_lut = nullptr;
    mfm_chan_state *d_state[2] = { nullptr, nullptr };
    uint64_t rot_entries = 0;

    /* input staging: nbuf buffers used in turn, each [history tail | blocks accepted since the last launch] */
    uint32_t *d_in[kMaxInBufs] = { nullptr, nullptr, nullptr };
    bool own_in = false;
    uint32_t *h_in[kMaxInBufs] = { nullptr, nullptr, nullptr }; /* pinned, for push() */
    uint16_t *d_raw[kMaxInBufs] = { nullptr, nullptr, nullptr }; /* push_bytes(): 8-bit IQ pairs as they came off the wire */
    hipEvent_t in_free[kMaxInBufs] = { nullptr, nullptr, nullptr };
    hipEvent_t in_free_wait[kMaxInBufs] = { nullptr, nullptr, nullptr }; /* the end of the launch that last read buffer i:
                                                          in_free[i], or the timing event recorded at the same point of
                                                          the stream.  acquire_input() waits on it, the coalescing policy
                                                          queries it */
    hipEvent_t in_ready = nullptr;
    int nbuf = 2;      /* 3 with coalesce_samples: one being read, one queued behind it, one being filled */
    int cur_in = 0;
    uint32_t tail = 0; /* unconsumed samples at the front of d_in[cur_in] (behind `hist`): the history the next outputs need */
    uint32_t hist = 0; /* second-generation kernel: already consumed samples kept in front of them - the decimation once the
                          stream has produced an output, 0 before -, from which a launch recomputes the output in front of
                          it instead of reading carried state (mfm_launch_v3::hist) */
    uint32_t pend = 0; /* samples accepted into d_in[cur_in] behind the history and not yet launched (coalesce_samples) */
    uint32_t last_launch_samples = 0; /* what the most recent launch read: a second launch is queued behind one in flight
                                         once a quarter of that has gathered */
    uint64_t submits = 0;
    /* pushes straight out of the caller's pinned memory (mfm_engine_push_pinned): a ticket per push, an event per ticket */
    /* Copies complete in order (one copy stream), so a ticket is through as soon as ANY event recorded behind it is: events
     * are recorded only when somebody asks about a ticket no recorded event covers yet (an event per push cost more host
     * time than the staging copy it replaced), a few of them in rotation. */
    static constexpr uint64_t kCopyRing = 4;
    hipEvent_t copy_ev[kCopyRing] = {};
    uint64_t copy_ev_seq[kCopyRing] = {}; /* the ticket each of them covers (0: not recorded) */
    uint64_t copy_ev_next = 0;
    uint64_t copy_seq = 0, copy_done_seq = 0;
    bool wait_copy_stream = false; /* blocks staged on the engine's copy stream since the last launch: the launch waits for
                                      that stream once, instead of every push recording and waiting for an event */
    int last_launch_buf = -1, last_launch_fmt = MFM_IN_CS16;
    uint32_t last_launch_hist = 0;
    /* 8-bit blocks may sit in the input buffers as they came off the wire (2 bytes per sample, push_bytes): the format of
     * what was staged into each buffer, and of the history at the front of d_in[cur_in] */
    int in_fmt[kMaxInBufs] = { MFM_IN_CS16, MFM_IN_CS16, MFM_IN_CS16 };
    int tail_fmt = MFM_IN_CS16;
    uint16_t *d_tailtmp = nullptr; /* a history kept as bytes is widened through here when an int16 block follows it */
    uint64_t launches_8bit = 0;

    /* outputs */
    OutSlot slots[kOutSlots];
    int nslots = kOutSlots;
    uint64_t submit_seq = 0; /* submits that produced outputs */
    uint64_t fetch_seq = 0;
    int last_slot = -1;

    hipStream_t s_in = nullptr, s_compute = nullptr, s_out = nullptr;
    /* MFM_F_OVERLAP: launches alternate between two compute streams (cs[0] = s_compute).  A launch of the second-generation
     * kernel depends on the one before it through input samples only (mfm_launch_v3::hist, ::k_base), so launch k + 1 fills
     * the workgroup slots launch k's shorter chunks free up instead of waiting for its last tile - and for a dispatch. */
    hipStream_t cs[2] = { nullptr, nullptr };
    uint32_t ncs = 1;
    uint32_t si = 0;              /* index into cs of the next launch */
    hipStream_t s_last = nullptr; /* the stream the most recent launch went to (mfm_engine_stream) */
    hipEvent_t tail_done[kMaxInBufs] = { nullptr, nullptr, nullptr }; /* overlap: the carry out of buffer i has been copied */
    bool tail_pending[kMaxInBufs] = { false, false, false };
    hipEvent_t kernel_done = nullptr;

    /* stream bookkeeping */
    int parity = 0;
    uint64_t samples_in = 0, outputs = 0, launches = 0;
    uint32_t grid_last = 0;

    /* one producer thread (push/submit) and one consumer thread (fetch/release) may use the engine
     * concurrently: this guards the bookkeeping they share; device waits happen outside it */
    std::mutex mu;

    /* timing */
    hipEvent_t t0[kTimingPairs] = {}, t1[kTimingPairs] = {};
    bool timing_events = false;
    uint64_t t_head = 0, t_tail = 0;
    double kernel_ms = 0.0;
    /* MFM_F_TIMING, second-generation kernels: per-launch durations in the shader's own clocks (mfm_launch_v3::cyc) */
    unsigned long long *d_cyc = nullptr; /* [kCycleRing][2] */
    std::vector<float> launch_ms; /* ring of the last kLaunchRing launch durations */
    uint64_t launch_ms_n = 0;     /* durations folded so far */
};

namespace {

/* The geometry half of a launch description - everything commit fixed, per input format; submit adds the block (input
 * address and counts, chunking, output slot, carried state).  Also what the kernel files' select functions look at. */
void fill_v3(const mfm_engine *e, int fmt, mfm_launch_v3 &V)
{
    V.decim = e->cfg.decimation;
    V.x_last4 = e->v_shift ? e->cap_in - 1u : (e->cap_in - 4u) & ~3u;
    V.kq = e->m_ks;
    V.rs = e->v_rs;
    V.sp_pitch = e->v_sp_pitch;
    V.plane_pitch = 4u * e->v_sp_pitch;
    V.buf_pitch = 8u * e->v_sp_pitch;
    V.nstage4 = e->v_nstage4;
    V.lut_off = 16u * e->v_sp_pitch;
    V.sta_off = V.lut_off + 2048u;
    for (int k = 0; k < 4; k++) {
        V.cross[k] = e->v_cross[k];
        V.within[k] = e->v_within[k];
    }
    V.layout = e->v_layout;
    V.t_per = e->v_t_per;
    V.t_pitch = e->v_t_pitch;
    if (3u == e->v_layout) {
        /* long filters: [two buffers of two byte planes | atan table | staging offsets | transposition areas + per-wave constants] */
        V.plane_pitch = e->v_plane;
        V.buf_pitch = 2u * e->v_plane;
        V.lut_off = 4u * e->v_plane;
        V.sta_off = V.lut_off + 2048u;
        V.tp_off = V.sta_off + e->v_sta_bytes;
        V.row_bytes = e->m_row_bytes;
        V.split_rows = (e->cfg.decimation % 4u) != 0u ? 1u : 0u;
        V.kq = e->v_kq;
        V.kq_used = e->m_kq_used;
        V.nh = e->v_nh;
        for (int k = 0; k < 4; k++) {
            V.kperm[k] = e->v_kperm[k];
        }
        V.ng = e->v_ng;
        V.rb = e->v_rb;
        V.nstage_p = e->v_nstage_p;
        V.shift = e->v_shift;
        if (e->v_shift) {
            V.sp_pitch = e->v_copy_pitch;
        }
    }
    V.nslices = e->v_nslices;
    V.nrb = e->m_nrb;
    V.nchan = (uint32_t)e->chans.size();
    V.pcm_scope = (V.nchan >= kPcmWriteThroughMinChannels && !(e->cfg.flags & MFM_F_PCM_WRITE_BACK)) ? 1u : 0u;
    V.out_stride = e->out_stride;
    V.ah_mask = e->m_ah_mask;
    V.rc = e->v_rc;
    V.afrag = e->d_afrag;
    V.krow = e->d_krow;
    V.info = e->d_info;
    V.rot = e->d_rot;
    V.lut = e->d_lut;
    if (fmt != MFM_IN_CS16) {
        /* the buffer holds 2-byte samples: twice as many fit, a 16-byte chunk is 8 of them */
        V.in8 = fmt == MFM_IN_RTLSDR_U8 ? 7u : 14u;
        V.in8_xor = fmt == MFM_IN_RTLSDR_U8 ? 0x80808080u : 0u;
        V.krow = e->d_krow8[fmt];
        V.nstage4 = 2u == e->v_layout ? (73u * 25u + 7u + 7u) / 8u : e->v_nstage4 / 2u;
        V.x_last4 = (2u * e->cap_in - 8u) & ~7u;
        if (3u == e->v_layout) {
            /* a staging chunk stays 4 samples there - an 8-byte load */
            V.nstage4 = e->v_nstage4;
            V.x_last4 = (2u * e->cap_in - 4u) & ~3u;
            if (e->v_shift) {
                V.x_last4 = 2u * e->cap_in - 1u; /* one-sample loads */
            }
        }
    }
}

void fill_mfma(const mfm_engine *e, int fmt, mfm_launch_mfma &M)
{
    const uint32_t C = (uint32_t)e->chans.size(), D = e->cfg.decimation;
    M.decim = D;
    M.x_last4 = (e->cap_in - 4u) & ~3u;
    M.kq = e->m_ks;
    M.kq_used = e->m_kq_used;
    M.ot = e->m_ot;
    M.nstage = e->m_nstage;
    M.rs = e->m_rs;
    M.row_bytes = e->m_row_bytes;
    M.split_rows = (D % 4u) != 0u ? 1u : 0u;
    M.plane_bytes = e->m_plane_bytes;
    M.fixed_planes = e->m_fixed_planes ? 1u : 0u;
    M.lut_off = e->m_lut_off;
    M.sta_off = e->m_lut_off + 2048u;
    M.bof_off = M.sta_off + ((M.nstage / 4u + MFM_MFMA_NW * 64u - 1u) / (MFM_MFMA_NW * 64u)) * MFM_MFMA_NW * 64u * 4u;
    M.tbl_off = C <= 256u ? M.bof_off + (e->m_ks > MFM_MFMA_KQ_MAX ? 2048u : 0u) : 0u;
    M.nslices = e->m_nslices;
    M.nrb = e->m_nrb;
    M.nchan = C;
    M.out_stride = e->out_stride;
    M.ah_mask = e->m_ah_mask;
    M.stream_taps = (e->cfg.flags & MFM_F_STREAM_TAPS) ? 1u : 0u;
    M.afrag = e->d_afrag;
    M.krow = e->d_krow;
    M.info = e->d_info;
    M.rot = e->d_rot;
    M.lut = e->d_lut;
    if (fmt != MFM_IN_CS16) {
        /* the buffer holds 2-byte samples: twice as many fit; a staging chunk (4 samples) is an 8-byte load */
        M.in8 = fmt == MFM_IN_RTLSDR_U8 ? 7u : 14u;
        M.in8_xor = fmt == MFM_IN_RTLSDR_U8 ? 0x80808080u : 0u;
        M.krow = e->d_krow8[fmt];
        M.x_last4 = (2u * e->cap_in - 4u) & ~3u;
    }
}

/* The dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) belongs to a kernel instance on a device, not to an
 * engine: two engines of one process can share an instance with different image sizes (first-generation kernels with
 * different decimations; the generic second-generation instances).  The limit is therefore only ever RAISED: the largest
 * value any engine of the process has asked for, per (device, instance). */
int raise_lds_limit(int device, const void *fn, uint32_t lds)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, uint32_t> limit;
    std::lock_guard<std::mutex> guard(mu);
    uint32_t &have = limit[std::make_pair(device, fn)];
    if (lds > have) {
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        have = lds;
    }
    return MFM_OK;
}

/*
 * The discriminator's division (mfm_numerics.h, mfm_div_unit) is correctly rounded with the reciprocals gfx950's
 * v_rcp_f32 returns - shown for all operands by tools/div_proof.c on the table tools/rcp_check.hip reads off the device.  It
 * would NOT be for every reciprocal within one ulp: 8388608 / 16777215 needs the correctly rounded reciprocal of
 * 2^24 - 1, 13981011 / 16777213 and 15099490 / 16777211 a reciprocal that is not too high.  Once per process and device
 * the engine therefore runs exactly those quotients (all octants, the kernel variant in use) through the device's
 * discriminator and compares with the host twin's IEEE division: a device that answers differently is refused at
 * commit instead of producing PCM that is one LSB off once in 10^7 samples.
 */
/* What v_rcp_f32 returns for every binary32 significand, as one number: the difference (in ulps) between the device's
 * reciprocal of 2^23 + i and the correctly rounded one, hashed with i; the counts of -1 / 0 / +1 / anything else beside it.
 * tools/div_proof.c's enumeration holds for the table this hash stands for (MFM_RCP_TABLE_HASH_GFX950, read off an MI355X
 * by tools/rcp_check.hip: 89 % correctly rounded, 9 % one ulp low, 2 % one ulp high). */
__global__ void mfm_rcp_table_kernel(unsigned long long *acc)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float b = (float)((1u << 23) + i);
    const float r = __builtin_amdgcn_rcpf(b);
    const float rn = __fdiv_rn(1.0f, b);
    const int32_t d = (int32_t)__float_as_uint(r) - (int32_t)__float_as_uint(rn);
    unsigned long long h = (unsigned long long)(uint32_t)(d + 8) * ((unsigned long long)i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull);
    for (int o = 32; o > 0; o >>= 1) {
        h += __shfl_down(h, o);
    }
    const unsigned long long lo = __ballot(d == -1), eq = __ballot(d == 0), hi = __ballot(d == 1);
    if ((threadIdx.x & 63u) == 0u) {
        atomicAdd(&acc[0], h);
        atomicAdd(&acc[1], (unsigned long long)__popcll(lo));
        atomicAdd(&acc[2], (unsigned long long)__popcll(eq));
        atomicAdd(&acc[3], (unsigned long long)__popcll(hi));
        atomicAdd(&acc[4], 64ull - (unsigned long long)(__popcll(lo) + __popcll(eq) + __popcll(hi)));
    }
}

/* the discriminator's division against the IEEE quotient on pseudo-random significand pairs and on the divisors next to a
 * power of two (the classical hard ones): the net under a device whose reciprocal table is not the one of the proof */
__global__ void mfm_div_sweep_kernel(uint32_t seed, uint32_t per_thread, unsigned long long *bad)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = ((uint64_t)seed << 32) ^ ((uint64_t)t * 0xD1B54A32D192ED03ull + 0x9E3779B97F4A7C15ull);
    uint32_t wrong = 0;
    for (uint32_t k = 0; k < per_thread; k++) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        uint32_t A = (1u << 23) | ((uint32_t)x & 0x7fffffu), B = (1u << 23) | ((uint32_t)(x >> 32) & 0x7fffffu);
        if ((k & 3u) == 3u) {
            const uint32_t near = (uint32_t)(x >> 55) & 0xffu; /* divisors within 256 of 2^24 and of 2^23 */
            B = (k & 4u) ? (1u << 24) - 1u - near : (1u << 23) + near;
        }
        const float b = (float)B;
        const float a = A <= B ? (float)A : 0.5f * (float)A;
        wrong += mfm_div_unit(a, b) != __fdiv_rn(a, b) ? 1u : 0u;
    }
    if (wrong) {
        atomicAdd(bad, (unsigned long long)wrong);
    }
}

int rcp_table_read(int device, uint64_t out[5])
{
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc(&d, 5 * sizeof(unsigned long long)));
    int rc = MFM_OK;
    if (hipMemset(d, 0, 5 * sizeof(unsigned long long)) != hipSuccess) {
        rc = fail(MFM_E_DEVICE, "hipMemset failed");
    } else {
        hipLaunchKernelGGL(mfm_rcp_table_kernel, dim3((1u << 23) / 256u), dim3(256), 0, nullptr, d);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, d, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(MFM_E_DEVICE, "the reciprocal-table kernel failed on device %d", device);
        }
    }
    (void)hipFree(d);
    return rc;
}

int div_sweep(int device, uint64_t *bad, uint64_t *tried)
{
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(unsigned long long)));
    int rc = MFM_OK;
    const uint32_t blocks = 4096, threads = 256, per_thread = 256; /* 2^28 quotients, a few ms */
    if (hipMemset(d, 0, sizeof(unsigned long long)) != hipSuccess) {
        rc = fail(MFM_E_DEVICE, "hipMemset failed");
    } else {
        hipLaunchKernelGGL(mfm_div_sweep_kernel, dim3(blocks), dim3(threads), 0, nullptr, 0x5eed5eedu, per_thread, d);
        unsigned long long h = 0;
        if (hipGetLastError() != hipSuccess || hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(MFM_E_DEVICE, "the division sweep failed on device %d", device);
        }
        *bad = h;
        *tried = (uint64_t)blocks * threads * per_thread;
    }
    (void)hipFree(d);
    return rc;
}

int division_selftest(int device, int variant)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, int> done;
    std::lock_guard<std::mutex> guard(mu);
    auto it = done.find(std::make_pair(device, variant));
    if (it != done.end()) {
        return it->second == MFM_OK ? MFM_OK : fail(it->second, "the division self-test failed on device %d before", device);
    }
    static const int32_t pairs[][2] = { { 8388608, 16777215 }, { 13981011, 16777213 }, { 15099490, 16777211 },
                                        { 8388607, 16777215 }, { 12582912, 16777215 }, { 1, 3 }, { 16777215, 16777215 },
                                        { 5, 2147483647 }, { 1073741823, 2147483646 } };
    std::vector<int32_t> re, im;
    for (const auto &p : pairs) {
        for (int o = 0; o < 8; o++) {
            const int32_t x = (o & 1) ? p[0] : p[1], y = (o & 1) ? p[1] : p[0];
            re.push_back((o & 2) ? -x : x);
            im.push_back((o & 4) ? -y : y);
        }
    }
    std::vector<int16_t> got(re.size());
    int rc = mfm_devtest_discriminate(variant, re.data(), im.data(), re.size(), got.data(), device);
    if (rc == MFM_OK) {
        for (size_t i = 0; i < re.size(); i++) {
            const int16_t want = (int16_t)mfm_hosttwin_discriminate(re[i], im[i]);
            if (got[i] != want) {
                rc = fail(MFM_E_DEVICE, "division self-test: discriminator(%d, %d) = %d on device %d, %d by IEEE division - this "
                                        "device's v_rcp_f32 differs from gfx950's (tools/div_proof.c)", re[i], im[i], got[i], device, want);
                break;
            }
        }
    }
    if (rc == MFM_OK) {
        /* once per device: is v_rcp_f32's table the one tools/div_proof.c enumerated?  If it is not (another stepping, other
         * microcode), the three quotients above are no proof: 2^28 quotients against the IEEE division decide, and a
         * single wrong one refuses the device. */
        static std::map<int, int> table_done;
        auto td = table_done.find(device);
        if (td == table_done.end()) {
            uint64_t t[5] = { 0, 0, 0, 0, 0 };
            int trc = rcp_table_read(device, t);
            if (trc == MFM_OK && t[0] != MFM_RCP_TABLE_HASH_GFX950) {
                uint64_t bad = 0, tried = 0;
                trc = div_sweep(device, &bad, &tried);
                if (trc == MFM_OK && bad != 0) {
                    trc = fail(MFM_E_DEVICE, "device %d: v_rcp_f32 is not gfx950's table (hash %016llx, %llu / %llu / %llu one ulp low / exact / "
                                             "high) and %llu of %llu quotients differ from the IEEE division", device,
                               (unsigned long long)t[0], (unsigned long long)t[1], (unsigned long long)t[2], (unsigned long long)t[3],
                               (unsigned long long)bad, (unsigned long long)tried);
                }
            }
            table_done[device] = trc;
            rc = trc;
        } else if (td->second != MFM_OK) {
            rc = fail(td->second, "the reciprocal-table check failed on device %d before", device);
        }
    }
    done[std::make_pair(device, variant)] = rc;
    return rc;
}


/* which instance runs blocks of each input format; its LDS limit is raised here, once (commit, on the engine's device) */
int select_kernels(mfm_engine *e)
{
    for (int fmt = MFM_IN_CS16; fmt <= MFM_IN_RTLSDR_U8; fmt++) {
        e->kfn[fmt] = nullptr;
        if (fmt != MFM_IN_CS16 && !e->raw8_ok) {
            continue;
        }
        const void *fn = nullptr;
        uint32_t lds = 0;
        if (e->use_v3) {
            mfm_launch_v3 V{};
            fill_v3(e, fmt, V);
            HIP_TRY(mfm_select_channel_kernel_v3(&V, e->any_iq ? 1 : 0, &fn));
            lds = e->v_lds_bytes;
        } else if (e->use_mfma) {
            mfm_launch_mfma M{};
            fill_mfma(e, fmt, M);
            uint32_t wps = 4;
            HIP_TRY(mfm_select_channel_kernel_mfma(&M, e->any_iq ? 1 : 0, &fn, &wps));
            lds = e->m_lds_bytes;
            /* a resident long-filter instance takes a SIMD's registers with two waves: one workgroup per CU */
            e->m_wg_fmt[fmt] = wps < 4u ? 1u : e->m_wg_per_cu;
            if (fmt == MFM_IN_CS16) {
                e->m_resident_taps = wps < 4u;
            }
        } else {
            HIP_TRY(mfm_select_channel_kernel(e->opl, e->any_iq ? 1 : 0, &fn));
            lds = e->lds_bytes;
        }
        {
            const int rc = raise_lds_limit(e->cfg.device, fn, lds);
            if (rc != MFM_OK) {
                return rc;
            }
        }
        e->kfn[fmt] = fn;
    }
    return MFM_OK;
}

int fold_timing(mfm_engine *e, bool all)
{
    while (e->t_tail < e->t_head && (all || e->t_head - e->t_tail >= (uint64_t)kTimingPairs)) {
        const int i = (int)(e->t_tail % kTimingPairs);
        HIP_TRY(hipEventSynchronize(e->t1[i]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e->t0[i], e->t1[i]));
        e->kernel_ms += ms;
        if (e->launch_ms.size() < kLaunchRing) {
            e->launch_ms.resize(kLaunchRing);
        }
        e->launch_ms[e->launch_ms_n % kLaunchRing] = ms;
        e->launch_ms_n++;
        e->t_tail++;
    }
    return MFM_OK;
}

void free_device(mfm_engine *e)
{
    if (!e->committed) {
        return;
    }
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    (void)hipFree(e->d_coef);
    (void)hipFree(e->d_tapoff);
    (void)hipFree(e->d_afrag);
    (void)hipFree(e->d_krow);
    for (int i = 0; i < 4; i++) {
        (void)hipFree(e->d_krow8[i]);
        e->d_krow8[i] = nullptr;
    }
    (void)hipFree(e->d_tailtmp);
    e->d_tailtmp = nullptr;
    (void)hipFree(e->d_info);
    (void)hipFree(e->d_rot);
    (void)hipFree(e->d_cyc);
    (void)hipFree(e->d_lut);
    for (int i = 0; i < 2; i++) {
        (void)hipFree(e->d_state[i]);
    }
    for (int i = 0; i < kMaxInBufs; i++) {
        if (e->own_in) {
            (void)hipFree(e->d_in[i]);
        }
        if (e->d_raw[i]) {
            (void)hipFree(e->d_raw[i]);
        }
        if (e->h_in[i]) {
            (void)hipHostFree(e->h_in[i]);
        }
        if (e->in_free[i]) {
            (void)hipEventDestroy(e->in_free[i]);
        }
    }
    for (int i = 0; i < kOutSlots; i++) {
        OutSlot &s = e->slots[i];
        (void)hipFree(s.d_pcm);
        (void)hipFree(s.d_iq);
        if (s.h_pcm) {
            (void)hipHostFree(s.h_pcm);
        }
        if (s.h_iq) {
            (void)hipHostFree(s.h_iq);
        }
        if (s.ready) {
            (void)hipEventDestroy(s.ready);
        }
    }
    if (e->timing_events) {
        for (int i = 0; i < kTimingPairs; i++) {
            if (e->t0[i]) {
                (void)hipEventDestroy(e->t0[i]);
            }
            if (e->t1[i]) {
                (void)hipEventDestroy(e->t1[i]);
            }
        }
    }
    if (e->in_ready) {
        (void)hipEventDestroy(e->in_ready);
    }
    for (uint64_t i = 0; i < mfm_engine::kCopyRing; i++) {
        if (e->copy_ev[i]) {
            (void)hipEventDestroy(e->copy_ev[i]);
            e->copy_ev[i] = nullptr;
        }
        e->copy_ev_seq[i] = 0;
    }
    e->copy_seq = e->copy_done_seq = e->copy_ev_next = 0;
    e->wait_copy_stream = false;
    if (e->kernel_done) {
        (void)hipEventDestroy(e->kernel_done);
    }
    if (e->s_in) {
        (void)hipStreamDestroy(e->s_in);
    }
    if (e->cs[1] && e->cs[1] != e->s_compute) {
        (void)hipStreamDestroy(e->cs[1]);
    }
    e->cs[0] = e->cs[1] = e->s_last = nullptr;
    for (int i = 0; i < kMaxInBufs; i++) {
        if (e->tail_done[i]) {
            (void)hipEventDestroy(e->tail_done[i]);
            e->tail_done[i] = nullptr;
        }
        e->tail_pending[i] = false;
    }
    if (e->s_compute) {
        (void)hipStreamDestroy(e->s_compute);
    }
    if (e->s_out) {
        (void)hipStreamDestroy(e->s_out);
    }
    e->d_coef = e->d_tapoff = e->d_afrag = nullptr;
    e->d_krow = nullptr;
    e->d_info = nullptr;
    e->d_rot = nullptr;
    e->d_cyc = nullptr;
    e->d_lut = nullptr;
    for (int i = 0; i < 2; i++) {
        e->d_state[i] = nullptr;
    }
    for (int i = 0; i < kMaxInBufs; i++) {
        e->d_in[i] = nullptr;
        e->d_raw[i] = nullptr;
        e->h_in[i] = nullptr;
        e->in_free[i] = e->in_free_wait[i] = nullptr;
    }
    for (int i = 0; i < kOutSlots; i++) {
        e->slots[i] = OutSlot();
    }
    e->in_ready = e->kernel_done = nullptr;
    e->s_in = e->s_compute = e->s_out = nullptr;
    e->timing_events = false;
    e->committed = false;
}

uint64_t input_capacity(uint32_t max_block, uint32_t coalesce, uint32_t nr_taps)
{
    /* history tail (< nr_taps samples) + block + one 16-byte staging chunk of slack (a chunk that starts on the last
     * real sample must still be readable in place: decimations that are not multiples of 4 start their chunks at any
     * sample), rounded to 64 samples.  A coalescing engine launches once coalesce_samples have gathered: fewer than that
     * plus one more block of any size always fit. */
    return ((uint64_t)max_block + coalesce + 2ull * nr_taps + 4u + 63u) & ~63ull; /* 2 x: history tail + mfm_engine::hist (<= taps) */
}

int write_state_fresh(mfm_engine *e)
{
    std::vector<mfm_chan_state> st(e->ngroups * MFM_CG);
    for (auto &s : st) {
        s.carry_q = 0; /* fm_demod.c: last sample starts at 0 (TZAALLOC, :29) */
        s.kb = 0;
    }
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipMemcpy(e->d_state[i], st.data(), st.size() * sizeof(mfm_chan_state), hipMemcpyHostToDevice));
    }
    e->parity = 0;
    return MFM_OK;
}

/*
 * 8-bit ingest on the device.  The reference widens 8-bit captures to int16 on the
 * host before anything else sees them: cs8 is a plain sign extension (multifm/file_if.c:91-103); cu8 reads the
 * bytes as SIGNED and subtracts 127 (:122,:139-151; after an odd number of samples the last one misses the
 * subtraction, :146-150); the RTL-SDR front end computes (u8 - 127) << 7 (multifm/rtl_sdr_if.c:146-158).
 * Here the raw byte pairs cross PCIe (half the bytes) and are widened where the channel kernel will read them.
 * HBM-bound: 2 bytes in, 4 bytes out per sample; one lane handles 8 samples (16 B in, 2 x 16 B out).
 */
__device__ __forceinline__ uint32_t mfm_widen_pair(uint32_t two_bytes, int format)
{
    int32_t a, b;
    if (format == MFM_IN_RTLSDR_U8) {
        a = (((int32_t)(two_bytes & 0xffu)) - 127) * 128;
        b = (((int32_t)((two_bytes >> 8) & 0xffu)) - 127) * 128;
    } else {
        a = (int8_t)(two_bytes & 0xffu);
        b = (int8_t)((two_bytes >> 8) & 0xffu);
        if (format == MFM_IN_CU8) {
            a -= 127;
            b -= 127;
        }
    }
    return mfm_pack16(a, b);
}

__global__ __launch_bounds__(256) void mfm_unpack_kernel(const uint16_t *raw, uint32_t *dst, uint32_t nr_samples, int format,
                                                         int cu8_odd_tail)
{
    const uint32_t ngroups = nr_samples / 8;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += gridDim.x * blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4 *>(raw)[g];
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
        uint32_t o[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            o[2 * k] = mfm_widen_pair(w[k] & 0xffffu, format);
            o[2 * k + 1] = mfm_widen_pair(w[k] >> 16, format);
        }
        reinterpret_cast<uint4 *>(dst)[2 * g] = make_uint4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<uint4 *>(dst)[2 * g + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
    /* the last nr_samples % 8, one lane each */
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = ngroups * 8 + t;
    if (s < nr_samples) {
        uint32_t out = mfm_widen_pair(raw[s], format);
        if (cu8_odd_tail && format == MFM_IN_CU8 && (nr_samples & 1u) && s == nr_samples - 1) {
            const uint32_t two = raw[s]; /* file_if.c:146-150: the remainder loop stores the bare cast */
            out = mfm_pack16((int8_t)(two & 0xffu), (int8_t)(two >> 8));
        }
        dst[s] = out;
    }
}

} /* namespace */

/* ------------------------------------------------------------------------------------- */

extern "C" {

const char *mfm_strerror(int err)
{
    switch (err) {
    case MFM_OK: return "ok";
    case MFM_E_INVAL: return "invalid argument";
    case MFM_E_NOMEM: return "out of memory";
    case MFM_E_BUSY: return "busy";
    case MFM_E_DEVICE: return "HIP device error";
    case MFM_E_STATE: return "invalid state";
    case MFM_E_DONE: return "done";
    default: return "unknown error";
    }
}

const char *mfm_last_error(void)
{
    return g_last_error;
}

/* for the other translation units of the library (mfm_resampler.hip, mfm_pocsag.hip) */
__attribute__((visibility("hidden"))) void mfm_internal_set_error(const char *msg)
{
    snprintf(g_last_error, sizeof(g_last_error), "%s", msg);
}

size_t mfm_engine_input_bytes(uint32_t max_block_samples, uint32_t nr_taps)
{
    return (size_t)input_capacity(max_block_samples, 0, nr_taps) * sizeof(uint32_t);
}

size_t mfm_engine_input_bytes_cfg(const struct mfm_engine_config *cfg, uint32_t nr_taps, uint32_t *nr_buffers)
{
    if (!cfg) {
        return 0;
    }
    if (nr_buffers) {
        *nr_buffers = cfg->coalesce_samples ? 3u : 2u;
    }
    return (size_t)input_capacity(cfg->max_block_samples, cfg->coalesce_samples, nr_taps) * sizeof(uint32_t);
}

int mfm_engine_create(struct mfm_engine **pe, const struct mfm_engine_config *cfg)
{
    if (!pe || !cfg) {
        return fail(MFM_E_INVAL, "NULL argument");
    }
    *pe = nullptr;
    if (cfg->abi_version != MFM_ABI_VERSION) {
        return fail(MFM_E_INVAL, "ABI version %u, library is %u", cfg->abi_version, MFM_ABI_VERSION);
    }
    if (0 == cfg->decimation || 0 == cfg->sample_rate_hz || 0 == cfg->max_block_samples) {
        return fail(MFM_E_INVAL, "decimation, sample rate and max block must be non-zero");
    }
    if (cfg->max_block_samples > (1u << 30)) {
        return fail(MFM_E_INVAL, "max_block_samples %u too large", cfg->max_block_samples);
    }
    if ((cfg->ext_input[0] == nullptr) != (cfg->ext_input[1] == nullptr) ||
        (cfg->coalesce_samples && (cfg->ext_input[0] == nullptr) != (cfg->ext_input[2] == nullptr))) {
        return fail(MFM_E_INVAL, "ext_input needs every buffer (two; three with coalesce_samples) or none");
    }
    if ((uint64_t)cfg->max_block_samples + cfg->coalesce_samples > (1u << 30)) {
        return fail(MFM_E_INVAL, "max_block_samples + coalesce_samples = %llu too large",
                    (unsigned long long)cfg->max_block_samples + cfg->coalesce_samples);
    }
    mfm_engine *e = new (std::nothrow) mfm_engine();
    if (!e) {
        return fail(MFM_E_NOMEM, "engine allocation failed");
    }
    e->cfg = *cfg;
    *pe = e;
    return MFM_OK;
}

void mfm_engine_destroy(struct mfm_engine **pe)
{
    if (!pe || !*pe) {
        return;
    }
    free_device(*pe);
    delete *pe;
    *pe = nullptr;
}

int mfm_engine_add_channel_q14(struct mfm_engine *e, const int16_t *coeff_re, const int16_t *coeff_im,
                               size_t nr_taps, int16_t rot_incr_re, int16_t rot_incr_im, int want_iq)
{
    if (!e || !coeff_re || !coeff_im || 0 == nr_taps) {
        return fail(MFM_E_INVAL, "NULL or empty taps");
    }
    if (e->committed) {
        return fail(MFM_E_STATE, "channel set is frozen after commit");
    }
    if (!e->chans.empty() && nr_taps != e->nr_taps) {
        return fail(MFM_E_INVAL, "all channels share one tap count (%u), got %zu", e->nr_taps, nr_taps);
    }
    if (nr_taps < e->cfg.decimation) {
        /* the reference dereferences a NULL sb_active in this configuration (direct_fir.c:394-398) */
        return fail(MFM_E_INVAL, "taps (%zu) < decimation (%u) is not a valid multifm configuration", nr_taps,
                    e->cfg.decimation);
    }
    if (nr_taps > 65535) {
        return fail(MFM_E_INVAL, "too many taps");
    }
    for (size_t i = 0; i < nr_taps; i++) {
        if (coeff_im[i] == INT16_MIN) {
            /* -ci is not an int16: the packed (cr,-ci) operand cannot hold it */
            return fail(MFM_E_INVAL, "imaginary tap %zu is -32768 (a tap of magnitude 2.0); not supported", i);
        }
    }
    Channel c;
    c.cre.assign(coeff_re, coeff_re + nr_taps);
    c.cim.assign(coeff_im, coeff_im + nr_taps);
    c.incr_re = rot_incr_re;
    c.incr_im = rot_incr_im;
    c.want_iq = want_iq != 0;
    e->nr_taps = (uint32_t)nr_taps;
    e->chans.push_back(std::move(c));
    return (int)e->chans.size() - 1;
}

int mfm_engine_add_channel(struct mfm_engine *e, int32_t offset_hz, const double *lpf_taps, size_t nr_taps,
                           double channel_gain, int want_iq)
{
    if (!e || !lpf_taps || 0 == nr_taps) {
        return fail(MFM_E_INVAL, "NULL or empty taps");
    }
    std::vector<int16_t> cre(nr_taps), cim(nr_taps);
    int16_t ir = 0, ii = 0;
    mfm_taps_rotate_q14(lpf_taps, nr_taps, offset_hz, e->cfg.sample_rate_hz, channel_gain, cre.data(), cim.data());
    mfm_taps_rot_increment(offset_hz, e->cfg.sample_rate_hz, e->cfg.decimation, &ir, &ii);
    return mfm_engine_add_channel_q14(e, cre.data(), cim.data(), nr_taps, ir, ii, want_iq);
}

int mfm_engine_get_channel(struct mfm_engine *e, uint32_t chan, int16_t *coeff_re, int16_t *coeff_im,
                           int16_t rot_incr[2])
{
    if (!e || chan >= e->chans.size()) {
        return fail(MFM_E_INVAL, "no such channel");
    }
    const Channel &c = e->chans[chan];
    if (coeff_re) {
        memcpy(coeff_re, c.cre.data(), c.cre.size() * sizeof(int16_t));
    }
    if (coeff_im) {
        memcpy(coeff_im, c.cim.data(), c.cim.size() * sizeof(int16_t));
    }
    if (rot_incr) {
        rot_incr[0] = c.incr_re;
        rot_incr[1] = c.incr_im;
    }
    return MFM_OK;
}

static int commit_locked(struct mfm_engine *e);

int mfm_engine_commit(struct mfm_engine *e)
{
    if (!e) {
        return fail(MFM_E_INVAL, "NULL engine");
    }
    if (e->committed) {
        return fail(MFM_E_STATE, "already committed");
    }
    for (int i = 0; i < kMaxInBufs; i++) {
        if (e->cfg.ext_input[i] && (reinterpret_cast<uintptr_t>(e->cfg.ext_input[i]) & 15u)) {
            return fail(MFM_E_INVAL, "ext_input buffers must be 16-byte aligned");
        }
    }
    const int rc = commit_locked(e);
    if (rc != MFM_OK && e->committed) {
        /* a failure half-way through the device allocations: release what exists and leave the engine
         * uncommitted (the data-path entry points check `committed`), keeping the failure's message */
        char keep[sizeof(g_last_error)];
        memcpy(keep, g_last_error, sizeof(keep));
        free_device(e);
        memcpy(g_last_error, keep, sizeof(keep));
    }
    return rc;
}

static int commit_locked(struct mfm_engine *e)
{
    if (e->chans.empty()) {
        return fail(MFM_E_INVAL, "no channels");
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        return fail(MFM_E_DEVICE, "no HIP device available (this library has no CPU path)");
    }
    if (e->cfg.device < 0 || e->cfg.device >= ndev) {
        return fail(MFM_E_DEVICE, "device %d out of range (%d devices)", e->cfg.device, ndev);
    }
    HIP_TRY(hipSetDevice(e->cfg.device));

    const uint32_t T = e->nr_taps, D = e->cfg.decimation, C = (uint32_t)e->chans.size();
    const bool dev_only = (e->cfg.flags & MFM_F_DEVICE_ONLY) != 0;

    /* ---- geometry: outputs per lane, LDS tile ---- */
    const uint32_t qrows = (T + D - 1) / D;
    auto tile_dwords = [&](int opl, uint32_t *rs2) {
        const uint32_t rows = 64u * opl - 1u + qrows;
        *rs2 = rows | 1u; /* odd stride: transposing stores hit 32 different banks */
        return (uint64_t)D * *rs2;
    };
    uint32_t rs2 = 0;
    int opl = 2;
    uint64_t dw = tile_dwords(2, &rs2);
    if ((dw + 512) * 4 > 53 * 1024) {
        opl = 1;
        dw = tile_dwords(1, &rs2);
    }
    if ((dw + 512) * 4 > 160 * 1024) {
        return fail(MFM_E_INVAL, "decimation %u needs a %llu-byte LDS tile (> 160 KiB)", D,
                    (unsigned long long)(dw + 512) * 4);
    }
    e->opl = opl;
    e->rs2 = rs2;
    e->lut_off = (uint32_t)((dw + 3) & ~3ull);
    e->lds_bytes = (e->lut_off + 512) * 4;
    e->nchunks = (T + MFM_TG - 1) / MFM_TG;
    e->ngroups = (C + MFM_CG - 1) / MFM_CG;
    e->gpw = std::min<uint32_t>(MFM_NW, e->ngroups);
    e->nslices = (e->ngroups + e->gpw - 1) / e->gpw;
    e->cap_in = (uint32_t)input_capacity(e->cfg.max_block_samples, e->cfg.coalesce_samples, T);
    e->nbuf = e->cfg.coalesce_samples ? 3 : 2;
    /* a multiple of 8 outputs: channel rows of the PCM buffer start 16-byte aligned (8-byte PCM / 16-byte IQ stores) */
    e->out_stride = ((e->cap_in - T) / D + 1 + 7) & ~7u;
    e->any_iq = false;
    for (const Channel &c : e->chans) {
        e->any_iq |= c.want_iq;
    }
    /* the kernels address outputs with 32-bit byte offsets from the buffer base: 2 bytes per PCM sample, 4 per
     * filtered-IQ sample, plus the dump slots behind the last row */
    const uint64_t out_limit = e->any_iq ? (1ull << 30) : (1ull << 31);
    if ((uint64_t)C * e->out_stride + 64 >= out_limit) {
        return fail(MFM_E_INVAL, "channels x outputs per block = %llu exceeds %llu (use smaller blocks)",
                    (unsigned long long)C * e->out_stride, (unsigned long long)out_limit);
    }

    /* ---- tap table: [group][chunk][tap in chunk][channel in group]{(cr,-ci),(ci,cr)} ---- */
    const size_t coef_dwords = (size_t)e->ngroups * e->nchunks * MFM_TG * MFM_CG * 2;
    std::vector<uint32_t> coef(coef_dwords, 0u);
    for (uint32_t c = 0; c < C; c++) {
        const Channel &ch = e->chans[c];
        const uint32_t g = c / MFM_CG, cc = c % MFM_CG;
        for (uint32_t i = 0; i < T; i++) {
            const uint32_t chunk = i / MFM_TG, k = i % MFM_TG;
            const size_t at = ((((size_t)g * e->nchunks + chunk) * MFM_TG + k) * MFM_CG + cc) * 2;
            const int32_t cr = ch.cre[i], ci = ch.cim[i];
            coef[at + 0] = mfm_pack16(cr, -ci);
            coef[at + 1] = mfm_pack16(ci, cr);
        }
    }
    std::vector<uint32_t> tapoff((size_t)e->nchunks * MFM_TG, 0u);
    for (uint32_t i = 0; i < T; i++) {
        tapoff[i] = ((i % D) * rs2 + i / D) * 4u;
    }

    /* ---- matrix-core path.  An LDS row (the D samples between two outputs) is 2*D plane bytes; the B fragments are
     *      16-byte reads, so rows are padded to 16-byte multiples when D is not a multiple of 8, and the taps (the A
     *      operand) carry zeros over the padding: decimation 25 of etc/pocsag_rtlsdr.json = rows of 50 + 14 bytes,
     *      128 taps = 5 full rows + 3 taps = 326 elements -> 6 k-steps instead of 4.  Usable while at least 3/4 of a
     *      row is samples, the padded filter fits the 16 k-steps of the streaming variant and every tap splits into
     *      two signed bytes. ---- */
    std::vector<uint32_t> afrag;
    std::vector<int32_t> krow;
    /* decimations 1, 2, 4 (etc/multifm_file.json: 1): rows shorter than a fragment - the long-filter kernel keeps 8 / D shifted
     * copies of the image instead of padding them (mfm_kernel_v3l.hip, SHIFT); the window is then the unpadded 2 T elements */
    const bool shift_geo = (1u == D || 2u == D || 4u == D) && T <= 512u && !(e->cfg.flags & (MFM_F_FORCE_MFMA_V1 | MFM_F_FORCE_DOT2));
    const uint32_t row_bytes_p = shift_geo ? 2u * D : (2u * D + 15u) & ~15u;
    const uint32_t k_elems = ((T - 1u) / D) * row_bytes_p + 2u * ((T - 1u) % D) + 2u; /* element index of the last tap + 1 */
    e->use_mfma = 8u * D >= 3u * row_bytes_p && k_elems <= 64u * MFM_MFMA_KQ_STREAM_MAX &&
                  !(e->cfg.flags & MFM_F_FORCE_DOT2);
    for (const Channel &ch : e->chans) {
        for (uint32_t i = 0; i < T && e->use_mfma; i++) {
            if (ch.cre[i] > 32639 || ch.cim[i] > 32639 || ch.cim[i] < -32639) {
                e->use_mfma = false; /* 256*Wh + Wl with both in int8 needs W <= 32639 */
            }
        }
    }
    if (e->use_mfma) {
        uint32_t kq = 1;
        while (64u * kq < k_elems) {
            kq *= 2;
        }
        e->m_kq_used = (k_elems + 63u) / 64u;
        const uint32_t row_bytes = row_bytes_p;
        const bool padded = row_bytes_p != 2u * D;
        /* LDS row stride: an ODD multiple of 32 bytes.  tools/ubench_lds.hip: the B-fragment read pattern (lane
         * 16 kg + n reads 16 bytes at n * rs + 16 kg) runs at the full ds_read_b128 rate for rs = 224 and at 76-81 %
         * of it for 80, 144, 176, 192, 208, 272 - odd multiples of 16 bytes are not enough. */
        uint32_t rs_m = (row_bytes + 31u) / 32u * 32u;
        if (((rs_m / 32u) & 1u) == 0u) {
            rs_m += 32u;
        }
        uint32_t ot = 0, plane = 0, lds = 0;
        const uint32_t want[] = { 2u * 31u, 31u }; /* new outputs per tile: two 31-output iterations, or one for large decimations */
        for (uint32_t cand : want) {
            /* samples a tile stages: its outputs' rows plus the rows the last window reaches into */
            const uint32_t nst = padded ? ((cand * D + ((64u * kq + row_bytes - 1u) / row_bytes) * D) + 3u) & ~3u
                                        : ((cand * D + 32u * kq) + 3u) & ~3u;
            const uint32_t rows = (nst + D - 1u) / D;
            const uint32_t pb = rows * rs_m;
            /* two staging buffers x two byte planes + atan LUT + staging offsets + rotator constants of up to 256 channels */
            const uint32_t nch_t = (nst / 4u + MFM_MFMA_NW * 64u - 1u) / (MFM_MFMA_NW * 64u); /* staging chunks per thread */
            const uint32_t need = 4u * pb + 2048u + nch_t * MFM_MFMA_NW * 64u * 4u + (kq > MFM_MFMA_KQ_MAX ? 2048u : 0u) +
                                  (C <= 256u ? 32u * C : 0u);
            /* up to 80 KB two workgroups share a CU; beyond that one per CU is still far better than the v_dot2 kernel */
            /* the kernels are built for up to 4 chunks per thread with two iterations, up to MFM_M_CH_MAX with one
             * (and then only for 128-tap-class and longer filters) */
            const uint32_t nch_max = cand == 62u ? 4u : MFM_M_CH_MAX;
            if (cand == 31u && kq < MFM_MFMA_KQ_MAX) {
                continue;
            }
            if (need <= 150u * 1024u && nch_t <= nch_max) {
                ot = cand;
                plane = pb;
                lds = need;
                break;
            }
        }
        if (0 == ot) {
            e->use_mfma = false;
        } else {
            e->m_ks = kq;
            e->m_row_bytes = row_bytes;
            e->m_nstage = padded ? ((ot * D + ((64u * kq + row_bytes - 1u) / row_bytes) * D) + 3u) & ~3u
                                 : ((ot * D + 32u * kq) + 3u) & ~3u;
            e->m_ot = ot;
            e->m_rs = rs_m;
            e->m_plane_bytes = plane;
            e->m_lut_off = 4u * plane;
            e->m_lds_bytes = lds;
            /* planes at a fixed 16 KiB pitch when they fit and two workgroups still share a CU: the kernel then
             * reaches the low-byte plane and the second staging buffer through instruction immediates */
            e->m_fixed_planes = false;
            if (plane <= MFM_M_PLANE_DIST && kq <= MFM_MFMA_KQ_MAX && ot == 62u) { /* streaming / single-iteration variants: packed planes */
                const uint32_t lds_fixed = 4u * MFM_M_PLANE_DIST + (lds - 4u * plane);
                if (2u * lds_fixed <= 160u * 1024u) {
                    e->m_fixed_planes = true;
                    e->m_lut_off = 4u * MFM_M_PLANE_DIST;
                    e->m_lds_bytes = lds_fixed;
                    lds = lds_fixed;
                }
            }
            e->m_nrb = (2u * C + 15u) / 16u;
            e->m_nslices = (e->m_nrb + MFM_MFMA_NW - 1u) / MFM_MFMA_NW;
            e->m_wg_per_cu = std::max(1u, std::min(2u, (160u * 1024u) / lds));

        }
    }

    /* ---- second-generation matrix kernel: 64-output tiles, four sub-planes per byte plane (mfm_kernel.h) ---- */
    e->use_v3 = false;
    e->v_layout = 0;
    e->v_shift = 0;
    e->v_rb = 1;
    if (e->use_mfma && !(e->cfg.flags & MFM_F_FORCE_MFMA_V1) && e->m_ks <= 4u && D % 8u == 0u && (2u * D) % 64u != 0u) {
        /* decimations that are multiples of 8 but not of 32 (40: etc/multifm.json, etc/multifm_1ch.json): the chunk-row
         * layout (mfm_kernel.h).  t_per = chunks of four outputs; slots: one per four staged rows, plus the window's reach */
        const uint32_t kq = e->m_ks, row_bytes = 2u * D, cpo = D / 8u, per = 4u * cpo;
        const uint32_t extra = (64u * kq - 1u) / row_bytes;
        const uint32_t rows = MFM_V3_LEAD + MFM_V3_OT + extra;
        const uint32_t nchunks16 = (rows * row_bytes + 15u) / 16u;
        const uint32_t x_max = cpo * 3u + 4u * (kq - 1u);
        const uint32_t pitch = std::max((nchunks16 + per - 1u) / per + 1u, 16u + 1u + x_max / per + 1u);
        const uint32_t plane = ((per + 3u) * pitch * 16u + 63u) & ~63u;
        const uint32_t nstage4 = rows * D / 4u; /* D % 8 == 0 */
        const uint32_t nch = (nstage4 + 511u) / 512u;
        const uint32_t lds = 4u * plane + 2048u + nch * 512u * 4u + 1024u + 2048u + nch * 512u * 4u;
        if (nch <= MFM_V3_CH_MAX && lds <= 160u * 1024u) {
            e->use_v3 = true;
            e->v_layout = 1;
            e->v_t_per = per;
            e->v_t_pitch = pitch;
            e->v_rs = 0;
            e->v_sp_pitch = plane / 4u; /* plane pitch = 4 * sp_pitch, buffer pitch = 8 * sp_pitch, as in the sub-plane layout */
            e->v_nstage4 = nstage4;
            e->v_lds_bytes = lds;
            e->v_wg_per_cu = std::max(1u, std::min(2u, (160u * 1024u) / lds));
            for (uint32_t k = 0; k < 4; k++) {
                e->v_cross[k] = e->v_within[k] = 0;
            }
        }
    }
    if (e->use_mfma && !(e->cfg.flags & MFM_F_FORCE_MFMA_V1) && e->m_ks <= 4u && (2u * D) % 64u == 0u) {
        const uint32_t kq = e->m_ks, row_bytes = 2u * D;
        uint32_t rs_v = (row_bytes + 31u) / 32u * 32u;
        if (((rs_v / 32u) & 1u) == 0u) {
            rs_v += 32u; /* odd multiple of 32 bytes: the B-fragment read pattern runs at the full ds_read_b128 rate */
        }
        const uint32_t extra = (64u * kq - 1u) / row_bytes; /* rows the last output's window reaches past the tile */
        const uint32_t rows = MFM_V3_LEAD + MFM_V3_OT + extra;
        const uint32_t sr = (rows + 3u) / 4u;
        uint32_t sp = sr * rs_v;
        sp = sp <= mfm_sp_pitch_v3() ? mfm_sp_pitch_v3() : (sp + 63u) & ~63u;
        const uint32_t nstage4 = rows * D / 4u; /* D % 16 == 0 here */
        const uint32_t nch = (nstage4 + 511u) / 512u;
        const uint32_t lds = 16u * sp + 2048u + nch * 512u * 4u + 1024u + 2048u; /* image, atan table, staging offsets, row constants + fold constants, exact-rotator table */
        if (nch <= MFM_V3_CH_MAX && lds <= 160u * 1024u) {
            e->use_v3 = true;
            e->v_rs = rs_v;
            e->v_sp_pitch = sp;
            e->v_nstage4 = nstage4;
            e->v_lds_bytes = lds;
            e->v_wg_per_cu = std::max(1u, std::min(2u, (160u * 1024u) / lds));
#ifdef MFM_EXP_ONE_WG_PER_CU /* occupancy experiment (tools/exp/snapeng.sh): LDS padded so that one workgroup fits a CU */
            e->v_lds_bytes = 100u * 1024u;
            e->v_wg_per_cu = 1u;
#endif
            for (uint32_t k = 0; k < 4; k++) {
                e->v_cross[k] = k < kq ? (64u * k) / row_bytes : 0u;
                e->v_within[k] = k < kq ? (64u * k) % row_bytes : 0u;
            }
        }
    }

    if (e->use_mfma && !(e->cfg.flags & MFM_F_FORCE_MFMA_V1) && 25u == D && T <= 150u) {
        /* decimation 25 (etc/pocsag_rtlsdr.json) on the second generation: rows of 50 plane bytes padded to 64, so that a k-step
         * is exactly one row and a window of up to 150 taps spans six of them (layout 2, mfm_kernel.h); sub-planes at the fixed
         * pitch, the image staged sample by sample.  The tap fragments are laid out for six k-steps. */
        e->use_v3 = true;
        e->v_layout = 2;
        e->v_rs = 96u;
        e->v_sp_pitch = mfm_sp_pitch_v3();
        e->v_nstage4 = (73u * 25u + 3u + 3u) / 4u; /* 16-byte chunks covering the image wherever it starts inside the first one */
        e->v_lds_bytes = 16u * mfm_sp_pitch_v3() + 2048u + 4u * 512u * 4u + 1024u + 2048u;
        e->v_wg_per_cu = 2u;
        for (uint32_t k = 0; k < 4; k++) {
            e->v_cross[k] = k;
            e->v_within[k] = 0;
        }
        e->m_ks = 6u;
        e->m_kq_used = 6u;
    }

    if (shift_geo && e->use_mfma) {
        /* ---- decimations 1, 2, 4 on the long-filter kernel's shifted copies (mfm_kernel_v3l.hip, SHIFT): whole-tile images of
         *      8 / D copies, one row block per wave; the first generation's geometry above does not apply to rows this short ---- */
        const uint32_t kq_inst = e->m_kq_used <= 4u ? 4u : e->m_kq_used <= 8u ? 8u : 16u;
        uint32_t hi_mask = 0;
        for (const Channel &ch : e->chans) {
            for (uint32_t i = 0; i < T; i++) {
                for (int32_t w : { (int32_t)ch.cre[i], (int32_t)ch.cim[i], -(int32_t)ch.cim[i] }) {
                    const int32_t wl = (int8_t)(w & 0xff);
                    if (((w - wl) >> 8) != 0) {
                        hi_mask |= 1u << ((2u * i) / 64u);
                    }
                }
            }
        }
        const uint32_t nc = 8u / D;
        const uint32_t img_samples = 63u * D + T;                       /* what the tile's 64 windows cover */
        const uint32_t read_bytes = 2u * D * 63u + 64u * kq_inst + 16u; /* ... and what the instance's fragment reads touch */
        /* bytes between two copies: at least the copy, and 2 D (mod 16) sixteen-byte units - the sixteen columns of a fragment read
         * (copy n % nc, offset 16 (n / nc)) then fall into sixteen different groups of four LDS banks */
        uint32_t cp16 = (std::max(2u * img_samples, read_bytes) + 15u) / 16u;
        while (cp16 % 16u != (2u * D) % 16u) {
            cp16++;
        }
        const uint32_t plane = nc * cp16 * 16u;
        const uint32_t aux = 8u * 8u * MFM_V3L_TP * 4u + 512u + 2048u;
        const uint32_t lds = 4u * plane + 2048u + 2048u + aux;
        mfm_launch_v3 probe{};
        probe.layout = 3u;
        probe.shift = 1u;
        probe.kq = kq_inst;
        probe.kq_used = e->m_kq_used;
        probe.nh = (uint32_t)__builtin_popcount(hi_mask);
        probe.ng = 4u;
        probe.rb = 1u;
        probe.nstage4 = img_samples;
        const void *fn = nullptr;
        if (img_samples <= 4u * 512u && lds <= 160u * 1024u && mfm_select_channel_kernel_v3(&probe, 0, &fn) == hipSuccess) {
            e->use_v3 = true;
            e->v_layout = 3u;
            e->v_shift = 1u;
            e->v_copy_pitch = cp16 * 16u;
            e->v_rs = 2u * D;
            e->v_plane = plane;
            e->v_sp_pitch = 0;
            e->v_ng = 4u;
            e->v_rb = 1u;
            e->v_nstage4 = img_samples;
            e->v_nstage_p = T;
            e->v_sta_bytes = 2048u;
            e->v_lds_bytes = lds;
            e->v_wg_per_cu = 1u; /* (per input format at launch time: mfm_v3l_wg_per_cu) */
            e->v_kq = kq_inst;
            e->v_nh = probe.nh;
            uint8_t order[16] = { 0 };
            uint32_t at = 0;
            for (uint32_t k = 0; k < e->m_kq_used; k++) {
                if ((hi_mask >> k) & 1u) {
                    order[at++] = (uint8_t)k;
                }
            }
            for (uint32_t k = 0; k < kq_inst; k++) {
                if (!((hi_mask >> k) & 1u)) {
                    order[at++] = (uint8_t)k;
                }
            }
            for (int w = 0; w < 4; w++) {
                e->v_kperm[w] = (uint32_t)order[4 * w] | ((uint32_t)order[4 * w + 1] << 8) | ((uint32_t)order[4 * w + 2] << 16) |
                                ((uint32_t)order[4 * w + 3] << 24);
            }
        } else {
            e->use_mfma = false; /* the v_dot2 kernel */
        }
    }
    /* ---- filters of 129..512 taps on the second-generation structure (layout 3, mfm_kernel_v3l.hip): the first
     *      generation's image - plain rows of m_row_bytes plane bytes at stride m_rs, any decimation - holding a whole
     *      64-output tile, or half of one when two whole ones do not fit LDS (decimation 400 of configs[4]: 112 KB per
     *      image of both planes); same tap fragments, same row constants.
     *      And (round 6) 128-tap filters on SLICES OF 128 CHANNELS where there are that many: two row blocks per wave share
     *      every B fragment and every staged image - half the LDS traffic and half the staging work per (channel, output) of
     *      the 64-channel layouts above, for two waves per SIMD instead of four.  multifm/receiver.c:195-244 builds as many
     *      channels as the configuration lists; north_star's shape is 1024 of them on one GPU. ---- */
    struct l3_cand { uint32_t rb, ng; };
    auto plan_layout3 = [&](const l3_cand *cand, size_t nr_cand) -> bool {
        const uint32_t row_bytes = e->m_row_bytes, rs_l = e->m_rs;
        const uint32_t kq_inst = mfm_v3l_built_kq(e->m_kq_used);
        /* k-steps whose high-byte tap plane is not all zero (what m_ah_mask will say once the fragments are built) */
        uint32_t hi_mask = 0;
        for (const Channel &ch : e->chans) {
            for (uint32_t i = 0; i < T; i++) {
                const uint32_t kst = ((i / D) * row_bytes + 2u * (i % D)) / 64u;
                for (int32_t w : { (int32_t)ch.cre[i], (int32_t)ch.cim[i], -(int32_t)ch.cim[i] }) {
                    const int32_t wl = (int8_t)(w & 0xff);
                    if (((w - wl) >> 8) != 0) {
                        hi_mask |= 1u << kst;
                    }
                }
            }
        }
        const uint32_t reach = (k_elems - 1u) / row_bytes;                       /* rows the last window reaches past its own */
        const uint32_t reach_read = (64u * kq_inst - 1u) / row_bytes + 1u;       /* ... and what the instance's fragment reads touch */
        /* (row blocks per wave, column groups per image), best first: slices of 128 channels - half the staging work and
         * half the B-fragment traffic per (channel, output) - where there are more than 64 channels and two row blocks' taps
         * fit 128 registers, on quarter-tile images; else slices of 64 on whole- or half-tile images */
        const uint32_t nh_inst = mfm_v3l_built_nh(kq_inst, (uint32_t)__builtin_popcount(hi_mask));
        for (size_t ci = 0; ci < nr_cand; ci++) {
            const uint32_t ng = cand[ci].ng, rbw = cand[ci].rb;
            if (2u == rbw && (e->m_nrb <= 8u || 8u * (kq_inst + nh_inst) > 128u || (e->cfg.flags & MFM_F_V3L_ONE_ROW_BLOCK))) {
                continue;
            }
            const uint32_t opi = 16u * ng;
            const uint32_t plane_used = (opi + std::max(reach, reach_read)) * rs_l;
            const uint32_t plane = mfm_v3l_plane_pitch(rbw); /* a constant: the low plane's reads are "high plane + immediate" */
            const uint32_t nstage4 = ((opi + reach) * D + 3u) / 4u;
            const uint32_t nch = (nstage4 + 511u) / 512u;
            if (nch > MFM_V3_CH_MAX || plane_used > plane || kq_inst > e->m_ks) {
                continue;
            }
            /* the instance's count of chunks (a surplus chunk is loaded and not stored); 16-bit offsets where no chunk straddles rows */
            const uint32_t sta = mfm_v3l_built_nch(nch) * ((D % 4u) != 0u ? 1536u : 1024u); /* 16-bit offsets (+ a byte per chunk: rows split) */
            const uint32_t aux = 8u * 8u * rbw * MFM_V3L_TP * 4u + 512u * rbw + 2048u * rbw;
            const uint32_t lds = 4u * plane + 2048u + sta + aux;
            if (lds > 160u * 1024u) {
                continue;
            }
            /* is the int16 instance for this geometry built?  (All are but a few that would need more than 256 registers.)  It
             * must be known HERE: the second generation orders its rows by rotator class, the first does not. */
            mfm_launch_v3 probe{};
            probe.layout = 3u;
            probe.kq = kq_inst;
            probe.kq_used = e->m_kq_used;
            probe.nh = (uint32_t)__builtin_popcount(hi_mask);
            probe.ng = ng;
            probe.rb = rbw;
            probe.nstage4 = nstage4;
            probe.split_rows = (D % 4u) != 0u ? 1u : 0u; /* (sample-by-sample staging: instances of their own, one row block per wave) */
            probe.ah_mask = hi_mask;
            const void *fn = nullptr;
            if (mfm_select_channel_kernel_v3(&probe, 0, &fn) != hipSuccess) {
                continue;
            }
            e->v_rb = rbw;
            e->use_v3 = true;
            e->v_layout = 3u;
            e->v_rs = rs_l;
            e->v_plane = plane;
            e->v_sp_pitch = 0;
            e->v_ng = ng;
            e->v_nstage4 = nstage4;
            e->v_nstage_p = ((1u + reach) * D + 3u) / 4u;
            e->v_sta_bytes = sta;
            e->v_lds_bytes = lds;
            e->v_wg_per_cu = 1u; /* two waves per SIMD hold a long filter's taps: one workgroup per CU */
            e->v_kq = kq_inst;
            e->v_nh = probe.nh;
            {
                /* the order the k-steps are multiplied in: those with a high-byte tap plane first, then the others that
                 * hold taps, then - up to the instance's count - steps of zero taps */
                uint8_t order[16] = { 0 };
                uint32_t at = 0;
                for (uint32_t k = 0; k < e->m_kq_used; k++) {
                    if ((hi_mask >> k) & 1u) {
                        order[at++] = (uint8_t)k;
                    }
                }
                for (uint32_t k = 0; k < kq_inst; k++) {
                    if (!((hi_mask >> k) & 1u)) {
                        order[at++] = (uint8_t)k;
                    }
                }
                for (int w = 0; w < 4; w++) {
                    e->v_kperm[w] = (uint32_t)order[4 * w] | ((uint32_t)order[4 * w + 1] << 8) | ((uint32_t)order[4 * w + 2] << 16) |
                                    ((uint32_t)order[4 * w + 3] << 24);
                }
            }
            return true;
        }
        return false;
    };
    if (e->use_mfma && !e->use_v3 && !(e->cfg.flags & (MFM_F_FORCE_MFMA_V1 | MFM_F_STREAM_TAPS)) && e->m_ks >= 8u &&
        e->m_ks <= MFM_V3L_KQ_MAX) {
        /* (row blocks per wave, column groups per image), best first: slices of 128 channels on quarter-tile images where there
         * are more than 64 channels and two row blocks' taps fit 128 registers; else slices of 64 on whole- or half-tile images */
        const l3_cand cand[3] = { { 2u, 1u }, { 1u, 4u }, { 1u, 2u } };
        plan_layout3(cand, 3);
    }
    e->slice128 = false;
    if (e->use_mfma && e->use_v3 && 0u == e->v_layout && 4u == e->m_ks && (2u * D) % 64u == 0u &&
        !(e->cfg.flags & (MFM_F_FORCE_MFMA_V1 | MFM_F_SLICE_64)) &&
        ((e->cfg.flags & MFM_F_SLICE_128) || C >= kSlice128MinChannels)) {
        /* the sub-plane layout above stays what runs when the instance is not built (more than two high-byte tap planes) */
        const l3_cand cand[1] = { { 2u, 4u } };
        e->slice128 = plan_layout3(cand, 1);
    }

    e->v_nslices = (e->use_v3 && 3u == e->v_layout) ? (e->m_nrb + 8u * e->v_rb - 1u) / (8u * e->v_rb) : e->m_nslices;

    /* ---- rotator classes and row order (filter/direct_fir.c:151-172,406-413).  An increment of exactly (16384, 0) - every
     *      channel whose offset is a multiple of the output rate, e.g. a 25 kHz grid at 2.4 MS/s / 96 - leaves the rotator at
     *      (16384, 0) for ever, and r14(f * 16384) = f: nothing to do.  An increment of (-16384, 0) - offsets at odd
     *      multiples of half the output rate - makes it alternate between (16384, 0) and (-16384, 0), and r14(f * -16384)
     *      = -f with the int16 cast's wrap: one packed multiply by +-1.  An increment of (0, +-16384) - offsets at odd
     *      multiples of a quarter of the output rate - walks the four axis points: r14(f * rot) = f * j^m, a swap of
     *      the halves and two signs.  Everything else is the general case (two Q14
     *      dot products against the tabulated rotator and a second rounding).  The second-generation kernel runs a
     *      64-channel slice without the table loads and the derotation arithmetic when all its channels are exact, so
     *      the rows are ordered by class (stable); where a row's PCM goes is in its mfm_chan_info. ---- */
    auto chan_class = [](const Channel &ch) -> uint32_t {
        if ((ch.incr_re == 16384 && ch.incr_im == 0) || (ch.incr_re == 0 && ch.incr_im == 0)) {
            return MFM_RC_IDENT; /* direct_fir.c:406 skips the derotation altogether for a zero increment */
        }
        if (ch.incr_re == -16384 && ch.incr_im == 0) {
            return MFM_RC_FLIP;
        }
        return (ch.incr_re == 0 && (ch.incr_im == 16384 || ch.incr_im == -16384)) ? MFM_RC_QUARTER : MFM_RC_GENERAL;
    };
    /* quarter turns per output: rot after k outputs = j^(turns * k) * 16384 for the exact classes */
    auto chan_turns = [](const Channel &ch) -> uint32_t {
        return ch.incr_im == 16384 ? 1u : ch.incr_im == -16384 ? 3u : ch.incr_re == -16384 ? 2u : 0u;
    };
    std::vector<uint32_t> perm(C);
    for (uint32_t c = 0; c < C; c++) {
        perm[c] = c;
    }
    if (e->use_v3) {
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
            return chan_class(e->chans[a]) > chan_class(e->chans[b]);
        });
    }
    e->rot_exact_channels = 0;
    for (uint32_t c = 0; c < C; c++) {
        e->rot_exact_channels += chan_class(e->chans[c]) != MFM_RC_GENERAL ? 1u : 0u;
    }
    e->v_rc = MFM_RC_IDENT;
    e->rot_fast_slices = 0;
    for (uint32_t sl = 0; sl < e->m_nslices && e->use_v3; sl++) {
        uint32_t cls = MFM_RC_IDENT; /* rows past the last channel have zero taps and store nothing */
        for (uint32_t c = sl * 64u; c < std::min(C, sl * 64u + 64u); c++) {
            cls = std::min(cls, chan_class(e->chans[perm[c]]));
        }
        e->v_rc = std::min(e->v_rc, cls);
        e->rot_fast_slices += cls != MFM_RC_GENERAL ? 1u : 0u;
    }
    if (!e->use_v3 || e->any_iq || e->v_layout >= 2u) {
        e->v_rc = MFM_RC_GENERAL; /* the exact-rotator instances are built without the filtered-IQ output, and for the
                                     sub-plane / chunk-row geometries only */
    }

    if (e->use_mfma) {
        const uint32_t kq = e->m_ks, row_bytes = e->m_row_bytes;
            /* W[2c] = (cr0,-ci0,cr1,-ci1..), W[2c+1] = (ci0,cr0,ci1,cr1..) (filter/complex.h:40-46) */
            const uint32_t K = 64u * kq;
            auto w_at = [&](uint32_t row, uint32_t k) -> int32_t {
                /* element k of a window = byte k % row_bytes of LDS row k / row_bytes: sample (k / row_bytes) * D +
                 * (k % row_bytes) / 2 while the byte is inside the 2 * D sample bytes, padding (zero tap) behind them */
                const uint32_t c = row / 2u, pos = k % row_bytes, i = (k / row_bytes) * D + pos / 2u;
                if (c >= C || pos >= 2u * D || i >= T) {
                    return 0;
                }
                const int32_t cr = e->chans[perm[c]].cre[i], ci = e->chans[perm[c]].cim[i];
                if (row & 1u) {
                    return (k & 1u) ? cr : ci;
                }
                return (k & 1u) ? -ci : cr;
            };
            /* v_mfma_i32_16x16x64_i8 A operand: lane (kg = lane >> 4, i = lane & 15) holds row i,
             * elements 64*kq + 16*kg + j, j = 0..15 */
            e->m_ah_mask = 0;
            afrag.assign((size_t)e->m_nrb * kq * 2 * 64 * 4, 0u);
            krow.assign((size_t)e->m_nrb * 16, 0);
            uint8_t *ab = reinterpret_cast<uint8_t *>(afrag.data());
            for (uint32_t rb = 0; rb < e->m_nrb; rb++) {
                for (uint32_t i = 0; i < 16; i++) {
                    const uint32_t row = rb * 16u + i;
                    uint32_t sum = 0;
                    for (uint32_t k = 0; k < K; k++) {
                        const int32_t w = w_at(row, k);
                        sum += (uint32_t)w;
                        const int32_t wl = (int8_t)(w & 0xff);
                        const int32_t wh = (w - wl) >> 8;
                        const uint32_t kst = k / 64u, gg = (k % 64u) / 16u, j = k % 16u;
                        const uint32_t ln = gg * 16u + i;
                        const size_t base = ((((size_t)rb * kq + kst) * 2u) * 64u + ln) * 16u + j;
                        if (wh != 0) {
                            e->m_ah_mask |= 1u << kst;
                        }
                        ab[base] = (uint8_t)(int8_t)wh;              /* plane 0: high bytes */
                        ab[base + 64u * 16u] = (uint8_t)(int8_t)wl;  /* plane 1: low bytes */
                    }
                    /* El = (e & 255) - 128 puts 128 * sum(W) into every product sum; 8192 is the rounding
                     * bias of the first round_q30_q15 (filter/complex.h:30-34), added here once */
                    krow[(size_t)rb * 16 + i] = (int32_t)(128u * sum + 8192u);
                }
            }
    }

    if (e->use_v3 && 3u == e->v_layout) {
        /* the long-filter kernel's fragments: v_kq k-steps per row block, in the order v_kperm */
        if (e->m_ah_mask != ([&] { uint32_t m = 0; for (uint32_t j = 0; j < e->v_nh; j++) { m |= 1u << ((e->v_kperm[j >> 2] >> (8u * (j & 3u))) & 0xffu); } return m; })()) {
            return fail(MFM_E_INVAL, "internal: the tap-plane mask changed between planning and building the fragments");
        }
        const size_t step_dw = 2u * 64u * 4u; /* dwords of one k-step: two planes x 64 lanes x 16 bytes */
        std::vector<uint32_t> af3((size_t)e->m_nrb * e->v_kq * step_dw, 0u);
        for (uint32_t rb = 0; rb < e->m_nrb; rb++) {
            for (uint32_t j = 0; j < e->v_kq; j++) {
                const uint32_t src = (e->v_kperm[j >> 2] >> (8u * (j & 3u))) & 0xffu;
                if (src < e->m_ks) { /* (a step past the laid-out ones holds zero taps) */
                    memcpy(&af3[((size_t)rb * e->v_kq + j) * step_dw], &afrag[((size_t)rb * e->m_ks + src) * step_dw], step_dw * 4u);
                }
            }
        }
        afrag.swap(af3);
    }

    /* ---- rotator tables (one per distinct increment) ---- */
    std::map<std::pair<int16_t, int16_t>, std::pair<uint64_t, std::pair<uint32_t, uint32_t>>> seen;
    std::vector<uint2> rot;
    for (Channel &ch : e->chans) {
        const auto key = std::make_pair(ch.incr_re, ch.incr_im);
        auto it = seen.find(key);
        if (it == seen.end()) {
            uint32_t mu = 0, lam = 1;
            int16_t ir = ch.incr_re, ii = ch.incr_im;
            if (0 == ir && 0 == ii) {
                /* direct_fir.c:406 skips derotation for a zero increment; a constant (16384,0)
                 * rotator is the identity through both Q14 roundings */
                ir = 16384;
                ii = 0;
            }
            if (!rot_cycle(ir, ii, kMaxRotEntries, &mu, &lam)) {
                return fail(MFM_E_INVAL, "rotator (%d,%d) has no cycle within %llu steps", ir, ii,
                            (unsigned long long)kMaxRotEntries);
            }
            /* unroll short cycles to at least one tile's worth of entries: a multiple of a period is
             * a period, and the kernels then fold an index with one conditional subtraction */
            lam = lam * ((kMaxOutputsPerTile + lam - 1) / lam);
            const uint64_t n = (uint64_t)mu + lam + kMaxOutputsPerTile;
            const uint64_t base = rot.size() + 1; /* one dummy entry in front: index -1 is readable */
            rot.resize(rot.size() + 1 + n);
            rot[base - 1] = make_uint2(0, 0);
            int16_t rr = 16384, ri = 0;
            for (uint64_t k = 0; k < n; k++) {
                if (ri == INT16_MIN) {
                    return fail(MFM_E_INVAL, "rotator state reached -32768");
                }
                rot[base + k] = make_uint2(mfm_pack16(rr, -(int32_t)ri), mfm_pack16(ri, rr));
                rot_step(rr, ri, ir, ii);
            }
            it = seen.emplace(key, std::make_pair(base, std::make_pair(mu, lam))).first;
        }
        ch.rot_base = it->second.first;
        ch.mu = it->second.second.first;
        ch.lam = it->second.second.second;
    }
    e->rot_entries = rot.size();
    if (rot.size() >= (1ull << 29)) {
        return fail(MFM_E_INVAL, "rotator tables need %zu entries (limit 2^29: byte offsets are 32-bit)", rot.size());
    }

    std::vector<mfm_chan_info> info((size_t)e->ngroups * MFM_CG);
    memset(info.data(), 0, info.size() * sizeof(mfm_chan_info));
    for (uint32_t c = 0; c < C; c++) {
        const Channel &ch = e->chans[perm[c]];
        info[c].rot_base = ch.rot_base;
        info[c].mu = ch.mu;
        info[c].lam = ch.lam;
        info[c].lam_magic = (uint32_t)std::min<uint64_t>(0xffffffffull, (1ull << 32) / ch.lam);
        info[c].out_row = perm[c];
        info[c].rot_class = chan_class(ch) | (chan_class(ch) != MFM_RC_GENERAL ? chan_turns(ch) << 4 : 0u);
    }

    /* ---- atan LUT: fast_atan2f.c:14-81, entries atan(i/255) at 7 significant digits ---- */
    float tbl[257];
    mfm_hosttwin_atan_table(tbl);
    if (!mfm_hosttwin_atan_table_ok()) {
        return fail(MFM_E_INVAL, "atan table self-check failed (host libm rounds atan() differently)");
    }
    std::vector<float2> lut(256);
    for (int i = 0; i < 256; i++) {
        lut[i] = make_float2(tbl[i], tbl[i + 1] - tbl[i]);
    }

    /* ---- device allocations ---- */
    e->committed = true; /* from here free_device() releases whatever was allocated */
    HIP_TRY(hipStreamCreateWithFlags(&e->s_in, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&e->s_compute, hipStreamNonBlocking));
    e->cs[0] = e->cs[1] = e->s_last = e->s_compute;
    e->ncs = 1;
    if ((e->cfg.flags & MFM_F_OVERLAP) && e->use_v3) {
        HIP_TRY(hipStreamCreateWithFlags(&e->cs[1], hipStreamNonBlocking));
        e->ncs = 2;
        for (int i = 0; i < kMaxInBufs; i++) {
            HIP_TRY(hipEventCreateWithFlags(&e->tail_done[i], hipEventDisableTiming));
        }
    }
    HIP_TRY(hipStreamCreateWithFlags(&e->s_out, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&e->in_ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->kernel_done, hipEventDisableTiming));

    HIP_TRY(hipMalloc(&e->d_coef, coef.size() * 4));
    HIP_TRY(hipMemcpy(e->d_coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&e->d_tapoff, tapoff.size() * 4));
    HIP_TRY(hipMemcpy(e->d_tapoff, tapoff.data(), tapoff.size() * 4, hipMemcpyHostToDevice));
    if (e->use_mfma) {
        HIP_TRY(hipMalloc(&e->d_afrag, afrag.size() * 4));
        HIP_TRY(hipMemcpy(e->d_afrag, afrag.data(), afrag.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&e->d_krow, krow.size() * 4));
        HIP_TRY(hipMemcpy(e->d_krow, krow.data(), krow.size() * 4, hipMemcpyHostToDevice));
    }
    /* 8-bit input read as it is (mfm_kernel_v3.hip, IN8): x = alpha * s + beta with s the byte as int8, so the row
     * constant is (beta * sum(W) + 8192) / alpha - exact for all three forms.  krow = 128 * sum(W) + 8192. */
    /* both matrix kernels have the form for it (not built for the filtered-IQ debug output) */
    e->raw8_ok = e->use_mfma && (!e->use_v3 || (e->v_nstage4 / 2u + 511u) / 512u <= 4u) && !e->any_iq &&
                !(e->cfg.flags & MFM_F_WIDEN_8BIT);
    if (e->raw8_ok) {
        std::vector<int32_t> k8(krow.size());
        for (int fmt = MFM_IN_CS8; fmt <= MFM_IN_RTLSDR_U8; fmt++) {
            for (size_t i = 0; i < krow.size(); i++) {
                const uint32_t sum = (uint32_t)(((int64_t)krow[i] - 8192) / 128); /* |sum(W)| < 2^24: no wrap in krow */
                k8[i] = fmt == MFM_IN_RTLSDR_U8 ? (int32_t)(sum + 64u)                  /* (128 * sum + 8192) / 128 */
                        : fmt == MFM_IN_CU8     ? (int32_t)(8192u - 127u * sum)           /* beta = -127 */
                                                : 8192;                                    /* cs8: beta = 0 */
            }
            HIP_TRY(hipMalloc(&e->d_krow8[fmt], k8.size() * 4));
            HIP_TRY(hipMemcpy(e->d_krow8[fmt], k8.data(), k8.size() * 4, hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMalloc(&e->d_tailtmp, ((size_t)T + D + 16u) * 2u));
    }
    HIP_TRY(hipMalloc(&e->d_info, info.size() * sizeof(mfm_chan_info)));
    HIP_TRY(hipMemcpy(e->d_info, info.data(), info.size() * sizeof(mfm_chan_info), hipMemcpyHostToDevice));
    if (e->use_v3 && mfm_rot_entry_bytes_v3() == 4u) {
        /* the second-generation kernel's build takes 4-byte entries (rr | ri << 16): half the table bytes per output */
        std::vector<uint32_t> rot4(rot.size());
        for (size_t i = 0; i < rot.size(); i++) {
            rot4[i] = mfm_pack16(mfm_lo16(rot[i].x), mfm_lo16(rot[i].y)); /* {(rr, -ri), (ri, rr)} -> (rr, ri) */
        }
        HIP_TRY(hipMalloc(&e->d_rot, rot4.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(e->d_rot, rot4.data(), rot4.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    } else {
        HIP_TRY(hipMalloc(&e->d_rot, rot.size() * sizeof(uint2)));
        HIP_TRY(hipMemcpy(e->d_rot, rot.data(), rot.size() * sizeof(uint2), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&e->d_lut, lut.size() * sizeof(float2)));
    HIP_TRY(hipMemcpy(e->d_lut, lut.data(), lut.size() * sizeof(float2), hipMemcpyHostToDevice));
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipMalloc(&e->d_state[i], info.size() * sizeof(mfm_chan_state)));
    }
    {
        int rc = write_state_fresh(e);
        if (rc != MFM_OK) {
            return rc;
        }
    }

    const size_t in_bytes = (size_t)e->cap_in * 4;
    e->own_in = e->cfg.ext_input[0] == nullptr;
    for (int i = 0; i < e->nbuf; i++) {
        if (e->own_in) {
            HIP_TRY(hipMalloc(&e->d_in[i], in_bytes));
        } else {
            e->d_in[i] = static_cast<uint32_t *>(e->cfg.ext_input[i]);
        }
        HIP_TRY(hipEventCreateWithFlags(&e->in_free[i], hipEventDisableTiming));
    }

    e->nslots = dev_only ? 2 : kOutSlots;
    for (int i = 0; i < e->nslots; i++) {
        OutSlot &s = e->slots[i];
        const size_t pcm_bytes = (size_t)C * e->out_stride * sizeof(int16_t);
        /* + a dump slot per lane behind the last channel row: the MFMA kernel stores unconditionally and sends
         * what is not an output there */
        HIP_TRY(hipMalloc(&s.d_pcm, pcm_bytes + 64 * sizeof(int16_t)));
        if (e->any_iq) {
            HIP_TRY(hipMalloc(&s.d_iq, pcm_bytes * 2 + 64 * sizeof(uint32_t)));
        }
        if (!dev_only) {
            HIP_TRY(hipHostMalloc(&s.h_pcm, pcm_bytes, hipHostMallocDefault));
            if (e->any_iq) {
                HIP_TRY(hipHostMalloc(&s.h_iq, pcm_bytes * 2, hipHostMallocDefault));
            }
        }
        HIP_TRY(hipEventCreateWithFlags(&s.ready, hipEventDisableTiming));
    }
    if ((e->cfg.flags & MFM_F_TIMING) && e->use_v3) {
        HIP_TRY(hipMalloc(&e->d_cyc, kCycleRing * 2 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(e->d_cyc, 0, kCycleRing * 2 * sizeof(unsigned long long)));
    }
    if (e->cfg.flags & MFM_F_TIMING) {
        e->timing_events = true;
        for (int i = 0; i < kTimingPairs; i++) {
            HIP_TRY(hipEventCreate(&e->t0[i]));
            HIP_TRY(hipEventCreate(&e->t1[i]));
        }
    }
    {
        int rc = select_kernels(e);
        if (rc == MFM_OK) {
            rc = division_selftest(e->cfg.device, e->use_v3 ? 2 : e->use_mfma ? 1 : 0);
        }
        if (rc != MFM_OK) {
            return rc;
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    return MFM_OK;
}

/* ---- the data path: accept blocks into the buffer being filled, launch it ---- */

} /* extern "C" */

namespace {

/* format of what the buffer being filled holds ([history | accepted blocks]); -1 = nothing, any format can start it */
int buffer_format(const mfm_engine *e)
{
    if (e->pend) {
        return e->in_fmt[e->cur_in];
    }
    return (e->hist + e->tail) ? e->tail_fmt : -1;
}

bool event_done(hipEvent_t ev)
{
    if (!ev) {
        return true;
    }
    if (hipEventQuery(ev) == hipErrorNotReady) {
        (void)hipGetLastError(); /* "not ready" must not surface in somebody's hipGetLastError() check */
        return false;
    }
    return true;
}

/* launches queued or running: 0, 1, or 2 for "two or more".  The launch before the one that last read buffer b ended
 * before it (one compute stream, in order). */
int launches_in_flight(mfm_engine *e)
{
    const int b1 = (e->cur_in + e->nbuf - 1) % e->nbuf, b2 = (e->cur_in + e->nbuf - 2) % e->nbuf;
    if (event_done(e->in_free_wait[b1])) {
        return 0;
    }
    if (b2 == e->cur_in) {
        return 1; /* two buffers: acquire_input() waited for the launch before that one */
    }
    return event_done(e->in_free_wait[b2]) ? 1 : 2;
}

struct SubmitPlan {
    bool must; /* the buffer has to be launched once this block is in (no coalescing, or coalesce_samples gathered) */
    bool may;  /* a launch would find a free output slot */
    bool want; /* the policy of mfm_engine_config::coalesce_samples would launch now although it does not have to */
    uint32_t n_new;
};

/* what accepting nr_samples more (0: what is there) would mean.  Takes the lock for the slot's state. */
void plan_block(mfm_engine *e, size_t nr_samples, SubmitPlan *p, bool locked)
{
    const uint32_t T = e->nr_taps, D = e->cfg.decimation, co = e->cfg.coalesce_samples;
    const uint64_t pend_after = (uint64_t)e->pend + nr_samples;
    const uint64_t n_avail = (uint64_t)e->tail + pend_after;
    p->n_new = n_avail >= T ? (uint32_t)((n_avail - T) / D + 1) : 0;
    p->must = 0 == co || pend_after >= co;
    if (e->cfg.flags & MFM_F_DEVICE_ONLY) {
        p->may = true;
    } else if (locked) {
        p->may = 0 == p->n_new || e->slots[e->submit_seq % e->nslots].state == OutSlot::FREE;
    } else {
        std::lock_guard<std::mutex> guard(e->mu);
        p->may = 0 == p->n_new || e->slots[e->submit_seq % e->nslots].state == OutSlot::FREE;
    }
    p->want = false;
    if (!p->must && p->n_new && !(e->cfg.flags & MFM_F_GATHER)) {
        const int fl = launches_in_flight(e);
        p->want = 0 == fl || (1 == fl && 4u * pend_after >= e->last_launch_samples);
    }
}

/*
 * One pass over d_in[cur_in] = [history tail | the blocks accepted since the last launch]: the kernel, the carry of the
 * unconsumed samples to the front of the next buffer, the copy of the outputs to the host mirror.  Caller holds e->mu and
 * has checked that an output slot is free.
 */
int launch_locked(mfm_engine *e)
{
    const uint32_t T = e->nr_taps, D = e->cfg.decimation, C = (uint32_t)e->chans.size();
    const bool dev_only = (e->cfg.flags & MFM_F_DEVICE_ONLY) != 0;
    const int cur = e->cur_in, nxt = (cur + 1) % e->nbuf;
    const uint32_t n_avail = e->tail + e->pend;
    const uint32_t n_new = n_avail >= T ? (n_avail - T) / D + 1 : 0;
    const int fmt = e->in_fmt[cur];
    const bool raw8 = fmt != MFM_IN_CS16;
    /* the stream of this launch, and of the one after it (the same one unless MFM_F_OVERLAP alternates them): what the
     * next launch needs from this buffer - the carry - is queued there */
    hipStream_t S = e->cs[e->si];
    /* Two streams pay off when a launch fills the workgroup slots more than once (its ragged end is what the next launch
     * moves into); a launch of one tile per slot or less costs more in the extra packets of the carry (a copy, two events)
     * than it gains: it keeps the carry in the kernel and the next launch stays behind it on the same stream. */
    bool two = false;
    if (e->ncs > 1u && e->use_v3 && n_new) {
        const uint32_t ntiles = (n_new + MFM_V3_OT - 1u) / MFM_V3_OT, slots = 256u * e->v_wg_per_cu;
        /* ... and only when what its carry copy reads - the last consumed row and the unconsumed samples behind it - lies
         * behind the [hist | tail] front of this buffer, which the previous launch's carry wrote on the OTHER stream: the
         * copy below is then ordered behind everything it reads (the H2D copies, through in_ready) without a wait of its
         * own.  True for every launch of more than a tile per slot; spelled out so that it does not rest on that. */
        two = (uint64_t)ntiles * e->v_nslices > slots && (uint64_t)n_new * D >= (uint64_t)e->tail + D;
    }
    hipStream_t S_after = two ? e->cs[e->si ^ 1u] : S;
    if (e->ncs > 1u && !two && e->in_free_wait[nxt]) {
        /* this launch (or its stream's copy) writes the carry into the next buffer: whoever read that one last - possibly on
         * the other stream - must be through */
        HIP_TRY(hipStreamWaitEvent(S, e->in_free_wait[nxt], 0));
    }

    OutSlot *slot = nullptr;
    int slot_idx = -1;
    if (n_new) {
        slot_idx = (int)(e->submit_seq % e->nslots);
        slot = &e->slots[slot_idx];
        if (!dev_only && slot->state != OutSlot::FREE) {
            return fail(MFM_E_BUSY, "all %d output slots hold unfetched blocks", e->nslots);
        }
    }
    if (e->wait_copy_stream) {
        HIP_TRY(hipEventRecord(e->in_ready, e->s_in));
        for (uint32_t i = 0; i < e->ncs; i++) {
            HIP_TRY(hipStreamWaitEvent(e->cs[i], e->in_ready, 0));
        }
        e->wait_copy_stream = false;
    }

    bool tail_in_kernel = false;
    hipEvent_t timing_end = nullptr;
    if (n_new) {
        const uint32_t ot = 64u * e->opl;
        mfm_launch L{};
        L.x = e->d_in[cur];
        L.n_avail = n_avail;
        L.n_new = n_new;
        L.decim = D;
        L.nchunks = e->nchunks;
        L.nstage = (ot - 1) * D + T;
        L.rs2 = e->rs2;
        L.lut_off = e->lut_off;
        L.ngroups = e->ngroups;
        L.gpw = e->gpw;
        L.nslices = e->nslices;
        L.ntiles = (n_new + ot - 2) / (ot - 1);
        L.nchan = C;
        L.out_stride = e->out_stride;
        L.coef = e->d_coef;
        L.tapoff = e->d_tapoff;
        L.info = e->d_info;
        L.rot = e->d_rot;
        L.st_in = e->d_state[e->parity];
        L.st_out = e->d_state[e->parity ^ 1];
        L.lut = e->d_lut;
        L.pcm = slot->d_pcm;
        L.iq_dbg = e->any_iq ? slot->d_iq : nullptr;

        if (!dev_only) {
            /* the previous D2H out of this slot must have drained before the kernel rewrites it */
            HIP_TRY(hipStreamWaitEvent(S, slot->ready, 0));
        }
        /* MFM_F_TIMING_SPARSE: every fourth launch carries the event pair.  An event record is a packet the command
         * processor handles between two kernels (~4 us each on MI355X): bracketing every launch of a back-to-back
         * stream costs 6 % of a 0.12 ms step, a sample of them costs under 1 % and measures the same kernel. */
        const bool timing = (e->cfg.flags & MFM_F_TIMING) != 0 &&
                            (!(e->cfg.flags & MFM_F_TIMING_SPARSE) || (e->launches % kSparseTiming) == 0);
        int ti = 0;
        if (timing) {
            int rc = fold_timing(e, false);
            if (rc != MFM_OK) {
                return rc;
            }
            ti = (int)(e->t_head % kTimingPairs);
            HIP_TRY(hipEventRecord(e->t0[ti], S));
        }
        if (e->use_v3) {
            mfm_launch_v3 V{};
            fill_v3(e, fmt, V);
            V.x = e->d_in[cur];
            V.n_avail = n_avail;
            V.n_new = n_new;
            V.hist = e->hist;
            V.k_base = e->outputs;
            V.ntiles = (n_new + MFM_V3_OT - 1u) / MFM_V3_OT;
            /* chunks of consecutive tiles, `rounds` per workgroup slot and slice, lengths equal to within one tile
             * (a chunk pays one extra column group) */
            /* (the long-filter kernel's small 8-bit instances are built for two workgroups per CU, the others for one) */
            const uint32_t wg_per_cu = (3u == e->v_layout && 2u * e->v_lds_bytes <= 160u * 1024u) ? mfm_v3l_wg_per_cu(&V) : e->v_wg_per_cu;
            const uint32_t slots = 256u * wg_per_cu;
            /* (a multiple of 8: the items are dealt chunk-major over the eight XCDs - item = 8 * (chunk / 8 * nslices + slice) +
             * chunk % 8, mfm3_decode_item - so a chunk count that is not one leaves holes in the last group of eight, the grid
             * overflows the slots and a few workgroups run a SECOND chunk while the others are done: with 3, 5, 6 or 12 slices
             * (130-192, 257-320, ... channels) a launch took up to twice as long as its work, profiles/r06_ab_chunking.txt) */
            uint32_t per_slice = std::max(1u, slots / V.nslices);
            per_slice = per_slice >= 8u ? per_slice & ~7u : per_slice;
            /* one chunk per slot: shorter chunks (3..10 tiles, several rounds) were 2-6 % slower on MI355X */
            V.nchunks = std::min(V.ntiles, per_slice);
            V.cl = (V.ntiles + V.nchunks - 1u) / V.nchunks;
            V.nitems = ((V.nchunks + 7u) / 8u) * 8u * V.nslices;
            /* the next launch's [hist | history tail]: the last consumed row and what was not consumed */
            V.tail_src = e->hist + n_new * D - D;
            V.tail_n = D + n_avail - n_new * D;
            V.tail_dst = e->d_in[nxt];
            tail_in_kernel = true;
            if (two) {
                V.tail_n = 0; /* the next launch must not wait for this kernel: the carry is copied on its own stream below */
                tail_in_kernel = false;
            }
            V.pcm = slot->d_pcm;
            V.iq_dbg = L.iq_dbg;
            if (e->d_cyc) {
                V.cyc = e->d_cyc + 2u * (e->launches % kCycleRing);
                V.cyc_tag = (uint32_t)((e->launches + 1u) & 0xffffffu);
                /* ... and zeroes the slot half a ring ahead, long before that launch folds its maxima into it: a slot never
                 * holds an older launch's stamp when its turn comes, whatever the 24-bit tags compare like after 2^24 launches */
                V.cyc_clear = e->d_cyc + 2u * ((e->launches + kCycleRing / 2u) % kCycleRing);
            }
            if (raw8) {
                e->launches_8bit++;
            }
            const uint32_t grid = std::min(V.nitems, slots);
            HIP_TRY(mfm_launch_channel_kernel_v3(e->kfn[fmt], &V, e->v_lds_bytes, grid, S));
            L.ntiles = grid; /* for grid_last below */
            L.nslices = 1;
        } else if (e->use_mfma) {
            mfm_launch_mfma M{};
            fill_mfma(e, fmt, M);
            M.x = e->d_in[cur];
            M.n_avail = n_avail;
            M.n_new = n_new;
            M.ntiles = (n_new + e->m_ot - 1u) / e->m_ot;
            M.nitems = ((M.ntiles + 7u) / 8u) * 8u * M.nslices;
            M.tail_src = n_new * D;
            M.tail_n = n_avail - n_new * D;
            M.tail_dst = e->d_in[nxt];
            tail_in_kernel = true;
            M.st_in = L.st_in;
            M.st_out = L.st_out;
            M.pcm = slot->d_pcm;
            M.iq_dbg = L.iq_dbg;
            if (raw8) {
                e->launches_8bit++;
            }
            const uint32_t grid = std::min(M.nitems, 256u * e->m_wg_fmt[fmt]);
            HIP_TRY(mfm_launch_channel_kernel_mfma(e->kfn[fmt], &M, e->m_lds_bytes, grid, S));
            L.ntiles = grid; /* for grid_last below */
            L.nslices = 1;
        } else {
            HIP_TRY(mfm_launch_channel_kernel(e->kfn[MFM_IN_CS16], &L, e->lds_bytes, S));
        }
        if (timing) {
            HIP_TRY(hipEventRecord(e->t1[ti], S));
            timing_end = e->t1[ti];
            e->t_head++;
        }
        e->parity ^= 1;
        e->launches++;
        e->grid_last = e->use_mfma ? L.ntiles : ((L.ntiles + 7) / 8) * 8 * L.nslices;
    }

    /* carry the unconsumed tail (and, for the second-generation kernel, the last consumed row in front of it) to the front
     * of the next staging buffer (the MFMA kernels have done it themselves) */
    const uint32_t consumed = n_new * D;
    const uint32_t new_tail = n_avail - consumed;
    const uint32_t new_hist = (e->use_v3 && n_new) ? D : e->hist;
    if (new_hist + new_tail && !tail_in_kernel) {
        const size_t ss = raw8 ? 2 : 4;
        if (two && e->in_free_wait[nxt]) {
            /* the launch that last read the next buffer may be on the other stream */
            HIP_TRY(hipStreamWaitEvent(S_after, e->in_free_wait[nxt], 0));
        }
        if (two && e->tail_pending[nxt]) {
            /* ... and so may the carry copy OUT of that buffer, when a launch in between did not alternate:
             * a wait on a completed event costs nothing */
            HIP_TRY(hipStreamWaitEvent(S_after, e->tail_done[nxt], 0));
        }
        HIP_TRY(hipMemcpyAsync(e->d_in[nxt],
                               reinterpret_cast<const uint8_t *>(e->d_in[cur]) + ((size_t)e->hist + consumed - new_hist) * ss,
                               ((size_t)new_hist + new_tail) * ss, hipMemcpyDeviceToDevice, S_after));
        if (two) {
            /* this buffer is free once its kernel AND this copy are through (acquire_input waits for both) */
            HIP_TRY(hipEventRecord(e->tail_done[cur], S_after));
            e->tail_pending[cur] = true;
        }
    }
    /* Every event record is a packet the command processor handles between two kernels (about 4 us each on
     * MI355X).  When the kernel carried the tail itself and its end is already stamped by the timing event, that
     * event also says "this input buffer is free". */
    if (tail_in_kernel && timing_end) {
        e->in_free_wait[cur] = timing_end;
    } else {
        HIP_TRY(hipEventRecord(e->in_free[cur], S));
        e->in_free_wait[cur] = e->in_free[cur];
    }

    if (n_new) {
        slot->first_output = e->outputs;
        slot->nr_outputs = n_new;
        if (dev_only) {
            /* nothing waits on slot->ready in device-only mode: consumers are ordered by the stream */
        } else {
            HIP_TRY(hipEventRecord(e->kernel_done, S));
            HIP_TRY(hipStreamWaitEvent(e->s_out, e->kernel_done, 0));
            const size_t row = (size_t)e->out_stride * sizeof(int16_t);
            HIP_TRY(hipMemcpy2DAsync(slot->h_pcm, row, slot->d_pcm, row, (size_t)n_new * sizeof(int16_t), C,
                                     hipMemcpyDeviceToHost, e->s_out));
            if (e->any_iq) {
                HIP_TRY(hipMemcpy2DAsync(slot->h_iq, row * 2, slot->d_iq, row * 2, (size_t)n_new * 4, C,
                                         hipMemcpyDeviceToHost, e->s_out));
            }
            HIP_TRY(hipEventRecord(slot->ready, e->s_out));
            slot->state = OutSlot::INFLIGHT;
        }
        e->last_slot = slot_idx;
        e->s_last = S;
        e->submit_seq++;
        e->outputs += n_new;
    }

    if (two) {
        e->si ^= 1u;
    }
    e->last_launch_samples = n_avail;
    e->last_launch_buf = cur;
    e->last_launch_fmt = fmt;
    e->last_launch_hist = e->hist;
    e->hist = new_hist;
    e->tail = new_tail;
    e->tail_fmt = fmt;
    e->in_fmt[cur] = MFM_IN_CS16;
    e->pend = 0;
    e->cur_in = nxt;
    return MFM_OK;
}

/* the buffer being filled holds blocks of another format than the one about to be accepted (a capture that changes its
 * sample format mid-stream; a cu8 block of odd length): what is there goes out as a launch of its own first */
int flush_for_format(mfm_engine *e, int fmt)
{
    if (0 == e->pend || e->in_fmt[e->cur_in] == fmt) {
        return MFM_OK;
    }
    std::lock_guard<std::mutex> guard(e->mu);
    return launch_locked(e);
}

} /* namespace */

extern "C" {

int mfm_engine_acquire_input(struct mfm_engine *e, void **d_dst, size_t *capacity_samples)
{
    if (!e || !d_dst) {
        return fail(MFM_E_INVAL, "NULL argument");
    }
    if (!e->committed) {
        return fail(MFM_E_STATE, "commit first");
    }
    HIP_TRY(hipSetDevice(e->cfg.device));
    const int rc = flush_for_format(e, MFM_IN_CS16);
    if (rc != MFM_OK) {
        return rc;
    }
    /* the kernel that last read this buffer (nbuf launches ago) must be done with it */
    if (e->in_free_wait[e->cur_in]) {
        HIP_TRY(hipEventSynchronize(e->in_free_wait[e->cur_in]));
        e->in_free_wait[e->cur_in] = nullptr; /* waited for: later blocks of this buffer need not ask again */
    }
    if (e->tail_pending[e->cur_in]) {
        HIP_TRY(hipEventSynchronize(e->tail_done[e->cur_in]));
        e->tail_pending[e->cur_in] = false;
    }
    if (0 == e->pend) {
        e->in_fmt[e->cur_in] = MFM_IN_CS16; /* what a device producer writes; mfm_engine_stage() says otherwise for raw bytes */
    }
    *d_dst = e->d_in[e->cur_in] + e->hist + e->tail + e->pend;
    if (capacity_samples) {
        *capacity_samples = std::min<size_t>(e->cap_in - e->hist - e->tail - e->pend, e->cfg.max_block_samples);
    }
    return MFM_OK;
}

/* would the engine keep an 8-bit block of this format as bytes now?  (mfm_engine_internal.h) */
int mfm_engine_can_take_bytes(struct mfm_engine *e, int format, size_t nr_samples)
{
    /* not a cu8 block of odd length (file_if.c:146-150 widens its last sample differently), not behind a history or
     * behind accepted blocks of another format */
    if (!e || !e->committed || !e->raw8_ok || !(format == MFM_IN_CS8 || format == MFM_IN_CU8 || format == MFM_IN_RTLSDR_U8) ||
        (format == MFM_IN_CU8 && (nr_samples & 1u))) {
        return 0;
    }
    const int bf = buffer_format(e);
    return bf == -1 || bf == format;
}

int mfm_engine_acquire_input_bytes(struct mfm_engine *e, int format, void **d_dst, size_t *capacity_samples)
{
    if (!e || !d_dst) {
        return fail(MFM_E_INVAL, "NULL argument");
    }
    if (!e->committed) {
        return fail(MFM_E_STATE, "commit first");
    }
    if (format != MFM_IN_CS8 && format != MFM_IN_CU8 && format != MFM_IN_RTLSDR_U8) {
        return fail(MFM_E_INVAL, "not an 8-bit sample format: %d", format);
    }
    HIP_TRY(hipSetDevice(e->cfg.device));
    const int rc = flush_for_format(e, format);
    if (rc != MFM_OK) {
        return rc;
    }
    if (!mfm_engine_can_take_bytes(e, format, 0)) {
        return fail(MFM_E_STATE, "this engine cannot read format %d as bytes now (kernel variant, or a history of another "
                                 "format): widen the block and use mfm_engine_acquire_input", format);
    }
    if (e->in_free_wait[e->cur_in]) {
        HIP_TRY(hipEventSynchronize(e->in_free_wait[e->cur_in]));
        e->in_free_wait[e->cur_in] = nullptr;
    }
    if (e->tail_pending[e->cur_in]) {
        HIP_TRY(hipEventSynchronize(e->tail_done[e->cur_in]));
        e->tail_pending[e->cur_in] = false;
    }
    e->in_fmt[e->cur_in] = format;
    *d_dst = reinterpret_cast<uint8_t *>(e->d_in[e->cur_in]) + ((size_t)e->hist + e->tail + e->pend) * 2;
    if (capacity_samples) {
        *capacity_samples = std::min<size_t>(e->cap_in - e->hist - e->tail - e->pend, e->cfg.max_block_samples); /* the block limit is that of int16 blocks */
    }
    return MFM_OK;
}

/* mode: MFM_SUBMIT_AUTO - the engine decides whether the buffer is launched now (mfm_engine_config::coalesce_samples);
 * _DEFER / _LAUNCH - a device group has decided for all its shards (mfm_engine_internal.h) */
static int submit_impl(struct mfm_engine *e, size_t nr_samples, void *producer_stream, int wait_producer, int mode, bool run);

int mfm_engine_submit_mode(struct mfm_engine *e, size_t nr_samples, void *producer_stream, int wait_producer, int mode)
{
    return submit_impl(e, nr_samples, producer_stream, wait_producer, mode, false);
}

/* run: several blocks of at most max_block_samples each, staged next to each other, accepted as one (mfm_engine_push_pinned_run) */
static int submit_impl(struct mfm_engine *e, size_t nr_samples, void *producer_stream, int wait_producer, int mode, bool run)
{
    if (!e) {
        return fail(MFM_E_INVAL, "NULL engine");
    }
    if (!e->committed) {
        return fail(MFM_E_STATE, "commit first");
    }
    if (0 == nr_samples) {
        /* receiver_sample_buf_deliver() treats an empty buffer as a bug (receiver.c:84) */
        return fail(MFM_E_INVAL, "empty block");
    }
    if (nr_samples > (size_t)e->cap_in - e->hist - e->tail - e->pend || (!run && nr_samples > e->cfg.max_block_samples)) {
        return fail(MFM_E_INVAL, "block of %zu samples exceeds max_block_samples %u", nr_samples,
                    e->cfg.max_block_samples);
    }
    HIP_TRY(hipSetDevice(e->cfg.device));
    std::lock_guard<std::mutex> guard(e->mu);

    const int cur = e->cur_in;
    const int fmt = e->in_fmt[cur];
    const bool raw8 = fmt != MFM_IN_CS16;
    if (raw8 && !mfm_engine_can_take_bytes(e, fmt, nr_samples)) {
        if (0 == e->pend) {
            e->in_fmt[cur] = MFM_IN_CS16;
        }
        return fail(MFM_E_INVAL, "a block of %zu samples of format %d cannot be read as bytes here (cu8 blocks must be of "
                                 "even length)", nr_samples, fmt);
    }

    SubmitPlan plan;
    plan_block(e, nr_samples, &plan, true);
    const bool launch = mode == MFM_SUBMIT_LAUNCH || (mode == MFM_SUBMIT_AUTO && (plan.must || (plan.want && plan.may)));
    if (launch && !plan.may) {
        return fail(MFM_E_BUSY, "all %d output slots hold unfetched blocks", e->nslots);
    }

    if (wait_producer && static_cast<hipStream_t>(producer_stream) == e->s_in && !launch) {
        e->wait_copy_stream = true; /* the launch that takes this block waits for the copy stream, once for all it gathered */
    } else if (wait_producer) {
        HIP_TRY(hipEventRecord(e->in_ready, static_cast<hipStream_t>(producer_stream)));
        for (uint32_t i = 0; i < e->ncs; i++) {
            HIP_TRY(hipStreamWaitEvent(e->cs[i], e->in_ready, 0)); /* the launch on one, the carry out of it on the other */
        }
        if (static_cast<hipStream_t>(producer_stream) == e->s_in) {
            e->wait_copy_stream = false; /* everything staged before is covered too */
        }
    }

    if (!raw8 && 0 == e->pend && (e->hist + e->tail) && e->tail_fmt != MFM_IN_CS16) {
        /* the history at the front of this buffer is bytes, the block behind it int16: widen it where it stands (the
         * kernel that wrote it is ahead of this on the compute stream; the block's own samples start 4 * tail bytes in) */
        const uint32_t nh = e->hist + e->tail;
        hipStream_t Sn = e->cs[e->si]; /* the stream of this buffer's launch: the carry was queued there */
        HIP_TRY(hipMemcpyAsync(e->d_tailtmp, e->d_in[cur], (size_t)nh * 2, hipMemcpyDeviceToDevice, Sn));
        hipLaunchKernelGGL(mfm_unpack_kernel, dim3((nh / 8u + 256u) / 256u), dim3(256), 0, Sn, e->d_tailtmp,
                           e->d_in[cur], nh, e->tail_fmt, 0);
        HIP_TRY(hipGetLastError());
        e->tail_fmt = MFM_IN_CS16;
    }

    e->pend += (uint32_t)nr_samples;
    e->samples_in += nr_samples;
    e->submits++;
    return launch ? launch_locked(e) : MFM_OK;
}

int mfm_engine_submit(struct mfm_engine *e, size_t nr_samples, void *producer_stream, int wait_producer)
{
    return mfm_engine_submit_mode(e, nr_samples, producer_stream, wait_producer, MFM_SUBMIT_AUTO);
}

/* what accepting a block of nr_samples would mean for this engine (mfm_engine_internal.h) */
int mfm_engine_plan(struct mfm_engine *e, size_t nr_samples, int *must, int *may, int *want)
{
    if (!e || !e->committed) {
        return fail(MFM_E_STATE, "commit first");
    }
    if (hipSetDevice(e->cfg.device) != hipSuccess) {
        return fail(MFM_E_DEVICE, "hipSetDevice(%d) failed", e->cfg.device);
    }
    SubmitPlan p;
    plan_block(e, nr_samples, &p, false);
    *must = p.must ? 1 : 0;
    *may = p.may ? 1 : 0;
    *want = p.want ? 1 : 0;
    return MFM_OK;
}

int mfm_engine_format_conflict(struct mfm_engine *e, int fmt)
{
    return (e && e->committed && e->pend && e->in_fmt[e->cur_in] != fmt) ? 1 : 0;
}

int mfm_engine_flush(struct mfm_engine *e)
{
    if (!e || !e->committed) {
        return fail(MFM_E_STATE, "commit first");
    }
    HIP_TRY(hipSetDevice(e->cfg.device));
    /* pend is tested under the lock: two callers (a producer thread's idle flush and somebody else's sync) may get here at
     * once, and the second one must find nothing left to launch instead of launching an empty pass that rotates the
     * input buffers under the producer */
    std::lock_guard<std::mutex> guard(e->mu);
    if (0 == e->pend) {
        return MFM_OK;
    }
    return launch_locked(e);
}

/* would a block of nr_samples be accepted?  (refused before anything is staged, so a caller can drain and retry) */
static int check_output_room(struct mfm_engine *e, size_t nr_samples)
{
    SubmitPlan p;
    plan_block(e, nr_samples, &p, false);
    if (p.must && !p.may) {
        return fail(MFM_E_BUSY, "all %d output slots hold unfetched blocks", e->nslots);
    }
    return MFM_OK;
}

/*
 * Host -> device staging of one block WITHOUT submitting it (mfm_engine_internal.h): the samples (int16 pairs, or
 * 8-bit pairs widened on the device exactly as the reference's front ends widen them on the host) are placed where
 * the next submit() expects them, by work queued on the engine's copy stream.  *d_dst is that device address - what a
 * device group broadcasts to its other members before every member submits.
 */
int mfm_engine_stage(struct mfm_engine *e, const void *data, size_t nr_samples, int format, int flags, void **d_dst)
{
    const int allow_raw = flags & MFM_STAGE_ALLOW_RAW;
    const bool pinned = (flags & MFM_STAGE_PINNED) != 0; /* the caller's memory is page-locked: the copy engine reads it directly */
    if (!e || !data) {
        return fail(MFM_E_INVAL, "NULL argument");
    }
    if (format != MFM_IN_CS16 && format != MFM_IN_CS8 && format != MFM_IN_CU8 && format != MFM_IN_RTLSDR_U8) {
        return fail(MFM_E_INVAL, "unknown sample format %d", format);
    }
    if (!e->committed) {
        return fail(MFM_E_STATE, "commit first");
    }
    if (0 == nr_samples || nr_samples > e->cfg.max_block_samples) {
        return fail(MFM_E_INVAL, "block of %zu samples (max %u)", nr_samples, e->cfg.max_block_samples);
    }
    HIP_TRY(hipSetDevice(e->cfg.device));
    /* the matrix kernel reads the bytes themselves (mfm_kernel_v3.hip, IN8): half the HBM bytes, half the matrix
     * instructions, no widening pass.  Everything else goes the int16 way. */
    const bool raw = allow_raw && mfm_engine_can_take_bytes(e, format, nr_samples);
    int rc = flush_for_format(e, raw ? format : MFM_IN_CS16);
    if (rc != MFM_OK) {
        return rc;
    }
    rc = check_output_room(e, nr_samples);
    if (rc != MFM_OK) {
        return rc;
    }
    void *dst = nullptr;
    size_t cap = 0;
    rc = raw ? mfm_engine_acquire_input_bytes(e, format, &dst, &cap) : mfm_engine_acquire_input(e, &dst, &cap);
    if (rc != MFM_OK) {
        return rc;
    }
    if (cap < nr_samples) {
        return fail(MFM_E_INVAL, "the input buffer has room for %zu samples, the block has %zu", cap, nr_samples);
    }
    const int cur = e->cur_in;
    if (!pinned && !e->h_in[cur]) {
        HIP_TRY(hipHostMalloc(&e->h_in[cur], (size_t)e->cap_in * 4, hipHostMallocDefault));
    }
    /* acquire_input() waited for the kernel that consumed the previous contents of this buffer, so the copies out of
     * h_in[cur] that fed it are done; blocks accepted since then sit in front of this one, as they do on the device */
    uint8_t *h = pinned ? const_cast<uint8_t *>(static_cast<const uint8_t *>(data))
                        : reinterpret_cast<uint8_t *>(e->h_in[cur]) + (size_t)e->pend * 4;
    if (raw) {
        if (!pinned) {
            memcpy(h, data, nr_samples * 2);
        }
        HIP_TRY(hipMemcpyAsync(dst, h, nr_samples * 2, hipMemcpyHostToDevice, e->s_in));
    } else if (format == MFM_IN_CS16) {
        if (!pinned) {
            memcpy(h, data, nr_samples * 4);
        }
        HIP_TRY(hipMemcpyAsync(dst, h, nr_samples * 4, hipMemcpyHostToDevice, e->s_in));
    } else {
        if (!e->d_raw[cur]) {
            HIP_TRY(hipMalloc(&e->d_raw[cur], (size_t)e->cap_in * 2 + 64));
        }
        /* the unpack kernels of the previous use of d_raw[cur] ran before the kernel acquire_input() waited for */
        uint16_t *draw = e->d_raw[cur] + ((e->pend + 7u) & ~7u); /* the unpack kernel reads 16-byte groups */
        if (!pinned) {
            memcpy(h, data, nr_samples * 2);
        }
        HIP_TRY(hipMemcpyAsync(draw, h, nr_samples * 2, hipMemcpyHostToDevice, e->s_in));
        uint32_t blocks = (uint32_t)((nr_samples / 8 + 255) / 256);
        blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
        hipLaunchKernelGGL(mfm_unpack_kernel, dim3(blocks), dim3(256), 0, e->s_in, draw,
                           static_cast<uint32_t *>(dst), (uint32_t)nr_samples, format, 1);
        HIP_TRY(hipGetLastError());
    }
    if (pinned) {
        e->copy_seq++; /* this push's ticket (mfm_engine_copy_done / _wait) */
    }
    if (d_dst) {
        *d_dst = dst;
    }
    return MFM_OK;
}

uint64_t mfm_engine_copy_ticket(struct mfm_engine *e)
{
    return e ? e->copy_seq : 0;
}

/* is ticket through?  wait: block until it is.  Looks at the events that cover it; records one if none does. */
static int copy_ticket_state(struct mfm_engine *e, uint64_t ticket, bool wait)
{
    if (!e || !e->committed || ticket > e->copy_seq) {
        return fail(MFM_E_INVAL, "no such ticket");
    }
    if (ticket <= e->copy_done_seq) {
        return 1;
    }
    HIP_TRY(hipSetDevice(e->cfg.device));
    /* the oldest recorded event that covers the ticket answers soonest */
    int best = -1;
    for (int i = 0; i < (int)mfm_engine::kCopyRing; i++) {
        if (e->copy_ev_seq[i] >= ticket && (best < 0 || e->copy_ev_seq[i] < e->copy_ev_seq[best])) {
            best = i;
        }
    }
    if (best < 0) {
        /* none yet: one behind everything staged so far.  Its slot's previous event is the oldest one; if that is still
         * pending it covers only older tickets, and re-recording it loses nothing we are asked about. */
        best = (int)(e->copy_ev_next++ % mfm_engine::kCopyRing);
        if (!e->copy_ev[best]) {
            HIP_TRY(hipEventCreateWithFlags(&e->copy_ev[best], hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(e->copy_ev[best], e->s_in));
        e->copy_ev_seq[best] = e->copy_seq;
    }
    if (wait) {
        HIP_TRY(hipEventSynchronize(e->copy_ev[best]));
    } else if (!event_done(e->copy_ev[best])) {
        return 0;
    }
    e->copy_done_seq = std::max(e->copy_done_seq, e->copy_ev_seq[best]);
    return 1;
}

int mfm_engine_copy_done(struct mfm_engine *e, uint64_t ticket)
{
    return copy_ticket_state(e, ticket, false);
}

int mfm_engine_copy_wait(struct mfm_engine *e, uint64_t ticket)
{
    const int rc = copy_ticket_state(e, ticket, true);
    return rc < 0 ? rc : MFM_OK;
}

int mfm_engine_push_pinned(struct mfm_engine *e, const void *data, size_t nr_samples, int format, uint64_t *ticket)
{
    int rc = mfm_engine_stage(e, data, nr_samples, format, MFM_STAGE_ALLOW_RAW | MFM_STAGE_PINNED, nullptr);
    if (rc != MFM_OK) {
        return rc;
    }
    if (ticket) {
        *ticket = e->copy_seq;
    }
    return mfm_engine_submit(e, nr_samples, e->s_in, 1);
}

size_t mfm_engine_input_room(struct mfm_engine *e)
{
    if (!e || !e->committed) {
        return 0;
    }
    std::lock_guard<std::mutex> guard(e->mu);
    return (size_t)e->cap_in - e->hist - e->tail - e->pend;
}

/*
 * A run of `count` page-locked buffers of nr_samples_each samples that lie stride_bytes apart (the receiver's pool hands its
 * frames out in address order, so buffers delivered one after the other are neighbours in the arena): ONE strided copy command
 * packs them into the input buffer, and they are accepted as one block.  A copy command per 512 KiB sample_buf runs at half the
 * link's rate, one per 16 KiB file_if buffer at a ninth (bench.py end_to_end.link).  *accepted = how many of the buffers were
 * taken - fewer than `count` when the buffer being filled has less room, 1 where a block has to be widened on the device.
 */
int mfm_engine_push_pinned_run(struct mfm_engine *e, const void *first, size_t stride_bytes, size_t nr_samples_each, size_t count,
                               int format, uint64_t *ticket, size_t *accepted)
{
    if (accepted) {
        *accepted = 0;
    }
    if (!e || !first || 0 == count || 0 == nr_samples_each) {
        return fail(MFM_E_INVAL, "NULL or empty run");
    }
    if (!e->committed) {
        return fail(MFM_E_STATE, "commit first");
    }
    const size_t bps = (format == MFM_IN_CS16) ? 4u : 2u;
    if (nr_samples_each > e->cfg.max_block_samples || stride_bytes < nr_samples_each * bps) {
        return fail(MFM_E_INVAL, "run of blocks of %zu samples, %zu bytes apart (max block %u)", nr_samples_each, stride_bytes,
                    e->cfg.max_block_samples);
    }
    HIP_TRY(hipSetDevice(e->cfg.device));
    const bool raw = format != MFM_IN_CS16 && mfm_engine_can_take_bytes(e, format, nr_samples_each);
    size_t k = count;
    if (format != MFM_IN_CS16 && !raw) {
        k = 1; /* widened on the device through a scratch buffer: one block at a time */
    }
    if (k > 1) {
        /* what is there of another format goes out first (then the room is that of an empty buffer) */
        const int rcf = flush_for_format(e, raw ? format : MFM_IN_CS16);
        if (rcf != MFM_OK) {
            return rcf;
        }
        const size_t room = mfm_engine_input_room(e);
        k = std::min(k, room / nr_samples_each);
        /* a run must not gather past the point where the engine would have launched: what is gathered plus the run stays
         * within coalesce_samples + one block */
        const size_t co = e->cfg.coalesce_samples;
        if (co) {
            const size_t pend = (size_t)mfm_engine_pending_samples(e);
            const size_t until = co > pend ? (co - pend + nr_samples_each - 1u) / nr_samples_each : 1u;
            k = std::min(k, std::max<size_t>(until, 1u));
            /* ... and not at all where ONE buffer pushed by itself would launch right away (the device has nothing to do, or a
             * quarter of the running launch has gathered: plan_block's `want`): the first buffer goes alone, as per-buffer pushes
             * would have sent it, and the run behind it gathers under that launch */
            SubmitPlan one;
            plan_block(e, nr_samples_each, &one, false);
            if (one.want || one.must) {
                k = 1;
            }
        } else {
            k = 1; /* no gathering: every buffer is a launch of its own */
        }
    }
    if (k <= 1) {
        const int rc = mfm_engine_push_pinned(e, first, nr_samples_each, format, ticket);
        if (rc == MFM_OK && accepted) {
            *accepted = 1;
        }
        return rc;
    }
    const size_t total = nr_samples_each * k;
    int rc = check_output_room(e, total);
    if (rc != MFM_OK) {
        return rc;
    }
    void *dst = nullptr;
    size_t cap = 0;
    rc = raw ? mfm_engine_acquire_input_bytes(e, format, &dst, &cap) : mfm_engine_acquire_input(e, &dst, &cap);
    if (rc != MFM_OK) {
        return rc;
    }
    HIP_TRY(hipMemcpy2DAsync(dst, nr_samples_each * bps, first, stride_bytes, nr_samples_each * bps, k, hipMemcpyHostToDevice, e->s_in));
    e->copy_seq++; /* one ticket for the run: its copy is one command */
    if (ticket) {
        *ticket = e->copy_seq;
    }
    rc = submit_impl(e, total, e->s_in, 1, MFM_SUBMIT_AUTO, true);
    if (rc == MFM_OK && accepted) {
        *accepted = k;
    }
    return rc;
}

/* The host <-> device link by itself (include/multifm_hip.h): what a host-fed figure is to be read against. */
int mfm_link_probe(int device, size_t piece_bytes, size_t total_bytes, double d2h_per_h2d, double *h2d_GBps, double *d2h_GBps)
{
    return mfm_link_probe_runs(device, piece_bytes, 0, 1, total_bytes, d2h_per_h2d, h2d_GBps, d2h_GBps);
}

int mfm_link_probe_runs(int device, size_t piece_bytes, size_t gap_bytes, size_t pieces_per_command, size_t total_bytes,
                        double d2h_per_h2d, double *h2d_GBps, double *d2h_GBps)
{
    if (0 == piece_bytes || total_bytes < piece_bytes || d2h_per_h2d < 0.0 || d2h_per_h2d > 4.0 || 0 == pieces_per_command) {
        return fail(MFM_E_INVAL, "piece / total bytes, or the D2H share");
    }
    HIP_TRY(hipSetDevice(device));
    /* one arena of page-locked memory cut into pieces (gap_bytes apart: a sample_buf's header sits between the data of two
     * frames), like the receiver's sample_buf pool (host/mfm_receiver.c) */
    const size_t stride = piece_bytes + gap_bytes;
    size_t pieces_in_arena = std::max<size_t>(std::min<size_t>(total_bytes, (size_t)64 << 20) / stride, pieces_per_command);
    pieces_in_arena = pieces_in_arena / pieces_per_command * pieces_per_command;
    const size_t arena = pieces_in_arena * stride;
    const size_t back_piece = (size_t)((double)piece_bytes * d2h_per_h2d) & ~(size_t)15;
    uint8_t *h_src = nullptr, *h_dst = nullptr, *d_in = nullptr, *d_out = nullptr;
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, f0 = nullptr, f1 = nullptr;
    int rc = MFM_OK;
    auto body = [&]() -> int {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_src), arena, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_in), pieces_in_arena * piece_bytes));
        memset(h_src, 0x5a, arena);
        if (back_piece) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_dst), back_piece * pieces_in_arena, hipHostMallocDefault));
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_out), back_piece * pieces_in_arena));
            HIP_TRY(hipMemset(d_out, 0, back_piece * pieces_in_arena));
        }
        HIP_TRY(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventCreate(&f0));
        HIP_TRY(hipEventCreate(&f1));
        const size_t n = total_bytes / piece_bytes;
        for (int pass = 0; pass < 2; pass++) { /* the first pass warms the mappings up */
            HIP_TRY(hipEventRecord(e0, s_in));
            HIP_TRY(hipEventRecord(f0, s_out));
            for (size_t i = 0; i + pieces_per_command <= n; i += pieces_per_command) {
                const size_t k = i % pieces_in_arena;
                if (1 == pieces_per_command) {
                    HIP_TRY(hipMemcpyAsync(d_in + k * piece_bytes, h_src + k * stride, piece_bytes, hipMemcpyHostToDevice, s_in));
                } else {
                    /* a run of frames that lie next to each other in the arena: ONE strided command, packed on the device */
                    HIP_TRY(hipMemcpy2DAsync(d_in + k * piece_bytes, piece_bytes, h_src + k * stride, stride, piece_bytes,
                                             pieces_per_command, hipMemcpyHostToDevice, s_in));
                }
                if (back_piece) {
                    HIP_TRY(hipMemcpyAsync(h_dst + k * back_piece, d_out + k * back_piece, back_piece * pieces_per_command,
                                           hipMemcpyDeviceToHost, s_out));
                }
            }
            HIP_TRY(hipEventRecord(e1, s_in));
            HIP_TRY(hipEventRecord(f1, s_out));
            HIP_TRY(hipStreamSynchronize(s_in));
            HIP_TRY(hipStreamSynchronize(s_out));
        }
        float ms_in = 0.0f, ms_out = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms_in, e0, e1));
        HIP_TRY(hipEventElapsedTime(&ms_out, f0, f1));
        const size_t done = n / pieces_per_command * pieces_per_command;
        if (h2d_GBps) {
            *h2d_GBps = (double)(done * piece_bytes) / ((double)ms_in * 1e6);
        }
        if (d2h_GBps) {
            *d2h_GBps = back_piece ? (double)(done * back_piece) / ((double)ms_out * 1e6) : 0.0;
        }
        return MFM_OK;
    };
    rc = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (f0) (void)hipEventDestroy(f0);
    if (f1) (void)hipEventDestroy(f1);
    if (s_in) (void)hipStreamDestroy(s_in);
    if (s_out) (void)hipStreamDestroy(s_out);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    (void)hipHostFree(h_src);
    (void)hipHostFree(h_dst);
    return rc;
}

void *mfm_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (0 == bytes || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void mfm_host_free(void *p)
{
    if (p) {
        (void)hipHostFree(p);
    }
}

void *mfm_engine_copy_stream(struct mfm_engine *e)
{
    return (e && e->committed) ? e->s_in : nullptr;
}

int mfm_engine_pending_blocks(struct mfm_engine *e)
{
    if (!e || !e->committed) {
        return 0;
    }
    std::lock_guard<std::mutex> guard(e->mu);
    return (int)(e->submit_seq - e->fetch_seq);
}

int mfm_engine_pending_samples(struct mfm_engine *e)
{
    if (!e || !e->committed) {
        return 0;
    }
    std::lock_guard<std::mutex> guard(e->mu);
    return (int)e->pend;
}

int mfm_engine_push(struct mfm_engine *e, const int16_t *iq, size_t nr_samples)
{
    const int rc = mfm_engine_stage(e, iq, nr_samples, MFM_IN_CS16, 0, nullptr);
    return rc != MFM_OK ? rc : mfm_engine_submit(e, nr_samples, e->s_in, 1);
}

int mfm_engine_push_bytes(struct mfm_engine *e, const void *bytes, size_t nr_samples, int format)
{
    const int rc = mfm_engine_stage(e, bytes, nr_samples, format, MFM_STAGE_ALLOW_RAW, nullptr);
    return rc != MFM_OK ? rc : mfm_engine_submit(e, nr_samples, e->s_in, 1);
}

int mfm_engine_last_launch_input(struct mfm_engine *e, void **d_in, size_t *nr_samples, int *format)
{
    if (!e || !e->committed || e->last_launch_buf < 0) {
        return fail(MFM_E_STATE, "no launch yet");
    }
    if (d_in) {
        /* the first sample the launch had not consumed before (the second-generation kernel keeps a row in front of it) */
        *d_in = reinterpret_cast<uint8_t *>(e->d_in[e->last_launch_buf]) +
                (size_t)e->last_launch_hist * (e->last_launch_fmt == MFM_IN_CS16 ? 4 : 2);
    }
    if (nr_samples) {
        *nr_samples = e->last_launch_samples;
    }
    if (format) {
        *format = e->last_launch_fmt;
    }
    return MFM_OK;
}

int mfm_engine_replay(struct mfm_engine *e, size_t block_samples, size_t nr_blocks)
{
    for (size_t i = 0; i < nr_blocks; i++) {
        void *dst = nullptr;
        size_t cap = 0;
        int rc = mfm_engine_acquire_input(e, &dst, &cap);
        if (rc != MFM_OK) {
            return rc;
        }
        if (cap < block_samples) {
            return fail(MFM_E_INVAL, "replay: block of %zu samples, room for %zu", block_samples, cap);
        }
        rc = mfm_engine_submit(e, block_samples, nullptr, 0);
        if (rc != MFM_OK) {
            return rc;
        }
    }
    return MFM_OK;
}


int mfm_engine_fetch(struct mfm_engine *e, struct mfm_block *blk)
{
    if (!e || !blk) {
        return fail(MFM_E_INVAL, "NULL argument");
    }
    if (!e->committed || (e->cfg.flags & MFM_F_DEVICE_ONLY)) {
        return fail(MFM_E_STATE, "fetch needs a committed engine without MFM_F_DEVICE_ONLY");
    }
    OutSlot *sp = nullptr;
    {
        std::lock_guard<std::mutex> guard(e->mu);
        if (e->fetch_seq >= e->submit_seq) {
            return MFM_E_DONE;
        }
        sp = &e->slots[e->fetch_seq % e->nslots];
        if (sp->state == OutSlot::FETCHED) {
            return fail(MFM_E_STATE, "release the previous block first");
        }
    }
    OutSlot &s = *sp;
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipEventSynchronize(s.ready)); /* outside the lock: the producer keeps submitting meanwhile */
    std::lock_guard<std::mutex> guard(e->mu);
    s.state = OutSlot::FETCHED;
    blk->first_output = s.first_output;
    blk->nr_outputs = s.nr_outputs;
    blk->stride = e->out_stride;
    blk->pcm = s.h_pcm;
    blk->iq = reinterpret_cast<const int16_t *>(s.h_iq);
    return MFM_OK;
}

int mfm_engine_release(struct mfm_engine *e)
{
    if (!e) {
        return fail(MFM_E_INVAL, "NULL engine");
    }
    std::lock_guard<std::mutex> guard(e->mu);
    if (!e->committed || e->fetch_seq >= e->submit_seq) {
        return fail(MFM_E_STATE, "nothing fetched");
    }
    OutSlot &s = e->slots[e->fetch_seq % e->nslots];
    if (s.state != OutSlot::FETCHED) {
        return fail(MFM_E_STATE, "nothing fetched");
    }
    s.state = OutSlot::FREE;
    e->fetch_seq++;
    return MFM_OK;
}

int mfm_engine_last_output_device(struct mfm_engine *e, void **d_pcm, size_t *stride, size_t *nr_outputs,
                                  void **d_iq)
{
    if (!e || !e->committed || e->last_slot < 0) {
        return fail(MFM_E_STATE, "no output yet");
    }
    const OutSlot &s = e->slots[e->last_slot];
    if (d_pcm) {
        *d_pcm = s.d_pcm;
    }
    if (stride) {
        *stride = e->out_stride;
    }
    if (nr_outputs) {
        *nr_outputs = s.nr_outputs;
    }
    if (d_iq) {
        *d_iq = s.d_iq;
    }
    return MFM_OK;
}

int mfm_engine_sync(struct mfm_engine *e)
{
    if (!e || !e->committed) {
        return fail(MFM_E_STATE, "commit first");
    }
    HIP_TRY(hipSetDevice(e->cfg.device));
    /* what has been accepted and not launched goes out first (MFM_E_BUSY: the caller fetches / releases and syncs again) */
    const int rc = mfm_engine_flush(e);
    if (rc != MFM_OK) {
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(e->s_in));
    HIP_TRY(hipStreamSynchronize(e->s_compute));
    if (e->ncs > 1u) {
        HIP_TRY(hipStreamSynchronize(e->cs[1]));
    }
    HIP_TRY(hipStreamSynchronize(e->s_out));
    return MFM_OK;
}

int mfm_engine_reset(struct mfm_engine *e)
{
    if (!e || !e->committed) {
        return fail(MFM_E_STATE, "commit first");
    }
    /* pending blocks are dropped, accepted-but-unlaunched samples with them */
    e->pend = 0;
    int rc = mfm_engine_sync(e);
    if (rc != MFM_OK) {
        return rc;
    }
    rc = write_state_fresh(e);
    if (rc != MFM_OK) {
        return rc;
    }
    for (int i = 0; i < e->nslots; i++) {
        e->slots[i].state = OutSlot::FREE;
    }
    e->fetch_seq = e->submit_seq = 0;
    e->last_slot = -1;
    e->tail = 0;
    e->hist = 0;
    e->tail_fmt = MFM_IN_CS16;
    e->in_fmt[0] = e->in_fmt[1] = e->in_fmt[2] = MFM_IN_CS16;
    e->cur_in = 0;
    e->wait_copy_stream = false;
    e->last_launch_samples = 0;
    e->last_launch_buf = -1;
    e->outputs = 0;
    e->samples_in = 0;
    return MFM_OK;
}

int mfm_engine_seek(struct mfm_engine *e, uint64_t outputs_before)
{
    int rc = mfm_engine_reset(e);
    if (rc != MFM_OK) {
        return rc;
    }
    /* the rotators stand where outputs_before steps of the recurrence leave them (filter/direct_fir.c:166-167); the
     * second-generation kernel folds the count itself, the others read the folded table position from the carried state */
    std::vector<mfm_chan_state> st(e->ngroups * MFM_CG);
    for (size_t c = 0; c < st.size(); c++) {
        st[c].carry_q = 0;
        st[c].kb = 0;
        if (c < e->chans.size()) {
            const Channel &ch = e->chans[c];
            st[c].kb = outputs_before < ch.mu ? (uint32_t)outputs_before : ch.mu + (uint32_t)((outputs_before - ch.mu) % ch.lam);
        }
    }
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipMemcpy(e->d_state[i], st.data(), st.size() * sizeof(mfm_chan_state), hipMemcpyHostToDevice));
    }
    e->outputs = outputs_before;
    return MFM_OK;
}

int mfm_engine_get_stats(struct mfm_engine *e, struct mfm_stats *st)
{
    if (!e || !st) {
        return fail(MFM_E_INVAL, "NULL argument");
    }
    memset(st, 0, sizeof(*st));
    if (e->committed && (e->cfg.flags & MFM_F_TIMING)) {
        HIP_TRY(hipSetDevice(e->cfg.device));
        int rc = fold_timing(e, true);
        if (rc != MFM_OK) {
            return rc;
        }
    }
    st->samples_in = e->samples_in;
    st->outputs = e->outputs;
    st->launches = e->launches;
    st->kernel_ms = e->kernel_ms;
    st->timed_launches = e->launch_ms_n;
    st->rot_exact_channels = e->rot_exact_channels;
    st->rot_fast_slices = e->rot_fast_slices;
    st->k_steps = e->use_mfma ? e->m_ks : 0u;
    st->tap_hi_mask = e->use_mfma ? e->m_ah_mask : 0u;
    st->taps_resident = ((e->use_mfma && !e->use_v3 && e->m_resident_taps) || (e->use_v3 && 3u == e->v_layout)) ? 1u : 0u;
    st->slice_channels = e->use_v3 ? ((3u == e->v_layout) ? 64u * e->v_rb : 64u) : e->use_mfma ? 64u : 0u;
    st->submits = e->submits;
    st->nr_channels = (uint32_t)e->chans.size();
    st->nr_taps = e->nr_taps;
    st->outputs_per_tile = e->use_v3 ? MFM_V3_OT : e->use_mfma ? e->m_ot : 64u * e->opl;
    st->lds_bytes = e->use_v3 ? e->v_lds_bytes : e->use_mfma ? e->m_lds_bytes : e->lds_bytes;
    st->kernel_variant = e->use_v3 ? 2u : e->use_mfma ? 1u : 0u;
    {
        std::lock_guard<std::mutex> guard(e->mu);
        st->pending_blocks = (uint32_t)(e->submit_seq - e->fetch_seq);
        st->pending_samples = e->pend; /* written under the lock by submit / launch */
        st->launches_8bit = e->launches_8bit;
    }
    st->grid_last = e->grid_last;
    st->tail_samples = e->tail;
    st->rot_table_entries = e->rot_entries;
    return MFM_OK;
}

size_t mfm_engine_get_launch_ms(struct mfm_engine *e, float *dst, size_t cap)
{
    if (!e || !e->committed || !(e->cfg.flags & MFM_F_TIMING)) {
        return 0;
    }
    if (hipSetDevice(e->cfg.device) != hipSuccess || fold_timing(e, true) != MFM_OK) {
        return 0;
    }
    const size_t have = (size_t)std::min<uint64_t>(e->launch_ms_n, kLaunchRing);
    const size_t n = std::min(have, cap);
    for (size_t i = 0; i < n && dst; i++) {
        dst[i] = e->launch_ms[(e->launch_ms_n - n + i) % kLaunchRing];
    }
    return n;
}

size_t mfm_engine_get_launch_cycles(struct mfm_engine *e, uint64_t *shader_ticks, uint64_t *ref_ticks, size_t cap)
{
    if (!e || !e->committed || !e->d_cyc) {
        return 0;
    }
    /* Only what HAS been launched is waited for: this is a read-only call (also on the engines of a device group, from whatever
     * thread looks at the figures) - a flush from here would launch one shard of a gathering group by itself and take the
     * shards out of step. */
    if (hipSetDevice(e->cfg.device) != hipSuccess || hipStreamSynchronize(e->s_compute) != hipSuccess ||
        (e->ncs > 1u && hipStreamSynchronize(e->cs[1]) != hipSuccess)) {
        return 0;
    }
    std::vector<unsigned long long> ring(kCycleRing * 2);
    if (hipMemcpy(ring.data(), e->d_cyc, ring.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) {
        return 0;
    }
    /* (a launch clears the slot half a ring ahead of its own: the newer half of the ring is what can be read) */
    const uint64_t have = std::min<uint64_t>(e->launches, kCycleRing / 2u);
    const size_t n = (size_t)std::min<uint64_t>(have, cap);
    for (size_t i = 0; i < n; i++) {
        const uint64_t seq = e->launches - n + i; /* 0-based index of the launch */
        const uint64_t tag = ((seq + 1u) & 0xffffffu) << 40, mask = (1ull << 40) - 1ull;
        const unsigned long long a = ring[2 * (seq % kCycleRing)], b = ring[2 * (seq % kCycleRing) + 1];
        if (shader_ticks) {
            shader_ticks[i] = (a & ~mask) == tag ? (a & mask) : 0u; /* 0: the launch left no stamp (no work, or not this kernel) */
        }
        if (ref_ticks) {
            ref_ticks[i] = (b & ~mask) == tag ? (b & mask) : 0u;
        }
    }
    return n;
}

void *mfm_engine_stream(struct mfm_engine *e)
{
    return (e && e->committed) ? e->s_last : nullptr;
}

/* ---- host twins (see include/multifm_hip.h) ---- */

static float g_atan_tbl[257];
static float2 g_atan_lut[256];
static bool g_atan_ok = false;
static void atan_tbl_once(void);

int mfm_devtest_rcp_table(int device, uint64_t *hash, uint64_t counts[4], uint64_t *sweep_bad, uint64_t *sweep_tried)
{
    if (!hash || !counts) {
        return fail(MFM_E_INVAL, "bad argument");
    }
    uint64_t t[5];
    int rc = rcp_table_read(device, t);
    if (rc != MFM_OK) {
        return rc;
    }
    *hash = t[0];
    counts[0] = t[1];
    counts[1] = t[2];
    counts[2] = t[3];
    counts[3] = t[4];
    if (sweep_bad && sweep_tried) {
        rc = div_sweep(device, sweep_bad, sweep_tried);
    }
    return rc;
}

/* the discriminator as the kernels of variant 0 / 1 / 2 compute it, on the device, for caller-supplied products */
int mfm_devtest_discriminate(int variant, const int32_t *s_re, const int32_t *s_im, size_t n, int16_t *pcm_out, int device)
{
    if (!s_re || !s_im || !pcm_out || variant < 0 || variant > 2 || n > (1u << 28)) {
        return fail(MFM_E_INVAL, "bad argument");
    }
    if (0 == n) {
        return MFM_OK;
    }
    atan_tbl_once();
    HIP_TRY(hipSetDevice(device));
    const size_t np = (n + 3u) & ~(size_t)3u;
    int *d_re = nullptr, *d_im = nullptr, *d_out = nullptr;
    float2 *d_lut = nullptr;
    int rc = MFM_OK;
    std::vector<int32_t> out(np);
    do {
        if (hipMalloc(&d_re, np * 4) != hipSuccess || hipMalloc(&d_im, np * 4) != hipSuccess || hipMalloc(&d_out, np * 4) != hipSuccess ||
            hipMalloc(&d_lut, sizeof(g_atan_lut)) != hipSuccess) {
            rc = fail(MFM_E_NOMEM, "device allocation failed");
            break;
        }
        if (hipMemset(d_re, 0, np * 4) != hipSuccess || hipMemset(d_im, 0, np * 4) != hipSuccess ||
            hipMemcpy(d_re, s_re, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_im, s_im, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_lut, g_atan_lut, sizeof(g_atan_lut), hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(MFM_E_DEVICE, "copy to the device failed");
            break;
        }
        const hipError_t err = variant == 2   ? mfm_disc_test_v3(d_re, d_im, d_out, (uint32_t)np, d_lut, nullptr)
                               : variant == 1 ? mfm_disc_test_mfma(d_re, d_im, d_out, (uint32_t)np, d_lut, nullptr)
                                              : mfm_disc_test_dot2(d_re, d_im, d_out, (uint32_t)np, d_lut, nullptr);
        if (err != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(out.data(), d_out, np * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(MFM_E_DEVICE, "discriminator test kernel failed: %s", hipGetErrorString(err));
            break;
        }
        for (size_t i = 0; i < n; i++) {
            pcm_out[i] = (int16_t)out[i];
        }
    } while (0);
    (void)hipFree(d_re);
    (void)hipFree(d_im);
    (void)hipFree(d_out);
    (void)hipFree(d_lut);
    return rc;
}

static void atan_tbl_build(void)
{
    /* multifm/fast_atan2f.c:14-81: the reference's literals are atan(i/255) printed with seven
     * significant digits; entry 256 repeats entry 255 (the index+1 read at :131). */
    for (int i = 0; i < 257; i++) {
        char txt[32];
        const int k = i < 255 ? i : 255;
        snprintf(txt, sizeof(txt), "%.6e", atan((double)k / 255.0));
        g_atan_tbl[i] = strtof(txt, nullptr);
    }
    for (int i = 0; i < 256; i++) {
        g_atan_lut[i] = make_float2(g_atan_tbl[i], g_atan_tbl[i + 1] - g_atan_tbl[i]);
    }
    /* FNV-1a over the 257 bit patterns; the constant was checked against values read back through
     * the reference's own fast_atan2f() (tests/test_oracle_atan2.py). A libm that rounds atan
     * differently at the 7th digit would trip this instead of silently changing PCM. */
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 257; i++) {
        uint32_t b;
        memcpy(&b, &g_atan_tbl[i], 4);
        for (int k = 0; k < 4; k++) {
            h ^= (b >> (8 * k)) & 0xffu;
            h *= 1099511628211ull;
        }
    }
    g_atan_ok = (h == MFM_ATAN_TABLE_FNV1A);
}

static void atan_tbl_once(void)
{
    static std::once_flag once;
    std::call_once(once, atan_tbl_build);
}

int mfm_hosttwin_atan_table_ok(void)
{
    atan_tbl_once();
    return g_atan_ok ? 1 : 0;
}

void mfm_hosttwin_atan_table(float tbl[257])
{
    atan_tbl_once();
    memcpy(tbl, g_atan_tbl, sizeof(g_atan_tbl));
}

int32_t mfm_hosttwin_discriminate(int32_t s_re, int32_t s_im)
{
    atan_tbl_once();
    return mfm_discriminate(s_re, s_im, g_atan_lut);
}

void mfm_hosttwin_discriminate_batch(const int32_t *s_re, const int32_t *s_im, size_t n, int16_t *out)
{
    atan_tbl_once();
    for (size_t i = 0; i < n; i++) {
        out[i] = (int16_t)mfm_discriminate(s_re[i], s_im[i], g_atan_lut);
    }
}

int16_t mfm_hosttwin_r14(int32_t a)
{
    return (int16_t)mfm_r14_wide(a);
}

void mfm_hosttwin_pcm_range(uint32_t first_bits, uint32_t count, int16_t *out)
{
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t b = first_bits + i;
        float m;
        memcpy(&m, &b, 4);
        out[i] = (int16_t)mfm_mag_to_pcm(m);
    }
}

} /* extern "C" */
