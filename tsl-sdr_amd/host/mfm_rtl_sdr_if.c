/*
 * mfm_rtl_sdr_if.c - RTL-SDR front end: librtlsdr's asynchronous reader feeding the receiver's sample_buf pool.
 *
 * Behaviour of the reference's multifm/rtl_sdr_if.c:
 *   - configuration (:339-441): sampleRateHz, centerFreqHz, device.deviceIndex, device.dBGainLNA (manual LNA gain, else
 *     automatic gain control), device.dbGainIF (E4000 only), device.ppmCorrection, device.iqDumpFile (created O_EXCL),
 *     sdrTestMode; the tuner is programmed in that order and the endpoint is reset before streaming;
 *   - 131072 complex samples per buffer (:46, what librtlsdr delivers per transfer by default: 16 * 32 * 512 bytes);
 *   - per transfer (:87-157): nothing when muted; the raw bytes go to the dump file; a buffer from the pool or the
 *     transfer is dropped (counted in receiver_sample_buf_alloc); every byte becomes ((int16)u8 - 127) << 7; nr_samples =
 *     bytes / 2; deliver.
 * What is different: the bytes are delivered as they came off the wire (RAW_COMPLEX_RTLSDR_UINT_8) and the engine
 * widens them on the GPU with that same expression - half the PCIe bytes, and the matrix kernel reads the byte plane
 * directly (device.gpuUnpack = false widens on the host as the reference does).  librtlsdr is bound with dlopen(), the
 * library's entry points are looked up by name, so nothing here needs its header.
 */
#include "mfm_rtl_sdr_if.h"

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>

#define MFM_MSG(sev, sys, msg, ...) MESSAGE("MULTIFM", sev, sys, msg, ##__VA_ARGS__)

#define RTL_SAMPLES_PER_BUF (16u * 32u * 512u / 2u) /* multifm/rtl_sdr_if.c:46 */
#define RTL_WIDEN_SHIFT 7                           /* multifm/rtl_sdr_if.c:45 */

/* tuner numbers of librtlsdr's enum rtlsdr_tuner (public API) */
enum { TUNER_UNKNOWN = 0, TUNER_E4000, TUNER_FC0012, TUNER_FC0013, TUNER_FC2580, TUNER_R820T, TUNER_R828D, TUNER_NR };

static const char *const tuner_names[TUNER_NR] = {
    [TUNER_UNKNOWN] = "Unknown Tuner Type", [TUNER_E4000] = "Elonics E4000",       [TUNER_FC0012] = "Fitipower FC0012",
    [TUNER_FC0013] = "Fitipower FC0013",    [TUNER_FC2580] = "Fitipower FC2580",   [TUNER_R820T] = "Rafael Micro R820T",
    [TUNER_R828D] = "Rafael Micro R828D",
};

typedef void (*rtl_async_cb_t)(unsigned char *buf, uint32_t len, void *ctx);

/* the slice of librtlsdr this front end calls, resolved once */
static struct rtl_api {
    void *lib;
    uint32_t (*get_device_count)(void);
    const char *(*get_device_name)(uint32_t);
    int (*open)(void **, uint32_t);
    int (*close)(void *);
    int (*get_tuner_type)(void *);
    int (*set_sample_rate)(void *, uint32_t);
    int (*set_center_freq)(void *, uint32_t);
    int (*set_freq_correction)(void *, int);
    int (*set_agc_mode)(void *, int);
    int (*get_tuner_gains)(void *, int *);
    int (*set_tuner_gain_mode)(void *, int);
    int (*set_tuner_gain)(void *, int);
    int (*get_tuner_gain)(void *);
    int (*set_tuner_if_gain)(void *, int, int);
    int (*set_testmode)(void *, int);
    int (*reset_buffer)(void *);
    int (*read_async)(void *, rtl_async_cb_t, void *, uint32_t, uint32_t);
    int (*cancel_async)(void *);
} rtl;

static bool _rtl_api_load(void)
{
    static const char *const sonames[] = { "librtlsdr.so.0", "librtlsdr.so", "librtlsdr.so.2" };
    static const struct {
        const char *name;
        size_t slot;
    } syms[] = {
#define RTL_SYM(f) { "rtlsdr_" #f, offsetof(struct rtl_api, f) }
        RTL_SYM(get_device_count), RTL_SYM(get_device_name),     RTL_SYM(open),           RTL_SYM(close),
        RTL_SYM(get_tuner_type),   RTL_SYM(set_sample_rate),     RTL_SYM(set_center_freq), RTL_SYM(set_freq_correction),
        RTL_SYM(set_agc_mode),     RTL_SYM(get_tuner_gains),     RTL_SYM(set_tuner_gain_mode), RTL_SYM(set_tuner_gain),
        RTL_SYM(get_tuner_gain),   RTL_SYM(set_tuner_if_gain),   RTL_SYM(set_testmode),   RTL_SYM(reset_buffer),
        RTL_SYM(read_async),       RTL_SYM(cancel_async),
#undef RTL_SYM
    };
    if (NULL != rtl.lib) {
        return true;
    }
    void *lib = NULL;
    for (size_t i = 0; i < sizeof(sonames) / sizeof(sonames[0]) && NULL == lib; i++) {
        lib = dlopen(sonames[i], RTLD_NOW | RTLD_LOCAL);
    }
    if (NULL == lib) {
        return false;
    }
    for (size_t i = 0; i < sizeof(syms) / sizeof(syms[0]); i++) {
        void *fn = dlsym(lib, syms[i].name);
        if (NULL == fn) {
            MFM_MSG(SEV_ERROR, "RTLSDR-LIBRARY", "librtlsdr lacks %s", syms[i].name);
            dlclose(lib);
            return false;
        }
        memcpy((char *)&rtl + syms[i].slot, &fn, sizeof(fn));
    }
    rtl.lib = lib;
    return true;
}

bool rtl_sdr_dump_devices(void)
{
    if (!_rtl_api_load()) {
        return false;
    }
    const uint32_t nr_devs = rtl.get_device_count();
    if (0 == nr_devs) {
        MFM_MSG(SEV_WARNING, "NO-DEVS-FOUND", "No RTL-SDR devices found.");
        return true;
    }
    MFM_MSG(SEV_INFO, "DEVS-FOUND", "Found %u RTL-SDR devices", nr_devs);
    for (uint32_t i = 0; i < nr_devs; i++) {
        fprintf(stderr, "%2u: %s\n", i, rtl.get_device_name(i));
    }
    return true;
}

struct rtl_sdr_thread {
    struct receiver rx;
    void *dev;
    int dump_fd;
    bool gpu_unpack;
    _Atomic int reader_state; /* 0 not started, 1 inside rtlsdr_read_async, 2 returned */
};

/* librtlsdr's reader thread calls this for every finished USB transfer (multifm/rtl_sdr_if.c:87-157) */
static void _rtl_on_transfer(unsigned char *bytes, uint32_t len, void *ctx)
{
    struct rtl_sdr_thread *thr = ctx;
    struct sample_buf *sbuf = NULL;

    if (thr->rx.muted) {
        return;
    }
    if (thr->dump_fd >= 0 && write(thr->dump_fd, bytes, len) < 0) {
        DIAG("Failed to write %u bytes to the RTL-SDR dump file.", len);
    }
    if (len < 2u || FAILED(receiver_sample_buf_alloc(&thr->rx, &sbuf))) {
        return; /* pool empty: dropped and counted (multifm/receiver.c:57-63) */
    }
    if (len > 2u * RTL_SAMPLES_PER_BUF) {
        len = 2u * RTL_SAMPLES_PER_BUF; /* a frame holds one default-sized transfer */
    }
    if (thr->gpu_unpack) {
        memcpy(sbuf->data_buf, bytes, len);
        sbuf->sample_type = RAW_COMPLEX_RTLSDR_UINT_8;
    } else {
        int16_t *out = (int16_t *)sbuf->data_buf;
        for (uint32_t i = 0; i < len; i++) {
            /* multifm/rtl_sdr_if.c:146-148 */
            out[i] = (int16_t)(((int16_t)bytes[i] - 127) << RTL_WIDEN_SHIFT);
        }
    }
    sbuf->nr_samples = len / 2u;
    TSL_BUG_IF_FAILED(receiver_sample_buf_deliver(&thr->rx, sbuf));
}

/* the receiver's thread: librtlsdr runs its transfer loop on it until rtlsdr_cancel_async() (multifm/rtl_sdr_if.c:159-177) */
static aresult_t _rtl_receive(struct receiver *rx)
{
    struct rtl_sdr_thread *thr = BL_CONTAINER_OF(rx, struct rtl_sdr_thread, rx);
    atomic_store(&thr->reader_state, 1);
    const int rc = rtl.read_async(thr->dev, _rtl_on_transfer, thr, 0, 0);
    if (0 != rc) {
        MFM_MSG(SEV_WARNING, "UNCLEAN-TERM", "The RTL-SDR Async Reader terminated with an error (%d).", rc);
    }
    MFM_MSG(SEV_INFO, "RECEIVER-THREAD-TERMINATED", "Terminating RTL-SDR Receiver thread...");
    atomic_store(&thr->reader_state, 2); /* the last access to thr: _rtl_cleanup() may free it now */
    return A_OK;
}

static void _rtl_release(struct rtl_sdr_thread *thr)
{
    if (NULL != thr->dev) {
        rtl.close(thr->dev);
        thr->dev = NULL;
    }
    if (thr->dump_fd >= 0) {
        close(thr->dump_fd);
        thr->dump_fd = -1;
    }
}

/* The reference cancels, closes and frees in one go while the reader may still be winding down (multifm/rtl_sdr_if.c:59-82);
 * here the device is closed and the context freed once the reader has really returned. */
static aresult_t _rtl_cleanup(struct receiver *rx)
{
    struct rtl_sdr_thread *thr = BL_CONTAINER_OF(rx, struct rtl_sdr_thread, rx);
    if (NULL != thr->dev && 1 == atomic_load(&thr->reader_state)) {
        TSL_BUG_ON(0 != rtl.cancel_async(thr->dev));
        for (int ms = 0; ms < 5000 && 2 != atomic_load(&thr->reader_state); ms++) {
            usleep(1000);
        }
    }
    if (1 == atomic_load(&thr->reader_state)) {
        MFM_MSG(SEV_WARNING, "READER-STUCK", "The RTL-SDR reader did not return after being cancelled; leaving its context alone.");
        return A_OK;
    }
    _rtl_release(thr);
    TFREE(thr);
    return A_OK;
}

/* ---- tuner set-up ---- */

/* Distribute an IF gain (tenths of a dB) over the E4000's six stages: every stage is raised by its step while the
 * remaining distance is larger than that step and the stage stays within its range, round after round until a round
 * changes nothing (multifm/rtl_sdr_if.c:179-224). */
static void _rtl_e4000_if_gain(void *dev, int want_tenths)
{
    static const int step[6] = { 90, 30, 30, 10, 30, 30 }, top[6] = { 60, 90, 90, 20, 150, 150 };
    int stage[6] = { -30, 0, 0, 0, 30, 30 }, total = 30;

    for (bool moved = true; moved;) {
        moved = false;
        for (int i = 0; i < 6; i++) {
            if (stage[i] + step[i] <= top[i] && want_tenths - total > step[i]) {
                stage[i] += step[i];
                total += step[i];
                moved = true;
            }
        }
    }
    for (int i = 0; i < 6; i++) {
        if (0 != rtl.set_tuner_if_gain(dev, i + 1, stage[i])) {
            MFM_MSG(SEV_WARNING, "FAILED-SETTING-GAIN", "Failed to set IF gain stage %d to value %d", i + 1, stage[i]);
        }
    }
}

static void _free_gain_list(int **p)
{
    free_memory((void **)p);
}

/* manual LNA gain: the first supported gain that reaches the request, else the largest (multifm/rtl_sdr_if.c:226-300) */
static aresult_t _rtl_lna_gain(void *dev, int want_tenths)
{
    int *supported CAL_CLEANUP(_free_gain_list) = NULL;

    if (0 != rtl.set_agc_mode(dev, 0)) {
        MFM_MSG(SEV_WARNING, "CANT-SET-AGC", "Failed to disable AGC.");
    }
    const int nr = rtl.get_tuner_gains(dev, NULL);
    if (nr <= 0) {
        MFM_MSG(SEV_ERROR, "CANT-GET-GAINS", "Unable to get list of supported gains.");
        return A_E_INVAL;
    }
    TSL_BUG_IF_FAILED(TCALLOC(&supported, sizeof(int), (size_t)nr));
    if (rtl.get_tuner_gains(dev, supported) <= 0) {
        MFM_MSG(SEV_ERROR, "CANT-GET-GAIN-LIST", "Unable to get list of gains.");
        return A_E_INVAL;
    }
    int pick = 0;
    while (pick + 1 < nr && supported[pick] < want_tenths) {
        pick++;
    }
    MFM_MSG(SEV_INFO, "RECV-GAIN", "Setting receive gain to %d.%d dB", supported[pick] / 10, supported[pick] % 10);
    if (rtl.set_tuner_gain_mode(dev, 1) < 0) {
        MFM_MSG(SEV_ERROR, "FAILED-TO-ENABLE-MANUAL-GAIN", "Unable to enable manual gain, aborting.");
        return A_E_INVAL;
    }
    if (0 != rtl.set_tuner_gain(dev, supported[pick])) {
        MFM_MSG(SEV_ERROR, "FAILED-SET-GAIN", "Failed to set requested gain.");
        return A_E_INVAL;
    }
    return A_OK;
}

/* what the configuration asks of the dongle */
struct rtl_settings {
    int dev_idx, sample_rate, center_freq, ppm;
    bool have_lna, have_if, test_mode, gpu_unpack;
    double lna_db, if_db;
    const char *dump_file;
};

static aresult_t _rtl_settings_read(struct config *cfg, struct rtl_settings *st)
{
    struct config device = CONFIG_INIT_EMPTY;
    aresult_t ret = A_OK;

    memset(st, 0, sizeof(*st));
    st->gpu_unpack = true;
    TSL_BUG_IF_FAILED(config_get(cfg, &device, "device"));
    if (FAILED(ret = config_get_integer(cfg, &st->sample_rate, "sampleRateHz"))) {
        MFM_MSG(SEV_INFO, "NO-SAMPLE-RATE", "Need to specify a sample rate, in Hertz.");
        return ret;
    }
    if (FAILED(ret = config_get_integer(cfg, &st->center_freq, "centerFreqHz"))) {
        MFM_MSG(SEV_INFO, "NO-CENTER-FREQ", "You forgot to specify a center frequency, in Hz.");
        return ret;
    }
    if (FAILED(ret = config_get_integer(&device, &st->dev_idx, "deviceIndex"))) {
        MFM_MSG(SEV_ERROR, "NO-DEV-SPEC", "Need to specify a 'deviceIndex' entry in configuration");
        return ret;
    }
    st->have_lna = !FAILED(config_get_float(&device, &st->lna_db, "dBGainLNA"));
    st->have_if = !FAILED(config_get_float(&device, &st->if_db, "dbGainIF"));
    if (FAILED(config_get_integer(&device, &st->ppm, "ppmCorrection"))) {
        st->ppm = 0;
    }
    if (FAILED(config_get_string(&device, &st->dump_file, "iqDumpFile"))) {
        st->dump_file = NULL;
    }
    if (FAILED(config_get_boolean(cfg, &st->test_mode, "sdrTestMode"))) {
        st->test_mode = false;
    }
    (void)config_get_boolean(&device, &st->gpu_unpack, "gpuUnpack");
    return A_OK;
}

/* programs the opened dongle; the order is the reference's (multifm/rtl_sdr_if.c:366-452) */
static aresult_t _rtl_program(void *dev, const struct rtl_settings *st, int *dump_fd)
{
    const int tuner = rtl.get_tuner_type(dev);

    MFM_MSG(SEV_INFO, "DEV-IDX-OPEN", "Successfully opened device at index %d", st->dev_idx);
    MFM_MSG(SEV_INFO, "DEV-IDX-OPEN", "Device: %s Tuner: %s", rtl.get_device_name((uint32_t)st->dev_idx),
            (tuner >= 0 && tuner < TUNER_NR) ? tuner_names[tuner] : tuner_names[TUNER_UNKNOWN]);
    if (TUNER_E4000 != tuner && TUNER_R820T != tuner) {
        MFM_MSG(SEV_WARNING, "DEV-UNTESTED", "This tuner type is not tested, so the performance could be poor");
    }
    MFM_MSG(SEV_INFO, "SAMPLE-RATE", "Setting sample rate to %u Hz", (unsigned)st->sample_rate);
    if (0 != rtl.set_sample_rate(dev, (uint32_t)st->sample_rate)) {
        MFM_MSG(SEV_ERROR, "BAD-SAMPLE-RATE", "Failed to set sample rate, aborting.");
        return A_E_INVAL;
    }
    MFM_MSG(SEV_INFO, "CENTER-FREQ", "Setting Center Frequency to %u Hz", (unsigned)st->center_freq);
    if (0 != rtl.set_center_freq(dev, (uint32_t)st->center_freq)) {
        MFM_MSG(SEV_ERROR, "BAD-CENTER-FREQ", "Failed to set center frequency, aborting.");
        return A_E_INVAL;
    }
    if (st->have_lna) {
        TSL_BUG_IF_FAILED(_rtl_lna_gain(dev, (int)(st->lna_db * 10)));
    } else {
        MFM_MSG(SEV_INFO, "AUTO-GAIN-CONTROL", "Enabling automatic gain control.");
        TSL_BUG_ON(0 != rtl.set_tuner_gain_mode(dev, 0));
    }
    if (TUNER_E4000 == tuner && st->have_if) {
        _rtl_e4000_if_gain(dev, (int)(st->if_db * 10));
    }
    DIAG("LNA gain set to: %f", (double)rtl.get_tuner_gain(dev) / 10.0);
    if (0 != st->ppm) {
        const int rc = rtl.set_freq_correction(dev, st->ppm);
        if (0 != rc) {
            MFM_MSG(SEV_ERROR, "CANT-SET-FREQ-CORR", "Failed to set frequency correction to %d PPM (error: %d)", st->ppm, rc);
            return A_E_INVAL;
        }
    }
    MFM_MSG(SEV_INFO, "FREQ-CORR", "Set frequency correction to %d PPM", st->ppm);
    if (NULL != st->dump_file) {
        *dump_fd = open(st->dump_file, O_RDWR | O_CREAT | O_EXCL, 0666);
        if (*dump_fd < 0) {
            const int errnum = errno;
            MFM_MSG(SEV_INFO, "DUMP-FILE-FAIL", "Failed to create dump file '%s' - reason: '%s' (%d)", st->dump_file,
                    strerror(errnum), errnum);
            return A_E_INVAL;
        }
        MFM_MSG(SEV_INFO, "DUMP-TO-FILE", "Dumping raw I-Q samples as 8-bit interleaved to '%s'", st->dump_file);
    }
    TSL_BUG_ON(0 != rtl.reset_buffer(dev));
    if (st->test_mode) {
        MFM_MSG(SEV_INFO, "TEST-MODE", "Enabling RTL-SDR test mode");
        if (0 != rtl.set_testmode(dev, 1)) {
            MFM_MSG(SEV_ERROR, "CANT-SET-TEST-MODE", "Failed to enable test mode, aborting.");
            return A_E_INVAL;
        }
    }
    return A_OK;
}

aresult_t rtl_sdr_worker_thread_new(struct receiver **pthr, struct config *cfg)
{
    struct rtl_sdr_thread *thr = NULL;
    struct rtl_settings st;
    aresult_t ret = A_OK;

    TSL_ASSERT_ARG(NULL != cfg);
    TSL_ASSERT_ARG(NULL != pthr);
    *pthr = NULL;

    if (!_rtl_api_load()) {
        MFM_MSG(SEV_FATAL, "RTLSDR-NOT-SUPPORTED", "RTL-SDR devices are not supported by this build.");
        return A_E_NOTFOUND;
    }
    if (FAILED(ret = _rtl_settings_read(cfg, &st))) {
        return ret;
    }
    if (FAILED(ret = TZAALLOC(thr, SYS_CACHE_LINE_LENGTH))) {
        return ret;
    }
    thr->dump_fd = -1;
    thr->gpu_unpack = st.gpu_unpack;
    if (0 != rtl.open(&thr->dev, (uint32_t)st.dev_idx) || NULL == thr->dev) {
        MFM_MSG(SEV_ERROR, "BAD-DEV-SPEC", "Could not open device index %d.", st.dev_idx);
        thr->dev = NULL;
        ret = A_E_INVAL;
    }
    if (!FAILED(ret)) {
        ret = _rtl_program(thr->dev, &st, &thr->dump_fd);
    }
    if (!FAILED(ret)) {
        ret = receiver_init(&thr->rx, cfg, _rtl_receive, _rtl_cleanup, RTL_SAMPLES_PER_BUF);
    }
    if (FAILED(ret)) {
        _rtl_release(thr);
        TFREE(thr);
        return ret;
    }
    *pthr = &thr->rx;
    return A_OK;
}
