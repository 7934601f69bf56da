/*
 * mfm_file_if.c - the file front end: reads a capture into sample_bufs and delivers them.
 *
 * Behaviour of the reference's multifm/file_if.c: device stanza {type:"file", filename, fileFormat in
 * cs16|cs8|cu8}; 4096 complex samples per buffer (:18); cs16 is read straight into the buffer (:46-64);
 * cs8 / cu8 go through a bounce buffer and are widened to int16 (:66-157).  Two quirks of the reference
 * are kept because they are behaviour: cu8 treats the bytes as SIGNED before subtracting 127
 * (:122,:140) and skips the subtraction for the last sample of an odd-sized final read (:146-150), and no pacing is applied (time_per_buf_ns is never set, :197).  One is not: at end of file
 * the reference delivers a 0-sample buffer and aborts on TSL_BUG_ON (receiver.c:84); here the front end
 * marks the input done and stops.
 */
#include "mfm_file_if.h"

#include <errno.h>
#include <fcntl.h>
#include <unistd.h>

#define SAMPLES_PER_BUF 4096 /* multifm/file_if.c:18 */
#define FL_MSG(sev, sys, msg, ...) MESSAGE("FILEIF", sev, sys, msg, ##__VA_ARGS__)

struct file_worker_thread;

/* How one capture format turns a read of raw bytes into what a sample_buf carries.  `widen` is the host-side conversion
 * (NULL: the bytes are the samples); `raw_type` is the tag under which the engine takes the bytes as they are and widens
 * them on the device ("gpuUnpack", default on: half the PCIe traffic). */
struct file_format {
    const char *name;            /* "fileFormat" value; matched as a prefix, like the reference's strncmp */
    size_t bytes_per_sample;     /* one complex sample on disk */
    enum sample_type raw_type;   /* for 8-bit formats: the sample_buf tag of the unwidened bytes */
    void (*widen)(const int8_t *in, size_t nr_values, int16_t *out);
};

static void _widen_cs8(const int8_t *in, size_t nr_values, int16_t *out)
{
    for (size_t i = 0; i < nr_values; i++) {
        out[i] = in[i]; /* multifm/file_if.c:91-103: sign extension */
    }
}

static void _widen_cu8(const int8_t *in, size_t nr_values, int16_t *out)
{
    /* multifm/file_if.c:122,139-151: the bytes are read as SIGNED, 127 is subtracted, and the remainder loop behind the
     * four-at-a-time body stores the bare cast - both are behaviour, so both are kept */
    const size_t body = nr_values & ~(size_t)3;
    for (size_t i = 0; i < body; i++) {
        out[i] = (int16_t)((int16_t)in[i] - 127);
    }
    for (size_t i = body; i < nr_values; i++) {
        out[i] = in[i];
    }
}

static const struct file_format file_formats[] = {
    { "cs16", 2 * sizeof(int16_t), COMPLEX_INT_16, NULL },
    { "cs8", 2 * sizeof(int8_t), RAW_COMPLEX_INT_8, _widen_cs8 },
    { "cu8", 2 * sizeof(int8_t), RAW_COMPLEX_FILE_UINT_8, _widen_cu8 },
};

static const struct file_format *_file_format_lookup(const char *name)
{
    for (size_t i = 0; i < sizeof(file_formats) / sizeof(file_formats[0]); i++) {
        if (0 == strncmp(name, file_formats[i].name, strlen(file_formats[i].name))) {
            return &file_formats[i];
        }
    }
    return NULL;
}

struct file_worker_thread {
    struct receiver rcvr;
    int fd;
    const struct file_format *fmt;
    int8_t *bounce;   /* one read of an 8-bit format that is widened on the host; NULL otherwise */
    bool gpu_unpack;  /* device stanza "gpuUnpack" (default true): widen 8-bit formats on the GPU */
};

/* read until `want` bytes or end of file (a pipe may return short reads); -1 on a read error */
static ssize_t _file_read_full(int fd, void *dst, size_t want)
{
    size_t n = 0;
    while (n < want) {
        const ssize_t r = read(fd, (uint8_t *)dst + n, want - n);
        if (r < 0 && EINTR == errno) {
            continue;
        }
        if (r < 0) {
            FL_MSG(SEV_FATAL, "FILE-READ-ERROR", "Failed to read data from file, reason: %s (%d)", strerror(errno), errno);
            return -1;
        }
        if (0 == r) {
            break;
        }
        n += (size_t)r;
    }
    return (ssize_t)n;
}

/* one buffer's worth of the capture into sbuf; false at end of input or on a read error */
static bool _file_fill(struct file_worker_thread *thr, struct sample_buf *sbuf)
{
    const struct file_format *f = thr->fmt;
    const size_t want = SAMPLES_PER_BUF * f->bytes_per_sample;
    const bool on_host = NULL != f->widen && !thr->gpu_unpack;
    const ssize_t got = _file_read_full(thr->fd, on_host ? (void *)thr->bounce : (void *)sbuf->data_buf, want);
    if (got <= 0) {
        return false;
    }
    if (on_host) {
        f->widen(thr->bounce, (size_t)got, (int16_t *)sbuf->data_buf);
    } else if (NULL != f->widen) {
        sbuf->sample_type = f->raw_type; /* the engine widens on the device, exactly as `widen` would */
    }
    sbuf->nr_samples = (uint32_t)((size_t)got / f->bytes_per_sample);
    return 0 != sbuf->nr_samples;
}

static aresult_t _file_worker_thread_work(struct receiver *rx)
{
    struct file_worker_thread *thr = BL_CONTAINER_OF(rx, struct file_worker_thread, rcvr);

    while (receiver_thread_running(rx)) {
        struct sample_buf *sbuf = NULL;
        if (FAILED(receiver_sample_buf_alloc(rx, &sbuf))) {
            usleep(1000); /* the pool is empty (dropped and counted there); a file has no clock of its own to wait on */
            continue;
        }
        if (!_file_fill(thr, sbuf)) {
            /* end of input: give the buffer back untouched and stop */
            sbuf->refcount = 1;
            TSL_BUG_IF_FAILED(sample_buf_decref(sbuf));
            break;
        }
        TSL_BUG_IF_FAILED(receiver_sample_buf_deliver(rx, sbuf));
    }
    receiver_mark_input_done(rx);
    return A_OK;
}

static void _file_worker_release(struct file_worker_thread *thr)
{
    if (thr->fd >= 0) {
        close(thr->fd);
        thr->fd = -1;
    }
    if (NULL != thr->bounce) {
        TFREE(thr->bounce);
    }
}

static aresult_t _file_worker_thread_cleanup(struct receiver *rx)
{
    _file_worker_release(BL_CONTAINER_OF(rx, struct file_worker_thread, rcvr));
    return A_OK;
}

/* what the "device" stanza says: {type: "file", filename, fileFormat in cs16 | cs8 | cu8 [, gpuUnpack]} */
struct file_source_desc {
    const char *filename;
    const struct file_format *fmt;
    bool gpu_unpack;
};

static aresult_t _file_source_parse(struct config *cfg, struct file_source_desc *d)
{
    struct config dev = CONFIG_INIT_EMPTY;
    const char *format = NULL;

    if (FAILED(config_get(cfg, &dev, "device"))) {
        FL_MSG(SEV_FATAL, "MISSING-DEVICE-STANZA", "Missing 'device' stanza of configuration, aborting.");
        return A_E_INVAL;
    }
    if (FAILED(config_get_string(&dev, &d->filename, "filename"))) {
        FL_MSG(SEV_FATAL, "CONFIG-NO-FILE", "Need to specify a filename in the device config, aborting.");
        return A_E_INVAL;
    }
    if (FAILED(config_get_string(&dev, &format, "fileFormat"))) {
        FL_MSG(SEV_FATAL, "CONFIG-NO-FORMAT", "Need to specify a fileFormat (cs16, cs8, cu8), aborting.");
        return A_E_INVAL;
    }
    if (NULL == (d->fmt = _file_format_lookup(format))) {
        FL_MSG(SEV_FATAL, "UNSUPPORTED-FILE-FORMAT", "File format [%s] is not supported, aborting.", format);
        return A_E_INVAL;
    }
    d->gpu_unpack = true;
    (void)config_get_boolean(&dev, &d->gpu_unpack, "gpuUnpack");
    return A_OK;
}

aresult_t file_worker_thread_new(struct receiver **pthr, struct config *cfg)
{
    struct file_source_desc desc = { NULL, NULL, true };
    struct file_worker_thread *thr = NULL;

    TSL_ASSERT_ARG(NULL != pthr);
    TSL_ASSERT_ARG(NULL != cfg);
    *pthr = NULL;

    aresult_t ret = _file_source_parse(cfg, &desc);
    if (FAILED(ret)) {
        return ret;
    }
    FL_MSG(SEV_INFO, "CREATING-FILE-SOURCE", "Sourcing samples in format %s from file [%s]", desc.fmt->name, desc.filename);

    if (FAILED(ret = TZAALLOC(thr, SYS_CACHE_LINE_LENGTH))) {
        return ret;
    }
    thr->fmt = desc.fmt;
    thr->gpu_unpack = desc.gpu_unpack;
    thr->fd = open(desc.filename, O_RDONLY);
    if (thr->fd < 0) {
        FL_MSG(SEV_FATAL, "BAD-FILE", "Unable to open file [%s], aborting. Reason: %s (%d)", desc.filename, strerror(errno), errno);
        ret = A_E_INVAL;
    } else if (NULL != desc.fmt->widen && !desc.gpu_unpack) {
        ret = TACALLOC(&thr->bounce, SAMPLES_PER_BUF, desc.fmt->bytes_per_sample, SYS_CACHE_LINE_LENGTH);
    }
    if (!FAILED(ret)) {
        ret = receiver_init(&thr->rcvr, cfg, _file_worker_thread_work, _file_worker_thread_cleanup, SAMPLES_PER_BUF);
    }
    if (FAILED(ret)) {
        _file_worker_release(thr);
        TFREE(thr);
        return ret;
    }
    *pthr = &thr->rcvr;
    return A_OK;
}
