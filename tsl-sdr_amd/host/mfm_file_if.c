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

enum file_worker_sample_format {
    FILE_WORKER_SAMPLE_FORMAT_UNKNOWN = 0,
    FILE_WORKER_SAMPLE_FORMAT_S8,
    FILE_WORKER_SAMPLE_FORMAT_U8,
    FILE_WORKER_SAMPLE_FORMAT_S16,
};

struct file_worker_thread {
    struct receiver rcvr;
    int fd;
    enum file_worker_sample_format sample_format;
    void *bounce_buf;
    size_t bounce_buf_bytes;
    bool gpu_unpack; /* device stanza "gpuUnpack" (default true): widen 8-bit formats on the GPU */
};

/* read until `want` bytes or end of file (a pipe may return short reads) */
static aresult_t _file_read_full(int fd, void *dst, size_t want, size_t *got)
{
    size_t n = 0;
    while (n < want) {
        ssize_t r = read(fd, (uint8_t *)dst + n, want - n);
        if (r < 0) {
            if (errno == EINTR) {
                continue;
            }
            FL_MSG(SEV_FATAL, "FILE-READ-ERROR", "Failed to read data from file, reason: %s (%d)", strerror(errno), errno);
            return A_E_INVAL;
        }
        if (0 == r) {
            break;
        }
        n += (size_t)r;
    }
    *got = n;
    return A_OK;
}

static aresult_t _file_fill(struct file_worker_thread *thr, struct sample_buf *sbuf)
{
    size_t nr_read = 0;
    int16_t *out = (int16_t *)sbuf->data_buf;

    if (thr->sample_format == FILE_WORKER_SAMPLE_FORMAT_S16) {
        if (FAILED(_file_read_full(thr->fd, sbuf->data_buf, SAMPLES_PER_BUF * 2 * sizeof(int16_t), &nr_read))) {
            return A_E_INVAL;
        }
        sbuf->nr_samples = (uint32_t)(nr_read / (2 * sizeof(int16_t)));
        return A_OK;
    }

    if (thr->gpu_unpack) {
        /* hand the bytes over as they are; the engine widens them on the device (half the PCIe traffic) */
        if (FAILED(_file_read_full(thr->fd, sbuf->data_buf, thr->bounce_buf_bytes, &nr_read))) {
            return A_E_INVAL;
        }
        sbuf->sample_type = thr->sample_format == FILE_WORKER_SAMPLE_FORMAT_S8 ? RAW_COMPLEX_INT_8 : RAW_COMPLEX_FILE_UINT_8;
        sbuf->nr_samples = (uint32_t)(nr_read / 2);
        return A_OK;
    }
    if (FAILED(_file_read_full(thr->fd, thr->bounce_buf, thr->bounce_buf_bytes, &nr_read))) {
        return A_E_INVAL;
    }
    const int8_t *in = thr->bounce_buf; /* signed for both 8-bit formats, as in the reference */
    if (thr->sample_format == FILE_WORKER_SAMPLE_FORMAT_S8) {
        for (size_t i = 0; i < nr_read; i++) {
            out[i] = in[i];
        }
    } else {
        /* the reference's remainder loop stores the last nr_read % 4 values without the subtraction (:146-150) */
        const size_t body = nr_read - nr_read % 4;
        for (size_t i = 0; i < body; i++) {
            out[i] = (int16_t)((int16_t)in[i] - 127);
        }
        for (size_t i = body; i < nr_read; i++) {
            out[i] = in[i];
        }
    }
    sbuf->nr_samples = (uint32_t)(nr_read / 2);
    return A_OK;
}

static aresult_t _file_worker_thread_work(struct receiver *rx)
{
    struct file_worker_thread *thr = BL_CONTAINER_OF(rx, struct file_worker_thread, rcvr);

    while (receiver_thread_running(rx)) {
        struct sample_buf *sbuf = NULL;
        if (FAILED(receiver_sample_buf_alloc(rx, &sbuf))) {
            usleep(1000);
            continue;
        }
        if (FAILED(_file_fill(thr, sbuf)) || 0 == sbuf->nr_samples) {
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

static aresult_t _file_worker_thread_cleanup(struct receiver *rx)
{
    struct file_worker_thread *fwt = BL_CONTAINER_OF(rx, struct file_worker_thread, rcvr);
    if (fwt->fd >= 0) {
        close(fwt->fd);
        fwt->fd = -1;
    }
    if (NULL != fwt->bounce_buf) {
        TFREE(fwt->bounce_buf);
    }
    return A_OK;
}

aresult_t file_worker_thread_new(struct receiver **pthr, struct config *cfg)
{
    aresult_t ret = A_OK;
    struct file_worker_thread *thr = NULL;
    int fd = -1;
    const char *filename = NULL, *format = NULL;
    struct config devcfg = CONFIG_INIT_EMPTY;
    enum file_worker_sample_format sample_format = FILE_WORKER_SAMPLE_FORMAT_UNKNOWN;

    TSL_ASSERT_ARG(NULL != pthr);
    TSL_ASSERT_ARG(NULL != cfg);
    *pthr = NULL;

    if (FAILED(ret = config_get(cfg, &devcfg, "device"))) {
        FL_MSG(SEV_FATAL, "MISSING-DEVICE-STANZA", "Missing 'device' stanza of configuration, aborting.");
        goto done;
    }
    if (FAILED(ret = config_get_string(&devcfg, &filename, "filename"))) {
        FL_MSG(SEV_FATAL, "CONFIG-NO-FILE", "Need to specify a filename in the device config, aborting.");
        goto done;
    }
    if (FAILED(ret = config_get_string(&devcfg, &format, "fileFormat"))) {
        FL_MSG(SEV_FATAL, "CONFIG-NO-FORMAT", "Need to specify a fileFormat (cs16, cs8, cu8), aborting.");
        goto done;
    }
    if (!strncmp(format, "cs16", 4)) {
        sample_format = FILE_WORKER_SAMPLE_FORMAT_S16;
    } else if (!strncmp(format, "cs8", 3)) {
        sample_format = FILE_WORKER_SAMPLE_FORMAT_S8;
    } else if (!strncmp(format, "cu8", 3)) {
        sample_format = FILE_WORKER_SAMPLE_FORMAT_U8;
    } else {
        FL_MSG(SEV_FATAL, "UNSUPPORTED-FILE-FORMAT", "File format [%s] is not supported, aborting.", format);
        ret = A_E_INVAL;
        goto done;
    }
    FL_MSG(SEV_INFO, "CREATING-FILE-SOURCE", "Sourcing samples in format %s from file [%s]", format, filename);

    if (0 > (fd = open(filename, O_RDONLY))) {
        FL_MSG(SEV_FATAL, "BAD-FILE", "Unable to open file [%s], aborting. Reason: %s (%d)", filename, strerror(errno), errno);
        ret = A_E_INVAL;
        goto done;
    }
    if (FAILED(ret = TZAALLOC(thr, SYS_CACHE_LINE_LENGTH))) {
        goto done;
    }
    thr->fd = fd;
    thr->sample_format = sample_format;
    thr->gpu_unpack = true;
    (void)config_get_boolean(&devcfg, &thr->gpu_unpack, "gpuUnpack");
    if (sample_format != FILE_WORKER_SAMPLE_FORMAT_S16) {
        thr->bounce_buf_bytes = SAMPLES_PER_BUF * 2 * sizeof(int8_t);
        if (FAILED(ret = TACALLOC(&thr->bounce_buf, SAMPLES_PER_BUF, 2 * sizeof(int8_t), SYS_CACHE_LINE_LENGTH))) {
            goto done;
        }
    }
    if (FAILED(ret = receiver_init(&thr->rcvr, cfg, _file_worker_thread_work, _file_worker_thread_cleanup, SAMPLES_PER_BUF))) {
        goto done;
    }
    *pthr = &thr->rcvr;

done:
    if (FAILED(ret)) {
        if (NULL != thr) {
            if (NULL != thr->bounce_buf) {
                TFREE(thr->bounce_buf);
            }
            TFREE(thr);
        }
        if (fd != -1) {
            close(fd);
        }
    }
    return ret;
}
