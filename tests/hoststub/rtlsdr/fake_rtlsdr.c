/*
 * fake_rtlsdr.c - a test double of librtlsdr: one "dongle" whose samples come from a file.
 *
 *   FAKE_RTLSDR_FILE   raw 8-bit unsigned I,Q bytes, handed to the reader callback in transfers of buf_len bytes
 *                      (0 = librtlsdr's default, 16 * 32 * 512); at end of file the reader keeps the "USB" open and idles
 *                      until rtlsdr_cancel_async(), as a real dongle would - unless FAKE_RTLSDR_EOF_RETURNS is set
 *   FAKE_RTLSDR_LOG    every control call is appended to this file as one line ("set_sample_rate 1200000"), so a test
 *                      can check what the front end programmed and in which order
 *   FAKE_RTLSDR_TUNER  number of the tuner type reported (default 5 = R820T)
 *   FAKE_RTLSDR_PACE_US  microseconds to sleep between transfers (default 0)
 * Built by the tests into a directory of its own as librtlsdr.so.0 and found through LD_LIBRARY_PATH.
 */
#include "rtl-sdr.h"

#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

struct rtlsdr_dev {
    FILE *src;
    _Atomic int cancel;
    int gain, tuner;
};

static void note(const char *fmt, long a, long b)
{
    const char *path = getenv("FAKE_RTLSDR_LOG");
    if (NULL == path) {
        return;
    }
    FILE *f = fopen(path, "a");
    if (NULL != f) {
        fprintf(f, fmt, a, b);
        fputc('\n', f);
        fclose(f);
    }
}

uint32_t rtlsdr_get_device_count(void)
{
    return NULL != getenv("FAKE_RTLSDR_FILE") ? 1u : 0u;
}

const char *rtlsdr_get_device_name(uint32_t index)
{
    return 0 == index ? "Fake RTL2838 (file backed)" : "";
}

int rtlsdr_open(rtlsdr_dev_t **dev, uint32_t index)
{
    const char *path = getenv("FAKE_RTLSDR_FILE");
    const char *tuner = getenv("FAKE_RTLSDR_TUNER");
    if (0 != index || NULL == path) {
        return -1;
    }
    rtlsdr_dev_t *d = calloc(1, sizeof(*d));
    if (NULL == d || NULL == (d->src = fopen(path, "rb"))) {
        free(d);
        return -2;
    }
    d->tuner = NULL != tuner ? atoi(tuner) : RTLSDR_TUNER_R820T;
    note("open %ld", (long)index, 0);
    *dev = d;
    return 0;
}

int rtlsdr_close(rtlsdr_dev_t *dev)
{
    if (NULL == dev) {
        return -1;
    }
    note("close", 0, 0);
    fclose(dev->src);
    free(dev);
    return 0;
}

int rtlsdr_set_center_freq(rtlsdr_dev_t *dev, uint32_t freq)
{
    (void)dev;
    note("set_center_freq %ld", (long)freq, 0);
    return 0;
}

int rtlsdr_set_freq_correction(rtlsdr_dev_t *dev, int ppm)
{
    (void)dev;
    note("set_freq_correction %ld", ppm, 0);
    return 0;
}

enum rtlsdr_tuner rtlsdr_get_tuner_type(rtlsdr_dev_t *dev)
{
    return (enum rtlsdr_tuner)dev->tuner;
}

static const int fake_gains[] = { 0, 9, 14, 27, 37, 77, 87, 125, 144, 157, 166, 197, 207, 229, 254, 280, 297, 328, 338, 364,
                                  372, 386, 402, 421, 434, 439, 445, 480, 496 };

int rtlsdr_get_tuner_gains(rtlsdr_dev_t *dev, int *gains)
{
    (void)dev;
    if (NULL != gains) {
        memcpy(gains, fake_gains, sizeof(fake_gains));
    }
    return (int)(sizeof(fake_gains) / sizeof(fake_gains[0]));
}

int rtlsdr_set_tuner_gain(rtlsdr_dev_t *dev, int gain)
{
    dev->gain = gain;
    note("set_tuner_gain %ld", gain, 0);
    return 0;
}

int rtlsdr_get_tuner_gain(rtlsdr_dev_t *dev)
{
    return dev->gain;
}

int rtlsdr_set_tuner_if_gain(rtlsdr_dev_t *dev, int stage, int gain)
{
    (void)dev;
    note("set_tuner_if_gain %ld %ld", stage, gain);
    return 0;
}

int rtlsdr_set_tuner_gain_mode(rtlsdr_dev_t *dev, int manual)
{
    (void)dev;
    note("set_tuner_gain_mode %ld", manual, 0);
    return 0;
}

int rtlsdr_set_sample_rate(rtlsdr_dev_t *dev, uint32_t rate)
{
    (void)dev;
    note("set_sample_rate %ld", (long)rate, 0);
    return 0;
}

int rtlsdr_set_testmode(rtlsdr_dev_t *dev, int on)
{
    (void)dev;
    note("set_testmode %ld", on, 0);
    return 0;
}

int rtlsdr_set_agc_mode(rtlsdr_dev_t *dev, int on)
{
    (void)dev;
    note("set_agc_mode %ld", on, 0);
    return 0;
}

int rtlsdr_reset_buffer(rtlsdr_dev_t *dev)
{
    (void)dev;
    note("reset_buffer", 0, 0);
    return 0;
}

int rtlsdr_read_async(rtlsdr_dev_t *dev, rtlsdr_read_async_cb_t cb, void *ctx, uint32_t buf_num, uint32_t buf_len)
{
    const char *pace = getenv("FAKE_RTLSDR_PACE_US");
    const unsigned pace_us = NULL != pace ? (unsigned)atoi(pace) : 0u;
    (void)buf_num;
    if (0 == buf_len) {
        buf_len = 16u * 32u * 512u;
    }
    unsigned char *xfer = malloc(buf_len);
    if (NULL == xfer) {
        return -1;
    }
    note("read_async %ld", (long)buf_len, 0);
    while (!atomic_load(&dev->cancel)) {
        const size_t got = fread(xfer, 1, buf_len, dev->src);
        if (got >= 2) {
            cb(xfer, (uint32_t)(got & ~(size_t)1), ctx);
        }
        if (got < buf_len) {
            if (NULL != getenv("FAKE_RTLSDR_EOF_RETURNS")) {
                break;
            }
            usleep(2000); /* nothing more on the wire: idle until cancelled */
        } else if (pace_us) {
            usleep(pace_us);
        }
    }
    free(xfer);
    note("read_async_returned", 0, 0);
    return 0;
}

int rtlsdr_cancel_async(rtlsdr_dev_t *dev)
{
    if (NULL == dev) {
        return -1;
    }
    note("cancel_async", 0, 0);
    atomic_store(&dev->cancel, 1);
    return 0;
}
