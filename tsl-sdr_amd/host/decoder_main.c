/*
 * decoder_main.c - `decoder_amd`: the reference's `decoder` process (decoder/decoder.c), batched.
 *
 * The reference runs one decoder process per channel FIFO: read 1024 int16 samples, optional inversion,
 * polyphase resampler, optional DC blocker, protocol object, JSON lines (decoder.c:580-673).  Here every
 * input named on the command line is one channel of a single GPU pipeline
 *
 *     PCM block (host) -> mfm_resampler (I/D, -b, -i) -> mfm_pocsag (slicer / sync / BCH)      on the MI355X
 *                                                      or  mfm_flex   (sync 1 / FIW / slicer / de-interleave)
 *     events (host)    -> pager_pocsag_on_events -> on_alpha / on_numeric -> JSON line         per channel
 *                      or pager_flex_on_events   -> on_alnum / on_num / on_siv
 *
 * Same options as decoder.c:404 (-I -D -S -F -f -o -c -b -p -i -m -d is not offered), same JSON line layout
 * (decoder.c:173-318); -m FLEX is the default, as in decoder.c:59.  Differences, all forced by batching: several inputs are allowed (with more than one,
 * -o NAME writes NAME.0, NAME.1, ...); inputs are read in lock step and processing stops at the shortest;
 * -B sets the block size, -g the device.  MFM_DECODER_FIXED_TIME=1 prints the epoch instead of the wall clock,
 * so two runs can be diffed.  AIS is not part of this build.
 */
#include <errno.h>
#include <ctype.h>
#include <fcntl.h>
#include <inttypes.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>

#include "mfm_config.h"
#include "mfm_pager_flex.h"
#include "mfm_pager_pocsag.h"

#define DEC_MSG(sev, sys, msg, ...) MESSAGE("DECODER", sev, sys, msg, ##__VA_ARGS__)
#define Q_15_SHIFT 14 /* filter/filter.h:16 */

enum proto { PROTO_FLEX = 0, PROTO_POCSAG = 1 }; /* decoder.c:51-59 */

struct chan {
    int fd;
    FILE *out;
    struct pager_pocsag *pocsag;
    struct pager_flex *flex;
};

static bool g_fixed_time = false;

/* decoder.c:121-166 */
static void put_alnum_char(FILE *fp, char ch)
{
    switch (ch) {
    case '\n':
    case '\r':
        fprintf(fp, "\\n");
        break;
    case '\"':
        fprintf(fp, "\\\"");
        break;
    case '\\':
        fprintf(fp, "\\\\");
        break;
    case '/':
        fprintf(fp, "\\/");
        break;
    case '\b':
        fprintf(fp, "<BKSP>");
        break;
    case '\f':
        fprintf(fp, "<FF>");
        break;
    case '\t':
        fprintf(fp, "\\t");
        break;
    case 0x03:
    case 0x04:
    case 0x17:
        fprintf(fp, " ");
        break;
    default:
        if (isprint((unsigned char)ch)) {
            fprintf(fp, "%c", ch);
        } else {
            fprintf(fp, "\\u%04x", (unsigned)ch);
        }
    }
}

/* decoder.c:264-318 */
static aresult_t on_page(struct pager_pocsag *p, const char *type, uint16_t baud_rate, uint32_t capcode, const char *data,
                         size_t data_len, uint8_t function)
{
    struct chan *ch = pager_pocsag_get_user(p);
    time_t now = g_fixed_time ? 0 : time(NULL);
    struct tm *gmt = gmtime(&now);
    fprintf(ch->out,
            "{\"proto\":\"pocsag\",\"type\":\"%s\",\"timestamp\":\"%04i-%02i-%02i %02i:%02i:%02i UTC\","
            "\"baud\":%i,\"capCode\":%u,\"function\":%u,\"message\":\"",
            type, gmt->tm_year + 1900, gmt->tm_mon + 1, gmt->tm_mday, gmt->tm_hour, gmt->tm_min, gmt->tm_sec, baud_rate, capcode,
            (unsigned)function);
    for (size_t i = 0; i < data_len; i++) {
        put_alnum_char(ch->out, data[i]);
    }
    fprintf(ch->out, "\"}\n");
    fflush(ch->out);
    return A_OK;
}

static aresult_t on_alnum(struct pager_pocsag *p, uint16_t baud, uint32_t cap, const char *data, size_t len, uint8_t fn)
{
    return on_page(p, "alphanumeric", baud, cap, data, len, fn);
}

static aresult_t on_num(struct pager_pocsag *p, uint16_t baud, uint32_t cap, const char *data, size_t len, uint8_t fn)
{
    return on_page(p, "numeric", baud, cap, data, len, fn);
}

/* decoder.c:173-262 */
static void flex_head(struct chan *ch, const char *type, uint16_t baud, uint8_t phase, uint8_t cycle_no, uint8_t frame_no,
                      uint64_t cap_code)
{
    time_t now = g_fixed_time ? 0 : time(NULL);
    struct tm *gmt = gmtime(&now);
    fprintf(ch->out,
            "{\"proto\":\"flex\",\"type\":\"%s\",\"timestamp\":\"%04i-%02i-%02i %02i:%02i:%02i UTC\","
            "\"baud\":%i,\"syncLevel\":%i,\"frameNo\":%u,\"cycleNo\":%u,\"phaseNo\":\"%c\",\"capCode\":%" PRIu64 ",",
            type, gmt->tm_year + 1900, gmt->tm_mon + 1, gmt->tm_mday, gmt->tm_hour, gmt->tm_min, gmt->tm_sec, baud, 0, frame_no,
            cycle_no, "ABCD"[phase & 3], cap_code);
}

static aresult_t on_flex_alnum(struct pager_flex *f, uint16_t baud, uint8_t phase, uint8_t cycle_no, uint8_t frame_no,
                               uint64_t cap_code, bool fragmented, bool maildrop, uint8_t seq_num, const char *message_bytes,
                               size_t message_len)
{
    struct chan *ch = pager_flex_get_user(f);
    flex_head(ch, "alphanumeric", baud, phase, cycle_no, frame_no, cap_code);
    fprintf(ch->out, "\"fragment\":%s,\"maildrop\":%s,\"fragSeq\":%u,\"message\":\"", fragmented ? "true" : "false",
            maildrop ? "true" : "false", seq_num);
    for (size_t i = 0; i < message_len; i++) {
        put_alnum_char(ch->out, message_bytes[i]);
    }
    fprintf(ch->out, "\"}\n");
    fflush(ch->out);
    return A_OK;
}

static aresult_t on_flex_num(struct pager_flex *f, uint16_t baud, uint8_t phase, uint8_t cycle_no, uint8_t frame_no,
                             uint64_t cap_code, const char *message_bytes, size_t message_len)
{
    struct chan *ch = pager_flex_get_user(f);
    flex_head(ch, "numeric", baud, phase, cycle_no, frame_no, cap_code);
    fprintf(ch->out, "\"message\":\"");
    for (size_t i = 0; i < message_len; i++) {
        put_alnum_char(ch->out, message_bytes[i]);
    }
    fprintf(ch->out, "\"}\n");
    fflush(ch->out);
    return A_OK;
}

static aresult_t on_flex_siv(struct pager_flex *f, uint16_t baud, uint8_t phase, uint8_t cycle_no, uint8_t frame_no,
                             uint64_t cap_code, uint8_t siv_msg_type, uint32_t data)
{
    struct chan *ch = pager_flex_get_user(f);
    if (PAGER_FLEX_SIV_TEMP_ADDRESS_ACTIVATION == siv_msg_type) { /* the only one decoder.c prints (:253-260) */
        flex_head(ch, "tempAddrActivation", baud, phase, cycle_no, frame_no, cap_code);
        fprintf(ch->out, "\"startFrameNo\":%u,\"tempAddressId\":%u}\n", data & 0x7f, (data >> 7) & 0xf);
    }
    return A_OK;
}

static void usage(const char *app)
{
    DEC_MSG(SEV_INFO, "USAGE",
            "%s -I [interpolate] -D [decimate] -F [filter file] -S [input sample rate] -f [center freq] [-c] "
            "[-o output JSON file] [-b] [-p pole] [-i] [-m FLEX|POCSAG] [-B block samples] [-g gpu] in_fifo [in_fifo ...]",
            app);
    exit(EXIT_SUCCESS);
}

/* a whole block unless the input ends (FIFOs return short reads) */
static ssize_t read_full(int fd, void *buf, size_t bytes)
{
    size_t got = 0;
    while (got < bytes) {
        ssize_t r = read(fd, (char *)buf + got, bytes - got);
        if (r < 0) {
            if (EINTR == errno) {
                continue;
            }
            return -1;
        }
        if (0 == r) {
            break;
        }
        got += (size_t)r;
    }
    return (ssize_t)got;
}

int main(int argc, char *const argv[])
{
    unsigned interpolate = 1, decimate = 1, input_sample_rate = 0, center_freq = 0, block = 1u << 18;
    int device = 0, arg;
    bool dc_blocker = false, invert = false, create_out = false;
    enum proto proto = PROTO_FLEX;
    double dc_block_pole = 0.9999;
    const char *filter_file = NULL, *out_file_name = NULL;

    while ((arg = getopt(argc, argv, "co:I:D:S:F:f:p:m:B:g:bih")) != -1) {
        switch (arg) {
        case 'o':
            out_file_name = optarg;
            break;
        case 'c':
            create_out = true;
            break;
        case 'f':
            center_freq = (unsigned)strtoll(optarg, NULL, 0);
            break;
        case 'I':
            interpolate = (unsigned)strtoll(optarg, NULL, 0);
            break;
        case 'D':
            decimate = (unsigned)strtoll(optarg, NULL, 0);
            break;
        case 'S':
            input_sample_rate = (unsigned)strtoll(optarg, NULL, 0);
            break;
        case 'F':
            filter_file = optarg;
            break;
        case 'b':
            dc_blocker = true;
            break;
        case 'p':
            dc_block_pole = strtod(optarg, NULL);
            break;
        case 'i':
            invert = true;
            break;
        case 'm':
            if (!strncasecmp(optarg, "pocsag", 6)) {
                proto = PROTO_POCSAG;
            } else if (!strncasecmp(optarg, "flex", 4)) {
                proto = PROTO_FLEX;
            } else {
                DEC_MSG(SEV_ERROR, "UNKNOWN-PROTOCOL-TYPE", "POCSAG and FLEX are built into decoder_amd (asked for: %s)", optarg);
                exit(EXIT_FAILURE);
            }
            break;
        case 'B':
            block = (unsigned)strtoll(optarg, NULL, 0);
            break;
        case 'g':
            device = (int)strtol(optarg, NULL, 0);
            break;
        case 'h':
        default:
            usage(argv[0]);
        }
    }
    g_fixed_time = NULL != getenv("MFM_DECODER_FIXED_TIME");
    if (optind >= argc) {
        DEC_MSG(SEV_FATAL, "MISSING-SRC-DEST", "Missing source file / FIFO");
        exit(EXIT_FAILURE);
    }
    if (0 == decimate || 0 == interpolate || 0 == block) {
        DEC_MSG(SEV_FATAL, "BAD-DECIMATION", "Interpolation, decimation and block size must be non-zero integers.");
        exit(EXIT_FAILURE);
    }
    if (0 == center_freq) {
        DEC_MSG(SEV_FATAL, "BAD-PAGER-FREQ", "Pager frequency must be non-zero");
        exit(EXIT_FAILURE);
    }
    if (NULL == filter_file) {
        DEC_MSG(SEV_FATAL, "BAD-FILTER-FILE", "Need to specify a filter JSON file.");
        exit(EXIT_FAILURE);
    }
    DEC_MSG(SEV_INFO, "CONFIG", "Resampling: %u/%u from %u to %f", interpolate, decimate, input_sample_rate,
            ((double)interpolate / (double)decimate) * (double)input_sample_rate);

    /* decoder.c:520-533: lpfCoeffs -> Q14 by truncation */
    struct config *cfg = NULL;
    double *coeffs_f = NULL;
    size_t nr_coeffs = 0;
    TSL_BUG_IF_FAILED(config_new(&cfg));
    if (FAILED(config_add(cfg, filter_file))) {
        DEC_MSG(SEV_INFO, "BAD-CONFIG", "Configuration file '%s' cannot be processed, aborting.", filter_file);
        exit(EXIT_FAILURE);
    }
    TSL_BUG_IF_FAILED(config_get_float_array(cfg, &coeffs_f, &nr_coeffs, "lpfCoeffs"));
    int16_t *coeffs = calloc(nr_coeffs, sizeof(int16_t));
    TSL_BUG_ON(NULL == coeffs);
    for (size_t i = 0; i < nr_coeffs; i++) {
        coeffs[i] = (int16_t)(coeffs_f[i] * (double)(1 << Q_15_SHIFT));
    }

    const unsigned nr_chan = (unsigned)(argc - optind);
    struct chan *ch = calloc(nr_chan, sizeof(*ch));
    TSL_BUG_ON(NULL == ch);
    for (unsigned c = 0; c < nr_chan; c++) {
        if (0 > (ch[c].fd = open(argv[optind + c], O_RDONLY))) {
            DEC_MSG(SEV_INFO, "BAD-INPUT", "Bad input - cannot open %s", argv[optind + c]);
            exit(EXIT_FAILURE);
        }
        if (NULL == out_file_name) {
            ch[c].out = stdout;
        } else {
            char name[4096];
            if (1 == nr_chan) {
                snprintf(name, sizeof(name), "%s", out_file_name);
            } else {
                snprintf(name, sizeof(name), "%s.%u", out_file_name, c);
            }
            if (NULL == (ch[c].out = fopen(name, create_out ? "w+" : "a"))) {
                DEC_MSG(SEV_INFO, "BAD-OUTPUT-FILE", "Failed to open output file '%s', aborting.", name);
                exit(EXIT_FAILURE);
            }
        }
        if (PROTO_POCSAG == proto) {
            TSL_BUG_IF_FAILED(pager_pocsag_new(&ch[c].pocsag, center_freq, on_num, on_alnum, false));
            pager_pocsag_set_user(ch[c].pocsag, &ch[c]);
        } else {
            TSL_BUG_IF_FAILED(pager_flex_new(&ch[c].flex, center_freq, on_flex_alnum, on_flex_num, on_flex_siv));
            pager_flex_set_user(ch[c].flex, &ch[c]);
        }
    }
    DEC_MSG(SEV_INFO, "PROTOCOL", "%s", PROTO_POCSAG == proto ? "Using the POCSAG Pager Protocol." : "Using the Motorola FLEX pager protocol.");

    struct mfm_resampler *rs = NULL;
    struct mfm_pocsag *pg = NULL;
    struct mfm_resampler_config rc = { .abi_version = MFM_ABI_VERSION, .device = device, .nr_channels = nr_chan,
        .interpolate = interpolate, .decimate = decimate, .max_in_samples = block, .invert = invert,
        .dc_block = dc_blocker, .dc_pole = dc_block_pole };
    if (mfm_resampler_create(&rs, &rc, coeffs, nr_coeffs)) {
        DEC_MSG(SEV_FATAL, "NO-RESAMPLER", "Cannot create the GPU resampler: %s", mfm_last_error());
        exit(EXIT_FAILURE);
    }
    const uint32_t max_pcm = (uint32_t)mfm_resampler_max_out(rs);
    struct mfm_pocsag_event *events = NULL;
    struct mfm_flex_event *fevents = NULL;
    struct mfm_flex_frame_words *fframes = NULL;
    struct mfm_flex *fx = NULL;
    size_t max_events, max_frames = 0;
    if (PROTO_POCSAG == proto) {
        struct mfm_pocsag_config pc = { .abi_version = MFM_ABI_VERSION, .device = device, .nr_channels = nr_chan,
            .max_in_samples = max_pcm, .max_events = 0, .flags = 0 };
        if (mfm_pocsag_create(&pg, &pc)) {
            DEC_MSG(SEV_FATAL, "NO-PAGER-STAGE", "Cannot create the GPU POCSAG stage: %s", mfm_last_error());
            exit(EXIT_FAILURE);
        }
        max_events = (size_t)nr_chan * (max_pcm / 2048 + 16);
        events = calloc(max_events, sizeof(*events));
        TSL_BUG_ON(NULL == events);
    } else {
        struct mfm_flex_config fc = { .abi_version = MFM_ABI_VERSION, .device = device, .nr_channels = nr_chan,
            .max_in_samples = max_pcm, .max_events = 0, .flags = 0 };
        if (mfm_flex_create(&fx, &fc)) {
            DEC_MSG(SEV_FATAL, "NO-PAGER-STAGE", "Cannot create the GPU FLEX stage: %s", mfm_last_error());
            exit(EXIT_FAILURE);
        }
        max_events = (size_t)nr_chan * (max_pcm / 1024 + 8);
        max_frames = (size_t)nr_chan * (max_pcm / 28672 + 2);
        fevents = calloc(max_events, sizeof(*fevents));
        fframes = calloc(max_frames, sizeof(*fframes));
        TSL_BUG_ON(NULL == fevents || NULL == fframes);
    }
    int16_t *pcm = calloc((size_t)nr_chan * block, sizeof(int16_t));
    TSL_BUG_ON(NULL == pcm);

    size_t sample_count = 0;
    for (;;) {
        size_t n = block;
        for (unsigned c = 0; c < nr_chan; c++) {
            ssize_t got = read_full(ch[c].fd, pcm + (size_t)c * block, (size_t)block * sizeof(int16_t));
            if (got < 0) {
                DEC_MSG(SEV_FATAL, "READ-FIFO-FAIL", "Failed to read from input fifo: %s (%d)", strerror(errno), errno);
                got = 0;
            }
            if ((size_t)got / sizeof(int16_t) < n) {
                n = (size_t)got / sizeof(int16_t);
            }
        }
        if (0 == n) {
            break;
        }
        int16_t *d_out = NULL;
        size_t out_stride = 0, nr_out = 0, nr_events = 0;
        TSL_BUG_ON(MFM_OK != mfm_resampler_process_host_to_device(rs, pcm, block, n, NULL, &d_out, &out_stride, &nr_out));
        /* events come grouped by channel, in stream order inside a channel */
        if (PROTO_POCSAG == proto) {
            TSL_BUG_ON(MFM_OK != mfm_pocsag_process_device(pg, d_out, out_stride, nr_out, NULL));
            TSL_BUG_ON(MFM_OK != mfm_pocsag_fetch_events(pg, events, max_events, &nr_events));
            size_t first = 0;
            while (first < nr_events) {
                size_t last = first;
                while (last < nr_events && events[last].channel == events[first].channel) {
                    last++;
                }
                TSL_BUG_IF_FAILED(pager_pocsag_on_events(ch[events[first].channel].pocsag, &events[first], last - first));
                first = last;
            }
        } else {
            size_t nr_frames = 0;
            TSL_BUG_ON(MFM_OK != mfm_flex_process_device(fx, d_out, out_stride, nr_out, NULL));
            TSL_BUG_ON(MFM_OK != mfm_flex_fetch_events(fx, fevents, max_events, &nr_events, fframes, max_frames, &nr_frames));
            size_t first = 0;
            while (first < nr_events) {
                size_t last = first;
                while (last < nr_events && fevents[last].channel == fevents[first].channel) {
                    last++;
                }
                TSL_BUG_IF_FAILED(pager_flex_on_events(ch[fevents[first].channel].flex, &fevents[first], last - first, fframes));
                first = last;
            }
        }
        sample_count += n;
        if (n < block) {
            break;
        }
    }
    DEC_MSG(SEV_INFO, "TERMINATING", "Terminating processing loop, processed %zu samples per channel", sample_count);

    mfm_pocsag_destroy(&pg);
    mfm_flex_destroy(&fx);
    mfm_resampler_destroy(&rs);
    for (unsigned c = 0; c < nr_chan; c++) {
        if (NULL != ch[c].pocsag) {
            pager_pocsag_delete(&ch[c].pocsag);
        }
        if (NULL != ch[c].flex) {
            pager_flex_delete(&ch[c].flex);
        }
        close(ch[c].fd);
        if (ch[c].out != stdout) {
            fclose(ch[c].out);
        }
    }
    free(events);
    free(fevents);
    free(fframes);
    free(pcm);
    free(ch);
    free(coeffs);
    free(coeffs_f);
    config_delete(&cfg);
    return EXIT_SUCCESS;
}
