/*
 * multifm_main.c - `multifm_amd cfg1.json [cfg2.json ...]`: the multifm channelizer driver on the MI355X
 * engine.  Same command line and config merging as the reference (multifm/multifm.c:89-173); device
 * types other than "file" need vendor libraries that are not part of this build.
 */
#include "mfm_file_if.h"

#include <signal.h>
#include <unistd.h>

#define MFM_MSG(sev, sys, msg, ...) MESSAGE("MULTIFM", sev, sys, msg, ##__VA_ARGS__)

static volatile sig_atomic_t g_running = 1;

static void _on_sigint(int sig)
{
    (void)sig;
    g_running = 0;
}

int main(int argc, const char *argv[])
{
    int ret = EXIT_FAILURE;
    struct config *cfg = NULL;
    struct config device = CONFIG_INIT_EMPTY;
    struct receiver *rx_thr = NULL;
    const char *dev_type = NULL;

    if (argc < 2) {
        fprintf(stderr, "usage: %s [Config File 1]{, Config File 2, ...} | %s -h\n", argv[0], argv[0]);
        return EXIT_FAILURE;
    }
    TSL_BUG_IF_FAILED(config_new(&cfg));
    for (int i = 1; i < argc; i++) {
        if (FAILED(config_add(cfg, argv[i]))) {
            MFM_MSG(SEV_FATAL, "MALFORMED-CONFIG", "Configuration file [%s] is malformed.", argv[i]);
            goto done;
        }
    }
    signal(SIGINT, _on_sigint);
    signal(SIGPIPE, SIG_IGN); /* EPIPE is handled per channel (demod.c:95-105) */

    if (FAILED(config_get(cfg, &device, "device"))) {
        MFM_MSG(SEV_FATAL, "MALFORMED-CONFIG", "Configuration is missing 'device' stanza. Aborting.");
        goto done;
    }
    if (FAILED(config_get_string(&device, &dev_type, "type"))) {
        MFM_MSG(SEV_FATAL, "MALFORMED-CONFIG", "The 'device' stanza is missing a 'type' specification. Aborting.");
        goto done;
    }
    if (!strncmp(dev_type, "file", 4)) {
        if (FAILED(file_worker_thread_new(&rx_thr, cfg))) {
            goto done;
        }
    } else if (!strncmp(dev_type, "rtlsdr", 6) || !strncmp(dev_type, "airspy", 6) || !strncmp(dev_type, "usrp", 4)) {
        MFM_MSG(SEV_FATAL, "DEVICE-NOT-SUPPORTED", "'%s' devices are not supported by this build.", dev_type);
        goto done;
    } else {
        MFM_MSG(SEV_FATAL, "UNKNOWN-DEV-TYPE", "Unknown device type: '%s'", dev_type);
        goto done;
    }

    TSL_BUG_IF_FAILED(receiver_set_mute(rx_thr, false));
    MFM_MSG(SEV_INFO, "CAPTURING", "Starting capture and demodulation process.");
    if (FAILED(receiver_start(rx_thr))) {
        goto done;
    }
    /* a file runs out; the reference's loop only ends on SIGINT (multifm.c:163-165) */
    while (g_running && !rx_thr->input_done) {
        usleep(10000);
    }
    ret = EXIT_SUCCESS;

done:
    if (NULL != rx_thr) {
        receiver_cleanup(&rx_thr);
    }
    config_delete(&cfg);
    return ret;
}
