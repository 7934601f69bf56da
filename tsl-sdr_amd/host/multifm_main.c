/*
 * multifm_main.c - `multifm_amd cfg1.json [cfg2.json ...]`: the multifm channelizer driver on the MI355X engine.
 *
 * Command line, configuration merging and the messages a user sees follow the reference's driver
 * (multifm/multifm.c:89-173): every argument is a JSON file merged into one configuration, whose "device" stanza
 * names the front end by "type".  The front ends this build knows sit in a table: a name (matched as a prefix, like the
 * reference's strncmp chain) and the constructor that turns the configuration into a running receiver, or NULL for the
 * types whose vendor libraries (libdespairspy, UHD) are not part of this build.  librtlsdr is bound at run time.
 */
#include "mfm_file_if.h"
#include "mfm_rtl_sdr_if.h"

#include <unistd.h>

#define MFM_MSG(sev, sys, msg, ...) MESSAGE("MULTIFM", sev, sys, msg, ##__VA_ARGS__)

typedef aresult_t (*front_end_new_func_t)(struct receiver **prx, struct config *cfg);

static const struct front_end {
    const char *type;            /* "device": { "type": ... } starts with this */
    front_end_new_func_t create; /* NULL: known to multifm, not built here */
} front_ends[] = {
    { "file", file_worker_thread_new },
    { "rtlsdr", rtl_sdr_worker_thread_new }, /* librtlsdr bound at run time; fails politely without it */
    { "airspy", NULL },
    { "usrp", NULL },
};

/* every command-line argument is one more file of the same configuration */
static struct config *load_configuration(int nr_files, const char *files[])
{
    struct config *cfg = NULL;
    TSL_BUG_IF_FAILED(config_new(&cfg));
    for (int i = 0; i < nr_files; i++) {
        if (FAILED(config_add(cfg, files[i]))) {
            MFM_MSG(SEV_FATAL, "MALFORMED-CONFIG", "Configuration file [%s] is malformed.", files[i]);
            config_delete(&cfg);
            return NULL;
        }
    }
    return cfg;
}

static const struct front_end *find_front_end(struct config *cfg)
{
    struct config device = CONFIG_INIT_EMPTY;
    const char *type = NULL;

    if (FAILED(config_get(cfg, &device, "device"))) {
        MFM_MSG(SEV_FATAL, "MALFORMED-CONFIG", "Configuration is missing 'device' stanza. Aborting.");
        return NULL;
    }
    if (FAILED(config_get_string(&device, &type, "type"))) {
        MFM_MSG(SEV_FATAL, "MALFORMED-CONFIG", "The 'device' stanza is missing a 'type' specification. Aborting.");
        return NULL;
    }
    for (size_t i = 0; i < sizeof(front_ends) / sizeof(front_ends[0]); i++) {
        if (0 == strncmp(type, front_ends[i].type, strlen(front_ends[i].type))) {
            if (NULL == front_ends[i].create) {
                MFM_MSG(SEV_FATAL, "DEVICE-NOT-SUPPORTED", "'%s' devices are not supported by this build.", type);
                return NULL;
            }
            return &front_ends[i];
        }
    }
    MFM_MSG(SEV_FATAL, "UNKNOWN-DEV-TYPE", "Unknown device type: '%s'", type);
    return NULL;
}

/* unmute, start, and stay until the operator interrupts or the input runs out (a file does; the reference's loop only ends
 * on SIGINT, multifm.c:163-165) */
static int run_receiver(struct receiver *rx)
{
    TSL_BUG_IF_FAILED(receiver_set_mute(rx, false));
    MFM_MSG(SEV_INFO, "CAPTURING", "Starting capture and demodulation process.");
    if (FAILED(receiver_start(rx))) {
        return EXIT_FAILURE;
    }
    while (app_running() && !rx->input_done) {
        usleep(10000);
    }
    return EXIT_SUCCESS;
}

int main(int argc, const char *argv[])
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s [Config File 1]{, Config File 2, ...} | %s -h\n", argv[0], argv[0]);
        (void)rtl_sdr_dump_devices(); /* multifm/multifm.c:83-85 */
        return EXIT_FAILURE;
    }
    struct config *cfg = load_configuration(argc - 1, argv + 1);
    if (NULL == cfg) {
        return EXIT_FAILURE;
    }
    /* the application scaffolding of the reference's driver (multifm.c:114-115): SIGINT / SIGTERM end the loop below,
     * EPIPE is an errno handled per channel (demod.c:95-105) */
    TSL_BUG_IF_FAILED(app_init("multifm", cfg));
    TSL_BUG_IF_FAILED(app_sigint_catch(NULL));

    int ret = EXIT_FAILURE;
    struct receiver *rx = NULL;
    const struct front_end *fe = find_front_end(cfg);
    if (NULL != fe && !FAILED(fe->create(&rx, cfg))) {
        ret = run_receiver(rx);
    }
    if (NULL != rx) {
        receiver_cleanup(&rx);
    }
    config_delete(&cfg);
    return ret;
}
