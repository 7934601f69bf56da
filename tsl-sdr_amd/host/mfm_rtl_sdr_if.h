/*
 * mfm_rtl_sdr_if.h - the RTL-SDR front end of multifm on the MI355X receiver.
 *
 * Same constructor as the reference's multifm/rtl_sdr_if.h:27 (and the same configuration keys, messages and tuner
 * set-up order as multifm/rtl_sdr_if.c:308-479).  librtlsdr is bound at run time (dlopen), so the host library builds
 * and loads on a machine without it; rtl_sdr_worker_thread_new() fails with RTLSDR-NOT-SUPPORTED when the library is
 * not there.  The reference's own rtl_sdr_if.c also compiles unchanged against host/compat (tests/test_host.py).
 */
#pragma once

#include "mfm_receiver.h"

aresult_t rtl_sdr_worker_thread_new(struct receiver **pthr, struct config *cfg);
/* lists the devices librtlsdr sees on stderr, as multifm's usage text does (multifm/multifm.c:57-77); false when the
 * library is not available */
bool rtl_sdr_dump_devices(void);
